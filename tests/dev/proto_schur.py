"""NumPy prototype of the substructured (interface Schur complement) snapshot solver.

Design scratch for the HIP path (romhighcontrast_amd/csrc): validates the algebra and the
floating-point behaviour against the oracle before any kernel is written.  Not product code,
not imported by anything.
"""
import sys
import time

import numpy as np
import scipy.linalg

sys.path.insert(0, ".")
from oracle import rom_oracle as ro  # noqa: E402  (tools/ is scratch, like tests/)


class UnitBlock:
    """Parameter-independent tables of one N x N-cell unit block (Dirichlet Laplacian)."""

    def __init__(self, N):
        self.N = N
        n1 = N - 1
        j = np.arange(1, N)
        self.Q = np.sqrt(2.0 / N) * np.sin(np.pi * np.outer(j, j) / N)  # symmetric orthonormal
        lam = 2.0 - 2.0 * np.cos(np.pi * j / N)
        phi = np.arccosh(1.0 + lam / 2.0)
        i = np.arange(0, N + 1)
        # rho[m, i] = sinh((N-i) phi_m) / sinh(N phi_m), overflow-free
        e = np.exp(-np.outer(phi, i))
        self.rho = e * (1 - np.exp(-2 * np.outer(phi, N - i))) / (1 - np.exp(-2 * N * phi))[:, None]
        self.lam = lam
        s = self.Q.sum(axis=0)
        self.W = self.Q @ (np.outer(s, s) / (lam[:, None] + lam[None, :])) @ self.Q  # L^{-1} 1, (i,j)
        # harmonic extension matrices H[s] : ((N-1)^2, N-1), rows (i,j) row-major, i,j=1..N-1
        Q, rho = self.Q, self.rho
        H0 = np.einsum("jm,mi,km->ijk", Q, rho[:, 1:N], Q)  # side i=0
        H1 = H0[::-1]  # side i=N : i -> N-i
        H2 = np.einsum("im,mj,km->ijk", Q, rho[:, 1:N], Q)  # side j=0
        H3 = H2[:, ::-1]
        self.H = [h.reshape(n1 * n1, n1) for h in (H0, H1, H2, H3)]
        t = np.arange(n1)
        adj = [0 * n1 + t, (n1 - 1) * n1 + t, t * n1 + 0, t * n1 + (n1 - 1)]  # rows next to each side
        self.adj = adj
        self.T = [[self.H[sc][adj[sr]] for sc in range(4)] for sr in range(4)]
        self.Wadj = [self.W.reshape(-1)[adj[s]] for s in range(4)]


class Substructure:
    def __init__(self, blocks, N):
        self.g = ro.Geometry(blocks, N)
        self.ub = UnitBlock(N)
        g = self.g
        n1 = N - 1
        nrb, ncb = g.nrb, g.ncb
        # interface numbering: horizontal edges, vertical edges, cross points
        self.hedge = {}
        self.vedge = {}
        self.cross = {}
        off = 0
        for p in range(1, nrb):
            for q in range(ncb):
                self.hedge[(p, q)] = off
                off += n1
        for q in range(1, ncb):
            for p in range(nrb):
                self.vedge[(p, q)] = off
                off += n1
        for p in range(1, nrb):
            for q in range(1, ncb):
                self.cross[(p, q)] = off
                off += 1
        self.nG = off
        # per block: side -> interface offset (or None if on the domain boundary)
        self.sides = {}
        for p in range(nrb):
            for q in range(ncb):
                self.sides[(p, q)] = [
                    self.hedge.get((p, q)) if p >= 1 else None,  # side 0: i=0  (r = pN)
                    self.hedge.get((p + 1, q)) if p + 1 < nrb else None,  # side 1: i=N
                    self.vedge.get((p, q)) if q >= 1 else None,  # side 2: j=0  (c = qN)
                    self.vedge.get((p, q + 1)) if q + 1 < ncb else None,  # side 3: j=N
                ]
        h2 = 1.0 / (N * N)
        gvec = np.full(self.nG, h2)
        for (p, q), s in self.sides.items():
            for sd in range(4):
                if s[sd] is not None:
                    gvec[s[sd]:s[sd] + n1] += h2 * self.ub.Wadj[sd]
        self.gvec = gvec

    def assemble_S(self, a):
        g, N = self.g, self.g.N
        n1 = N - 1
        a = np.asarray(a, dtype=np.float64).reshape(g.nrb, g.ncb)
        S = np.zeros((self.nG, self.nG))
        ar = np.arange(n1)
        for (p, q), o in self.hedge.items():
            up, dn = a[p - 1, q], a[p, q]
            S[o + ar, o + ar] = up + up + dn + dn  # order of the oracle's diag formula
            cpl = -(dn + up) / 2
            S[o + ar[:-1], o + ar[1:]] = cpl
            S[o + ar[1:], o + ar[:-1]] = cpl
            if (p, q) in self.cross:
                x = self.cross[(p, q)]
                S[x, o] = S[o, x] = cpl
            if (p, q + 1) in self.cross:
                x = self.cross[(p, q + 1)]
                S[x, o + n1 - 1] = S[o + n1 - 1, x] = cpl
        for (p, q), o in self.vedge.items():
            lf, rt = a[p, q - 1], a[p, q]
            S[o + ar, o + ar] = lf + rt + lf + rt
            cpl = -(rt + lf) / 2
            S[o + ar[:-1], o + ar[1:]] = cpl
            S[o + ar[1:], o + ar[:-1]] = cpl
            if (p, q) in self.cross:
                x = self.cross[(p, q)]
                S[x, o] = S[o, x] = cpl
            if (p + 1, q) in self.cross:
                x = self.cross[(p + 1, q)]
                S[x, o + n1 - 1] = S[o + n1 - 1, x] = cpl
        for (p, q), x in self.cross.items():
            S[x, x] = a[p - 1, q - 1] + a[p - 1, q] + a[p, q - 1] + a[p, q]
        for (p, q), s in self.sides.items():
            for sr in range(4):
                if s[sr] is None:
                    continue
                for sc in range(4):
                    if s[sc] is None:
                        continue
                    S[s[sr]:s[sr] + n1, s[sc]:s[sc] + n1] -= a[p, q] * self.ub.T[sr][sc]
        return S

    def solve(self, a):
        g, N = self.g, self.g.N
        n1 = N - 1
        a = np.asarray(a, dtype=np.float64).reshape(g.nrb, g.ncb)
        S = self.assemble_S(a)
        c = scipy.linalg.cho_factor(S, lower=True)
        uG = scipy.linalg.cho_solve(c, self.gvec)
        U = np.zeros((g.nr + 2, g.nc + 2))  # vertex grid incl. boundary
        h2 = 1.0 / (N * N)
        for (p, q), s in self.sides.items():
            ui = (h2 / a[p, q]) * self.ub.W.reshape(-1)
            for sd in range(4):
                if s[sd] is not None:
                    ui = ui + self.ub.H[sd] @ uG[s[sd]:s[sd] + n1]
            U[p * N + 1:(p + 1) * N, q * N + 1:(q + 1) * N] = ui.reshape(n1, n1)
        for (p, q), o in self.hedge.items():
            U[p * N, q * N + 1:(q + 1) * N] = uG[o:o + n1]
        for (p, q), o in self.vedge.items():
            U[p * N + 1:(p + 1) * N, q * N] = uG[o:o + n1]
        for (p, q), x in self.cross.items():
            U[p * N, q * N] = uG[x]
        return U[1:-1, 1:-1].reshape(-1), np.linalg.cond(S)


def check(blocks, N, a_list, label):
    sub = Substructure(blocks, N)
    g = sub.g
    B = ro.load_vector(g)
    worst = 0
    for a in a_list:
        u, cond = sub.solve(a)
        uo = ro.solve_one(g, a, B, "lsqsparse")
        ud = ro.solve_one(g, a, B, "lsq") if g.dim <= 4000 else uo
        e = ro.H10norm(g, (u - uo)[None])[0] / ro.H10norm(g, uo[None])[0]
        e2 = ro.H10norm(g, (ud - uo)[None])[0] / ro.H10norm(g, uo[None])[0]
        worst = max(worst, e)
        print(f"{label} N={N} contrast={np.max(a)/np.min(a):.1e} cond(S)={cond:.2e} "
              f"schur-vs-sparse {e:.2e}   (dense-vs-sparse {e2:.2e})")
    return worst


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    check((2, 2), 10, [np.ones((2, 2)), [[1, 1], [1, 100]], [[1, 2], [3, 4]]], "2x2")
    check((2, 2), 16, 10.0 ** rng.uniform(0, 8, size=(4, 2, 2)), "2x2")
    check((2, 3), 5, 10.0 ** rng.uniform(0, 2, size=(3, 2, 3)), "2x3")
    check((3, 2), 4, 10.0 ** rng.uniform(0, 2, size=(3, 3, 2)), "3x2")
    a = np.ones((3, 3)); a[1, 1] = 1e10
    a2 = np.ones((3, 3)); a2[0, 0] = 1e10
    check((3, 3), 11, list(10.0 ** rng.uniform(0, 8, size=(3, 3, 3))) + [a, a2], "3x3")
    a = np.ones((4, 4)); a[2, 2] = 1e10
    check((4, 4), 8, list(10.0 ** rng.uniform(0, 8, size=(3, 4, 4))) + [a], "4x4")
    t = time.time()
    check((2, 2), 64, 10.0 ** rng.uniform(0, 2, size=(2, 2, 2)), "2x2")
    check((2, 2), 128, 10.0 ** rng.uniform(0, 2, size=(2, 2, 2)), "2x2")
    print("time", time.time() - t)
