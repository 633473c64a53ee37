"""Snapshot blocks in factored form.

A snapshot row of libromhc is a fixed linear image of its system's *interface vector* (reduced unknowns,
cross-point values, coefficient blocks and the scalars 1/(a_p+a_q), h^2/a_b; ``Fem.reduced_stride`` doubles:
784 against 65 025 at 256x256 / 2x2):

    U = Y B^T ,     B (dim x K) parameter independent  (``Fem.expansion_is_linear``)

so everything the basis stage needs from a snapshot block can be formed from ``Y``:

    Gram      U U^T            = Y S Y^T ,  S = B^T B   (K x K, once per FE space)
    mean row  mean(U)          = expand(mean(Y))
    POD mode  sum_m w_m U_m    = expand(sum_m w_m Y_m)

The (M, dim) block itself is never needed: it is what the GPUs of a node exchange (sweep.py) and what the
POD of a gathered sweep works on (``pod_modes_factored``); rows are materialised on demand (``rows``).
The reference has no counterpart (its snapshots are plain NumPy rows, src/lib/SolutionsManagers.py:64-68).
"""
from __future__ import annotations

import numpy as np

from . import _ffi


class ExpansionMap:
    """S = B^T B of one FE space (device, K x K) and the expansion itself."""

    def __init__(self, sm):
        fem, ctx = sm._fem, sm._ctx
        if not fem.expansion_is_linear:
            raise _ffi.RomLibraryError("this geometry recovers some edges node by node: its expansion is not a "
                                       "linear map of the interface vectors (use snapshot rows)")
        self.sm, self.fem, self.ctx = sm, fem, ctx
        self.K, self.dim = fem.reduced_stride, fem.dim
        self._ones = {}
        # the expansion only reads part of an interface vector (not the nodal edge blocks): all algebra below
        # runs on those Kc "input" coordinates -- the compact form that also travels between ranks
        # (rom_fem_pack_reduced_async / rom_fem_unpack_reduced_async move between the two)
        self.inputs = fem.reduced_inputs
        self.Kc = Kc = fem.compact_stride
        Bt = self._basis_rows()
        self.S = ctx.alloc(Kc * Kc)
        ctx.gram(Kc, self.dim, Bt, 0, self.dim, self.S, 0, Kc)
        ctx.synchronize()
        del Bt

    def _basis_rows(self):
        """B^T restricted to the input coordinates: (Kc, dim) on the device."""
        Bt = self.ctx.alloc(self.Kc * self.dim)
        self.expand_compact(self.ctx.upload(np.eye(self.Kc)), self.Kc, Bt)
        return Bt

    def compact(self, Y, M):
        """(M, Kc) input coordinates of the interface vectors Y (M, K)."""
        Yc = self.ctx.alloc(max(M * self.Kc, 1))
        if M:
            self.fem.pack_reduced(Y, M, Yc)
        return Yc

    def expand_compact(self, Wc, n, U, row0=0):
        """U[row0:row0+n] = expansion of n vectors given in input coordinates (n, Kc)."""
        W = self.ctx.alloc(max(n * self.K, 1))
        if n:
            self.fem.unpack_reduced(Wc, n, W)
        self.expand_into(W, n, U, row0=row0)

    # ---- H^1_0 geometry of the snapshots in coordinates of the interface vectors ----------------------------
    def energy_coordinates(self):
        """(E, Mb, beta): with xi = y E (k' numbers per snapshot) the H^1_0 inner product of two snapshots is the
        EUCLIDEAN inner product of their xi (E = V Lambda^(1/2) from S1 = B^T A_1 B = V Lambda V^T, directions below
        1e-15 of the largest eigenvalue dropped); Mb[b] (k' x k') is the form u^T A_b v of block b in these
        coordinates and beta (k') the load functional u -> u . B_total.  Built once per FE space."""
        if getattr(self, "_energy", None) is not None:
            return self._energy
        ctx, fem, K, dim = self.ctx, self.fem, self.Kc, self.dim
        Bt = self._basis_rows()
        ABt = ctx.alloc(K * dim)
        Sd = ctx.alloc(K * K)
        fem.stencil_apply(Bt, K, ABt)                      # rows: A_1 B e_i
        ctx.gemm_nt(K, K, dim, Bt, 0, dim, ABt, 0, dim, Sd, 0, K)
        S1 = Sd.download(K * K, shape=(K, K))
        S1 = (S1 + S1.T) / 2
        # the columns of B have wildly different energies (the slot of h^2/a_b carries L^-1 1 ~ N^2, others O(1)):
        # equilibrate before the eigen-decomposition, otherwise its absolute error (eps * lambda_max) swamps
        # the energy of ordinary snapshots
        d = np.sqrt(np.maximum(np.diag(S1), 0.0))
        d[d == 0] = 1.0
        lam, V = np.linalg.eigh(S1 / np.outer(d, d))
        keep = lam > 1e-15 * lam[-1]
        lam, V = lam[keep][::-1], V[:, keep][:, ::-1]
        E = (V * np.sqrt(lam)) * d[:, None]                # K x k'
        Einv = (V / np.sqrt(lam)) / d[:, None]             # y-coordinates of the xi basis vectors, K x k'
        Mb = []
        for b in range(fem.kblk):
            one = np.zeros(fem.kblk)
            one[b] = 1.0
            fem.stencil_apply(Bt, K, ABt, a_one=one)
            ctx.gemm_nt(K, K, dim, Bt, 0, dim, ABt, 0, dim, Sd, 0, K)
            Sb = Sd.download(K * K, shape=(K, K))
            Mb.append(Einv.T @ ((Sb + Sb.T) / 2) @ Einv)
        bt = ctx.alloc(K)
        ctx.gemm_nt(K, 1, dim, Bt, 0, dim, ctx.upload(fem.load_vector()), 0, dim, bt, 0, 1)
        beta = Einv.T @ bt.download(K)
        self._energy = (E, np.array(Mb), beta)
        return self._energy

    def l2_coordinates(self):
        """(E, Einv): with zeta = y E (k' numbers per snapshot) the EUCLIDEAN inner product of two snapshot rows is the
        Euclidean inner product of their zeta (E = V Lambda^(1/2) from S = B^T B = V Lambda V^T, directions below 1e-15 of
        the largest eigenvalue dropped), and a direction r in zeta coordinates is the snapshot-space vector B (Einv r) with
        Einv = V Lambda^(-1/2): B Einv has orthonormal columns.  The POD of a factored block is therefore the POD of the
        small (M, k') matrix Y E.  Built once per FE space (host eigh of a Kc x Kc matrix, equilibrated like
        energy_coordinates)."""
        if getattr(self, "_l2", None) is not None:
            return self._l2
        K = self.Kc
        S = self.S.download(K * K, shape=(K, K))
        S = (S + S.T) / 2
        d = np.sqrt(np.maximum(np.diag(S), 0.0))
        d[d == 0] = 1.0
        lam, V = np.linalg.eigh(S / np.outer(d, d))
        keep = lam > 1e-15 * lam[-1]
        lam, V = lam[keep][::-1], V[:, keep][:, ::-1]
        self._l2 = ((V * np.sqrt(lam)) * d[:, None], (V / np.sqrt(lam)) / d[:, None])
        return self._l2

    def _a_dummy(self, M):
        if M not in self._ones:
            self._ones = {M: self.ctx.upload(np.ones((M, self.fem.kblk)))}
        return self._ones[M]

    def expand_into(self, Y, M, U, y_row0=0, row0=0):
        """U[row0:row0+M] = Y[y_row0:y_row0+M] B^T (the parameters are not read by a linear expansion)."""
        self.fem.expand(self._a_dummy(M), M, Y, U, y_row0=y_row0, row0=row0)
        self.ctx.solve_status()


def expansion_map(sm) -> ExpansionMap:
    em = getattr(sm, "_expansion_map", None)
    if em is None:
        em = sm._expansion_map = ExpansionMap(sm)
    return em


class FactoredSnapshots:
    """M snapshots held as their interface vectors ``Y`` (device, (M, K) row-major)."""

    def __init__(self, sm, Y: "_ffi.Buffer", M: int):
        self.sm, self.Y, self.M = sm, Y, int(M)
        self.map = expansion_map(sm)

    @property
    def K(self):
        return self.map.K

    @property
    def Yc(self):
        """(M, Kc) input coordinates of the interface vectors (what the Gram / POD / greedy algebra works on)."""
        if getattr(self, "_Yc", None) is None:
            self._Yc = self.map.compact(self.Y, self.M)
        return self._Yc

    def rows(self, lo=0, hi=None):
        """Materialise snapshot rows [lo, hi) as a DeviceArray."""
        from .lib.SolutionsManagers import DeviceArray
        hi = self.M if hi is None else hi
        n = max(hi - lo, 0)
        U = self.map.ctx.alloc(max(n * self.map.dim, 1))
        if n:
            self.map.expand_into(self.Y, n, U, y_row0=lo)
        return DeviceArray(U, n, self.map.dim)

    def take(self, idx) -> "FactoredSnapshots":
        """The sub-block of the snapshots ``idx`` (a gather of interface vectors)."""
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        Y = self.map.ctx.alloc(max(idx.size * self.K, 1))
        if idx.size:
            Y.gather_rows_from(self.Y, idx, self.K)
        return FactoredSnapshots(self.sm, Y, idx.size)

    def gram(self):
        """U U^T as a device buffer (M x M), from Y alone."""
        ctx, K, M = self.map.ctx, self.map.Kc, self.M
        T = ctx.alloc(M * K)
        ctx.gemm_nt(M, K, K, self.Yc, 0, K, self.map.S, 0, K, T, 0, K)  # T = Y S (S symmetric)
        G = ctx.alloc(M * M)
        ctx.gemm_nt(M, M, K, T, 0, K, self.Yc, 0, K, G, 0, M)
        return G


def pod_modes_factored(fs: FactoredSnapshots, n: int, center=True):
    """Leading ``n`` POD modes / singular values of the snapshot block ``fs`` without ever forming it.

    With U = Y B^T and S = B^T B = E E^T (``ExpansionMap.l2_coordinates``) the rows of Z = Y E (M x k', k' <= a few
    hundred) have the same Euclidean geometry as the snapshot rows, so the POD of the block IS the POD of Z: the same
    routine as for rows (``lib.ReducedBasis.pod_modes``: Gram matrix on MFMA, deflation + sketches for the small modes,
    Rayleigh-Ritz) runs on a matrix dim / k' times narrower, and a mode r in zeta coordinates is expanded as
    B (Einv r).  Cost O(M^2 k' + n K dim) instead of O(M^2 dim).  Returns (modes (n, dim) NumPy, singular values); rows
    follow scikit-learn's ``svd_flip(u_based_decision=False)`` sign convention (the PCA call at
    src/lib/ReducedBasis.py:196)."""
    from .lib.ReducedBasis import pod_modes
    from .lib.SolutionsManagers import DeviceArray
    em, M, K = fs.map, fs.M, fs.map.Kc
    ctx, dim = em.ctx, em.dim
    E, Einv = em.l2_coordinates()
    kp = E.shape[1]
    n = min(n, M, dim)
    Yc = ctx.alloc(M * K).copy_from(fs.Yc, M * K)
    if center:
        ctx.center_rows(Yc, M, K, ctx.alloc(K))  # the expansion is linear: the mean row is the expansion of the mean vector
    Z = ctx.alloc(max(M * kp, 1))
    ctx.gemm_nn(M, kp, K, Yc, 0, K, ctx.upload(E), 0, kp, Z, 0, kp)
    nz = min(n, kp)
    modes_z, sig_z = pod_modes(ctx, DeviceArray(Z, M, kp), nz, center=False)
    info = dict(pod_modes.last_info)
    sig = np.zeros(n)
    sig[:nz] = sig_z
    if n == 0:
        return np.zeros((0, dim)), sig
    W = modes_z @ Einv.T                       # (nz, Kc): the modes as interface vectors
    V = ctx.alloc(n * dim)
    em.expand_compact(ctx.upload(W), nz, V)
    # B Einv has orthonormal columns to the accuracy of the eigen-decomposition of S (~1e-9): clean up the expanded modes
    ctx.symmetric_orthonormalize(V, nz, dim)
    if nz < n:
        # more modes requested than the snapshot manifold has dimensions: completed like pod_modes completes what the
        # data do not determine (orthonormal directions of singular value 0)
        ctx.complete_orthonormal(V, nz, n - nz, dim)
        info["completed_modes"] = info.get("completed_modes", 0) + n - nz
    ctx.rows_sign_flip(V, n, dim)  # svd_flip(u_based_decision=False)
    info["executed_flops"] = info.get("executed_flops", 0.0) + 2.0 * M * K * kp + 2.0 * nz * K * dim
    info.pop("useful_flops", None)
    pod_modes_factored.last_info = info
    return V.download(n * dim, shape=(n, dim)), sig


def energy_coordinates_of(fs: FactoredSnapshots):
    """Xi (device, (M, k')): the snapshots of ``fs`` in coordinates in which the H^1_0 inner product is Euclidean."""
    em = fs.map
    E, _, _ = em.energy_coordinates()
    ctx, M, K, kp = em.ctx, fs.M, em.Kc, E.shape[1]
    Xi = ctx.alloc(max(M * kp, 1))
    ctx.gemm_nn(M, kp, K, fs.Yc, 0, K, ctx.upload(E), 0, kp, Xi, 0, kp)
    return Xi, kp


def h10norm_factored(fs: FactoredSnapshots) -> np.ndarray:
    """H^1_0 norms of the snapshots from their interface vectors (= ``sm.H10norm(fs.rows())``)."""
    Xi, kp = energy_coordinates_of(fs)
    return fs.map.ctx.l2norm(Xi, 0, fs.M, kp)


def greedy_factored(fs: FactoredSnapshots, a2train, n: int, galerkin: bool, h1norm):
    """The strong greedy of ReducedBasisGreedy.build (src/lib/ReducedBasis.py:105-139) on a training block held in
    factored form.  In energy coordinates the H^1_0-orthogonal projection onto the picked snapshots is a Euclidean
    projection and the Galerkin ROM is a batch of small dense solves, so an iteration costs O(M k' n) instead of
    O(M dim n).  Returns (picks, max relative errors per iteration)."""
    em = fs.map
    ctx, fem, M = em.ctx, em.fem, fs.M
    E, Mb, beta = em.energy_coordinates()
    Xi, kp = energy_coordinates_of(fs)
    h1norm = np.broadcast_to(np.asarray(h1norm, dtype=np.float64), (M,))
    a2train = np.ascontiguousarray(np.asarray(a2train, dtype=np.float64).reshape(M, -1))
    picks, max_errors = [], []
    Q = np.zeros((0, kp))                        # H^1_0-orthonormal basis of the picks (energy coordinates), host
    R = ctx.alloc(M * kp).copy_from(Xi, M * kp)  # H10 mode: residuals of all training snapshots
    Ahat = np.zeros((fem.kblk, 0, 0))
    a_dev = ctx.upload(a2train)
    for it in range(n):
        if Q.shape[0] == 0:
            # empty basis: the approximation is zero and the error of snapshot i is ||u_i|| / h1norm_i (:129).  With
            # the snapshots' own norms as h1norm (what experiment() passes) the reference gets exactly 1.0 for every i
            # and argmax takes index 0; the norms formed here in energy coordinates agree with a caller's stencil norms
            # to rounding only, so quotients within 1e-10 of 1 are that tie.  Any other normalisation (the documented
            # default h1norm = 1 included) picks the largest ||u_i|| / h1norm_i, as the reference does.
            rel = ctx.l2norm(Xi, 0, M, kp) / h1norm
            rel = np.where(np.abs(rel - 1.0) <= 1e-10, 1.0, rel)
        elif not galerkin:
            rel = ctx.l2norm(R, 0, M, kp) / h1norm
        else:
            m = Q.shape[0]
            c = ctx.alloc(M * m)
            ctx.reduced_solve_batch(m, fem.kblk, M, ctx.upload(Ahat), a_dev, ctx.upload(Q @ beta), False, c)
            D = ctx.alloc(M * kp).copy_from(Xi, M * kp)
            ctx.gemm_nn(M, kp, m, c, 0, m, ctx.upload(Q), 0, kp, D, 0, kp, alpha=-1.0, beta=1.0)
            rel = ctx.l2norm(D, 0, M, kp) / h1norm
        ix = int(np.argmax(rel))
        picks.append(ix)
        max_errors.append(float(rel[ix]))
        if it == n - 1:
            break
        # new basis vector: the pick, orthogonalised (twice) against the earlier ones
        q = Xi.download(kp, offset=ix * kp)
        for _ in range(2):
            q = q - Q.T @ (Q @ q)
        nq = np.linalg.norm(q)
        if nq <= 1e-14 * np.linalg.norm(Xi.download(kp, offset=ix * kp)):
            continue                             # a duplicate pick (errors at roundoff): the span does not grow
        q /= nq
        if galerkin:
            t = Mb @ q                            # (kblk, k')
            new = np.zeros((fem.kblk, Q.shape[0] + 1, Q.shape[0] + 1))
            new[:, :-1, :-1] = Ahat
            new[:, -1, :-1] = new[:, :-1, -1] = t @ Q.T
            new[:, -1, -1] = t @ q
            Ahat = new
        else:
            qd = ctx.upload(q)
            cq = ctx.alloc(M)
            ctx.gemm_nt(M, 1, kp, R, 0, kp, qd, 0, kp, cq, 0, 1)                        # (R q)
            ctx.gemm_nn(M, kp, 1, cq, 0, 1, qd, 0, kp, R, 0, kp, alpha=-1.0, beta=1.0)  # R -= (R q) q^T
        Q = np.vstack((Q, q))
    return picks, max_errors
