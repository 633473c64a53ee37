"""NumPy prototype of closed-form edge elimination + low-rank compression of the active edges (2x2 geometry).
Validates the algebra of DESIGN.md section 9 item 7 against the oracle before any kernel work."""
import sys
import numpy as np
import scipy.linalg
sys.path.insert(0, "."); sys.path.insert(0, "tests/dev")
from oracle import rom_oracle as ro
from proto_schur import Substructure


def run(N, a, tol=1e-18):
    sub = Substructure((2, 2), N)
    g, ub = sub.g, sub.ub
    n1 = N - 1
    a = np.asarray(a, float).reshape(2, 2)
    S = sub.assemble_S(a)          # full interface matrix (order: h(1,0), h(1,1), v(0,1), v(1,1), cross)
    gv = sub.gvec
    h0, h1 = sub.hedge[(1, 0)], sub.hedge[(1, 1)]
    v0, v1 = sub.vedge[(0, 1)], sub.vedge[(1, 1)]
    x = sub.cross[(1, 1)]
    ih0, ih1 = np.arange(h0, h0 + n1), np.arange(h1, h1 + n1)
    iv0, iv1 = np.arange(v0, v0 + n1), np.arange(v1, v1 + n1)
    # K: self block of any edge divided by (a_p + a_q)
    K = S[np.ix_(ih0, ih0)] / (a[0, 0] + a[1, 0])
    assert np.allclose(K, S[np.ix_(iv0, iv0)] / (a[0, 0] + a[0, 1]), atol=1e-13)
    Kinv = np.linalg.inv(K)
    # coupling ranges of v0: towards h0 (block 00), h1 (block 01), cross (end node)
    def basis(iv):
        cols = [S[np.ix_(iv, ih0)], S[np.ix_(iv, ih1)], S[np.ix_(iv, [x])]]
        C = np.hstack([c / max(np.abs(c).max(), 1e-300) for c in cols])
        Q, R, piv = scipy.linalg.qr(C, pivoting=True, mode="economic")
        d = np.abs(np.diag(R))
        r = int((d > tol * d[0]).sum())
        return Q[:, :r]
    out = {}
    for name, iv, s in (("v0", iv0, a[0, 0] + a[0, 1]), ("v1", iv1, a[1, 0] + a[1, 1])):
        W = basis(iv)
        Kt = np.linalg.inv(W.T @ Kinv @ W)
        P = Kinv @ W @ Kt
        out[name] = dict(W=W, Kt=Kt, P=P, s=s, iv=iv)
    # reduced system on z = [z_v0, z_v1, u_x] after eliminating h0, h1 (closed form) and the W-perp parts
    r0, r1 = out["v0"]["W"].shape[1], out["v1"]["W"].shape[1]
    # first: condensed S2 on (v0, v1, x) in nodal coordinates
    keep = np.concatenate([iv0, iv1, [x]])
    elim = np.concatenate([ih0, ih1])
    See = S[np.ix_(elim, elim)]
    S2 = S[np.ix_(keep, keep)] - S[np.ix_(keep, elim)] @ np.linalg.solve(See, S[np.ix_(elim, keep)])
    g2 = gv[keep] - S[np.ix_(keep, elim)] @ np.linalg.solve(See, gv[elim])
    # compressed: change of basis blockdiag(W0, W1, 1) + closed-form elimination of the complements
    n = r0 + r1 + 1
    Z = np.zeros((2 * n1 + 1, n))
    Z[:n1, :r0] = out["v0"]["W"]; Z[n1:2 * n1, r0:r0 + r1] = out["v1"]["W"]; Z[-1, -1] = 1
    # S3 = blockdiag(s K~) - Z^T F Z, F = blockdiag(sK, sK, .) - S2 (everything that is not the edge self block)
    D2 = np.zeros_like(S2)
    D2[:n1, :n1] = out["v0"]["s"] * K; D2[n1:2 * n1, n1:2 * n1] = out["v1"]["s"] * K
    F = D2 - S2
    Dt = np.zeros((n, n))
    Dt[:r0, :r0] = out["v0"]["s"] * out["v0"]["Kt"]; Dt[r0:r0 + r1, r0:r0 + r1] = out["v1"]["s"] * out["v1"]["Kt"]
    S3 = Dt - Z.T @ F @ Z
    # rhs: g~_f = K~ W^T K^-1 g2_f for edges, plain for the cross
    g3 = np.zeros(n)
    g3[:r0] = out["v0"]["Kt"] @ out["v0"]["W"].T @ Kinv @ g2[:n1]
    g3[r0:r0 + r1] = out["v1"]["Kt"] @ out["v1"]["W"].T @ Kinv @ g2[n1:2 * n1]
    g3[-1] = g2[-1]
    z = scipy.linalg.cho_solve(scipy.linalg.cho_factor(S3), g3)
    # recover nodal values of v0, v1
    uG = np.zeros(sub.nG)
    for name, zz, gg in (("v0", z[:r0], g2[:n1]), ("v1", z[r0:r0 + r1], g2[n1:2 * n1])):
        o = out[name]
        p0 = Kinv @ gg - o["P"] @ (o["W"].T @ Kinv @ gg)
        uG[o["iv"]] = o["P"] @ zz + p0 / o["s"]
    uG[x] = z[-1]
    uG[elim] = np.linalg.solve(See, gv[elim] - S[np.ix_(elim, keep)] @ uG[keep])
    ref = np.linalg.solve(S, gv)
    return r0, r1, np.abs(uG - ref).max() / np.abs(ref).max(), np.linalg.cond(S3)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for N in (16, 64, 128):
        for a in (np.ones((2, 2)), 10.0 ** rng.uniform(0, 2, (2, 2)), 10.0 ** rng.uniform(0, 8, (2, 2))):
            for tol in (1e-16,):
                r0, r1, err, cond = run(N, a, tol)
                print(f"N={N} contrast={a.max()/a.min():.1e} ranks ({r0},{r1}) tol={tol:g}: interface rel err {err:.2e}, cond(S3)={cond:.1e}")
