"""Extended-precision referee shared by tests/golden/make_referee.py (fixtures) and the GPU parity tests (rows checked on
the spot).  TEST INFRASTRUCTURE: imports the oracle; nothing in the product path may import this.

    x_0 = SuperLU solve of the oracle's CSC matrix (the reference's call at src/lib/SolutionsManagers.py:31), then
    x_{k+1} = x_k + LU^-1 (b - A x_k)   with the residual evaluated in 80-bit long double in EDGE form
    (A x)_i = sum_j w_ij (x_i - x_j) + (boundary weights) x_i

(differences of neighbouring values first: inside a block that dominates its neighbours they are ~1/c of the values, so the
products carry no cancellation).  The correction contracts by ~kappa * eps per step; iteration stops when it stalls.
"""
import numpy as np
import scipy.sparse.linalg as spla

from oracle import rom_oracle as ro

LD = np.longdouble


def edge_weights(g, a):
    """diag / east / north of the oracle (fp64, exactly what every solver is given) -> edge weights in long double:
    w_e[r,c] couples (r,c)-(r,c+1), w_n[r,c] couples (r,c)-(r+1,c), w_b[r,c] = diag + sum of off-diagonals = the weight
    of the edges to boundary vertices."""
    d, e, n = ro.stencil_arrays(g, a)
    d, e, n = d.astype(LD), e.astype(LD), n.astype(LD)
    wb = d.copy()
    wb[:, :-1] += e
    wb[:, 1:] += e
    wb[:-1, :] += n
    wb[1:, :] += n
    return -e, -n, wb


def residual_ld(g, we, wn, wb, B, x):
    """b - A x in long double, edge form."""
    X = x.reshape(g.nr, g.nc)
    Ax = wb * X
    dh = X[:, :-1] - X[:, 1:]
    Ax[:, :-1] += we * dh
    Ax[:, 1:] -= we * dh
    dv = X[:-1, :] - X[1:, :]
    Ax[:-1, :] += wn * dv
    Ax[1:, :] -= wn * dv
    return (B.astype(LD).reshape(g.nr, g.nc) - Ax).ravel()


def h10_ld(g, v):
    V = v.reshape(g.nr, g.nc)
    s = (V[:, 0] ** 2).sum() + (V[:, -1] ** 2).sum() + (V[0, :] ** 2).sum() + (V[-1, :] ** 2).sum()
    s += ((V[:, :-1] - V[:, 1:]) ** 2).sum() + ((V[:-1, :] - V[1:, :]) ** 2).sum()
    return np.sqrt(s)


def referee(blocks, N, a, max_steps=12, verbose=True, lu=None):
    g = ro.Geometry(blocks, N)
    B = ro.load_vector(g)
    if lu is None:
        lu = spla.splu(ro.assemble_csc(g, a))
    x0 = lu.solve(B)
    we, wn, wb = edge_weights(g, a)
    x = x0.astype(LD)
    hist = []
    for k in range(max_steps):
        r = residual_ld(g, we, wn, wb, B, x)
        dx = lu.solve(np.asarray(r, dtype=np.float64)).astype(LD)
        rel = float(h10_ld(g, dx) / h10_ld(g, x))
        hist.append(rel)
        x = x + dx
        if verbose:
            print(f"  step {k}: |dx|/|x| (H10) = {rel:.3e}", flush=True)
        if rel < 1e-17 or (k > 0 and rel > 0.5 * hist[-2]):
            break
    truth = np.asarray(x, dtype=np.float64)            # nearest fp64 vector to the long-double solution
    err_superlu = float(h10_ld(g, x0.astype(LD) - x) / h10_ld(g, x))
    return g, truth, x0, err_superlu, hist


# ---- the reduced Galerkin systems (C A(a) C^T) c = C B in extended precision ------------------------------------------
def _energy_forms_ld(g, Q):
    """Per coefficient block b: S_b = Q A_b Q^T (n x n, long double), edge form -- sum_edges w (dq_i)(dq_j) + boundary
    terms -- with the oracle's own fp64 stencil weights of the one-hot coefficient (exact small numbers)."""
    n = Q.shape[0]
    Q3 = Q.reshape(n, g.nr, g.nc)
    Dh = (Q3[:, :, :-1] - Q3[:, :, 1:]).reshape(n, -1)
    Dv = (Q3[:, :-1, :] - Q3[:, 1:, :]).reshape(n, -1)
    Qf = Q3.reshape(n, -1)
    forms = []
    for p, q, e in ro._block_onehots(g):
        we, wn, wb = edge_weights(g, e)
        S = np.einsum("ik,jk->ij", Dh * we.ravel(), Dh) + np.einsum("ik,jk->ij", Dv * wn.ravel(), Dv) \
            + np.einsum("ik,jk->ij", Qf * wb.ravel(), Qf)
        forms.append(0.5 * (S + S.T))
    return forms


def _a1_dots_ld(g, X, Y):
    """X A_1 Y^T in long double (edge form; rows of X, Y are FE vectors)."""
    we, wn, wb = edge_weights(g, np.ones((g.nrb, g.ncb)))
    nx, ny = X.shape[0], Y.shape[0]
    X3, Y3 = X.reshape(nx, g.nr, g.nc), Y.reshape(ny, g.nr, g.nc)
    out = np.einsum("ik,jk->ij", ((X3[:, :, :-1] - X3[:, :, 1:]) * we).reshape(nx, -1), (Y3[:, :, :-1] - Y3[:, :, 1:]).reshape(ny, -1))
    out += np.einsum("ik,jk->ij", ((X3[:, :-1, :] - X3[:, 1:, :]) * wn).reshape(nx, -1), (Y3[:, :-1, :] - Y3[:, 1:, :]).reshape(ny, -1))
    out += np.einsum("ik,jk->ij", (X3 * wb).reshape(nx, -1), Y3.reshape(ny, -1))
    return out


def a1_orthonormal_span_ld(g, C, drop=1e-16):
    """Rows of C (fp64) -> A_1-orthonormal long-double basis of their EXACT span, nested in the row order (modified
    Gram-Schmidt, twice); a row whose remainder is below `drop` of its own norm is reported as dependent (kept as zero)."""
    C = np.asarray(C, dtype=np.float64)
    Q = C.astype(LD).copy()
    keep = np.ones(len(Q), dtype=bool)
    for i in range(len(Q)):
        n0 = np.sqrt(_a1_dots_ld(g, Q[i:i + 1], Q[i:i + 1])[0, 0])
        for _ in range(2):
            if i:
                h = _a1_dots_ld(g, Q[i:i + 1], Q[:i])[0]
                Q[i] -= h @ Q[:i]
        n1 = np.sqrt(_a1_dots_ld(g, Q[i:i + 1], Q[i:i + 1])[0, 0])
        if not n1 > drop * n0:
            keep[i] = False
            Q[i] = 0
        else:
            Q[i] /= n1
    return Q, keep


def _chol_solve_batched_ld(A, b):
    """A (M, n, n) SPD, b (n,) or (M, n): long-double Cholesky + substitutions, batched over M."""
    M, n, _ = A.shape
    Lc = np.zeros_like(A)
    for j in range(n):
        d = A[:, j, j] - np.einsum("mk,mk->m", Lc[:, j, :j], Lc[:, j, :j])
        Lc[:, j, j] = np.sqrt(d)
        if j + 1 < n:
            Lc[:, j + 1:, j] = (A[:, j + 1:, j] - np.einsum("mik,mk->mi", Lc[:, j + 1:, :j], Lc[:, j, :j])) / Lc[:, j, j][:, None]
    y = np.zeros((M, n), dtype=LD)
    bb = np.broadcast_to(np.asarray(b, dtype=LD), (M, n))
    for i in range(n):
        y[:, i] = (bb[:, i] - np.einsum("mk,mk->m", Lc[:, i, :i], y[:, :i])) / Lc[:, i, i]
    x = np.zeros((M, n), dtype=LD)
    for i in range(n - 1, -1, -1):
        x[:, i] = (y[:, i] - np.einsum("mk,mk->m", Lc[:, i + 1:, i], x[:, i + 1:])) / Lc[:, i, i]
    return x


def galerkin_truth_nested(g, a, C, U, sizes):
    """Relative H10 errors of the Galerkin ROM (src/lib/SolutionsManagers.py:88-106) on span{C[0], ..., C[j-1]} for every
    j in `sizes`, in 80-bit arithmetic from the fp64 inputs (basis rows C, snapshots U, parameters a): the truth the fp64
    routes to the same numbers -- reference, oracle, GPU rows, GPU factored -- are measured against where they disagree.
    With an A_1-orthonormal long-double basis Q of the exact span, (sum_b a_b S_b) c = Q B has a condition number <= the
    contrast, and  ||u - Q^T c||^2 = (||u||^2 - |p|^2) + |p - c|^2,  p = Q A_1 u  (a projection residual + a sum of squares).
    Returns {j: errors (M,)} as float64."""
    a = np.asarray(a, dtype=np.float64).reshape(len(a), -1)
    U = np.asarray(U, dtype=np.float64)
    Q, keep = a1_orthonormal_span_ld(g, C)
    forms = _energy_forms_ld(g, Q)
    bhat = Q @ ro.load_vector(g).astype(LD)
    P = _a1_dots_ld(g, U.astype(LD), Q)                      # (M, n)
    u2 = np.array([h10_ld(g, u.astype(LD)) ** 2 for u in U])
    aL = a.astype(LD)
    out = {}
    for j in sizes:
        idx = np.flatnonzero(keep[:j])
        A = sum(aL[:, b][:, None, None] * forms[b][np.ix_(idx, idx)][None] for b in range(a.shape[1]))
        c = _chol_solve_batched_ld(A, bhat[idx])
        Pj = P[:, idx]
        e2 = (u2 - np.einsum("mk,mk->m", Pj, Pj)) + np.einsum("mk,mk->m", Pj - c, Pj - c)
        out[j] = np.asarray(np.sqrt(np.maximum(e2, 0) / u2), dtype=np.float64)
    return out
