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
