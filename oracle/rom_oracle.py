"""CPU oracle for the ROMHighContrast hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A NumPy/SciPy restatement of the reference's snapshot-generation + reduced-basis path.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product (``romhighcontrast_amd``) never does.

Parity status: PINNED.  The reference's own tests hold no numerical vector for this path
(SURVEY.md section 4), so the oracle is pinned by golden fixtures generated in the build
container by importing the reference unmodified (``tests/golden/make_golden.py``), and by the
known answers listed in SURVEY.md section 8c.  ``tests/test_oracle_golden.py`` checks every
function below against those fixtures.

All citations are relative to the reference tree (``src/lib/...``).

The reference stores the stiffness operator as a dense tensor ``A_preassembled[nrb,ncb,dim,dim]``
(SolutionsManagers.py:217-218).  That tensor is a 5-point variable-coefficient stencil; the
oracle keeps it as three ``(nr, nc)`` arrays (diag / east / north) and feeds the *same*
third-party solver the reference's ``method="lsqsparse"`` branch calls
(``scipy.sparse.linalg.spsolve``, SolutionsManagers.py:31) with the matrix in CSC form.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg
import scipy.sparse as sp
import scipy.sparse.linalg as spla

INFINIT_A = 1e10  # ReducedBasis.py:11
GREEDY_FOR_H10 = r"$H^1_0$"  # ReducedBasis.py:101
GREEDY_FOR_GALERKIN = "galerkin"  # ReducedBasis.py:102


# --------------------------------------------------------------------------------------
# geometry + assembly  (SolutionsManagers.py:146-219)
# --------------------------------------------------------------------------------------
class Geometry:
    """Sizes the reference derives in ``SolutionsManagerFEM.__init__`` (:147-171)."""

    def __init__(self, blocks_geometry, N):
        nrb, ncb = blocks_geometry
        self.nrb, self.ncb, self.N = int(nrb), int(ncb), int(N)
        self.nr = self.nrb * self.N - 1  # nr_inner_vertices (:154)
        self.nc = self.ncb * self.N - 1  # nc_inner_vertices (:153)
        self.dim = self.nr * self.nc  # (:155)
        self.nr_cells = self.nrb * self.N + 1  # (:157)  (vertex rows, reference naming)
        self.nc_cells = self.ncb * self.N + 1  # (:156)
        self.x_domain = (-self.ncb / 2.0, self.ncb / 2.0)  # (:149)
        self.y_domain = (-self.nrb / 2.0, self.nrb / 2.0)  # (:150)
        self.points_c = np.linspace(*self.x_domain, self.nc_cells)  # (:168)
        self.points_r = np.linspace(*self.y_domain, self.nr_cells)  # (:169)

    @property
    def k(self):
        return self.nrb * self.ncb


def load_vector(g: Geometry) -> np.ndarray:
    """``B_total`` (:177-185): each of the 6 triangles around an inner vertex gives area/6.

    The reference adds area/6, area/3, area/3, area/6 to the four corners of every cell;
    every inner vertex collects 1/6+1/3+1/3+1/6 = 1 cell area = 1/N**2.  We repeat the
    additions in the reference's order so that the value is bit-identical.
    """
    area = (1 / g.N) * (1 / g.N)
    B = np.zeros((g.nr_cells, g.nc_cells))
    # vectorised form of the double loop at :178-184 (each slot receives its four
    # contributions in the same order i,j ascending as the reference)
    B[:-1, :-1] += area / 6
    B[1:, :-1] += area / 3
    B[:-1, 1:] += area / 3
    B[1:, 1:] += area / 6
    return B[1:-1, 1:-1].reshape(g.dim).copy()


def cell_coefficients(g: Geometry, a: np.ndarray) -> np.ndarray:
    """kappa[line, col] = a[line // N, col // N]  (:190-192)."""
    a = np.asarray(a, dtype=np.float64).reshape(g.nrb, g.ncb)
    return np.repeat(np.repeat(a, g.N, axis=0), g.N, axis=1)


def stencil_arrays(g: Geometry, a: np.ndarray):
    """Closed form of the triangle loop ``A(a)`` (:187-215) restricted to inner vertices.

    Returns (diag[nr,nc], east[nr,nc-1], north[nr-1,nc]):
      diag(r,c)  = k[r-1,c-1] + k[r-1,c] + k[r,c-1] + k[r,c]
      east(r,c)  = -(k[r,c] + k[r-1,c]) / 2      coupling (r,c)<->(r,c+1)
      north(r,c) = -(k[r,c] + k[r,c-1]) / 2      coupling (r,c)<->(r+1,c)
    with (r,c) the 1-based vertex-grid coordinates of an inner vertex.  No diagonal
    (SW-NE) coupling survives: the two triangles of a cell contribute +-0 to it.
    """
    k = cell_coefficients(g, a)
    nr, nc = g.nr, g.nc
    # inner vertex (r,c), r=1..nr -> array index r-1
    diag = k[0:nr, 0:nc] + k[0:nr, 1:nc + 1] + k[1:nr + 1, 0:nc] + k[1:nr + 1, 1:nc + 1]
    east = -(k[1:nr + 1, 1:nc] + k[0:nr, 1:nc]) / 2
    north = -(k[1:nr, 1:nc + 1] + k[1:nr, 0:nc]) / 2
    return diag, east, north


def assemble_csc(g: Geometry, a: np.ndarray) -> sp.csc_matrix:
    """Sparse form of ``einsum('pqij,pq->ij', A_preassembled, a)`` (:19-23)."""
    diag, east, north = stencil_arrays(g, a)
    nr, nc = g.nr, g.nc
    idx = np.arange(g.dim).reshape(nr, nc)
    rows = [idx.ravel(), idx[:, :-1].ravel(), idx[:, 1:].ravel(), idx[:-1, :].ravel(), idx[1:, :].ravel()]
    cols = [idx.ravel(), idx[:, 1:].ravel(), idx[:, :-1].ravel(), idx[1:, :].ravel(), idx[:-1, :].ravel()]
    vals = [diag.ravel(), east.ravel(), east.ravel(), north.ravel(), north.ravel()]
    return sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                         shape=(g.dim, g.dim))


def assemble_dense(g: Geometry, a: np.ndarray) -> np.ndarray:
    return assemble_csc(g, a).toarray()


# --------------------------------------------------------------------------------------
# parametric solve  (SolutionsManagers.py:17-40, 64-68)
# --------------------------------------------------------------------------------------
def galerkin_dense(a, B_total, A_preassembled, method="lsq"):
    """Literal restatement of ``galerkin`` (:17-40) for (small) dense tensors.

    Used for the *reduced* systems (``A_preassembled`` of shape (nrb,ncb,n,n)), where the
    reference goes through exactly this function (:105, :136-138).
    """
    A = np.einsum("pqij,pq->ij", A_preassembled, a)
    m = method.lower()
    if m == "lsq":
        return scipy.linalg.solve(A, B_total, assume_a="pos")
    if m == "lsqsparse":
        return spla.spsolve(sp.csc_matrix(A), B_total)
    raise Exception(f"Method {method} Not implemented.")


def solve_one(g: Geometry, a: np.ndarray, B: np.ndarray, method="lsqsparse") -> np.ndarray:
    m = method.lower()
    if m == "lsqsparse":
        return spla.spsolve(assemble_csc(g, a), B)  # (:31)
    if m == "lsq":
        return scipy.linalg.solve(assemble_dense(g, a), B, assume_a="pos")  # (:29)
    raise Exception(f"Method {method} Not implemented.")


def generate_solutions(g: Geometry, a2try, method="lsqsparse") -> np.ndarray:
    """``SolutionsManager.generate_solutions`` (:64-68): map ``galerkin`` over the sweep."""
    B = load_vector(g)
    return np.array([solve_one(g, a, B, method) for a in a2try])


# --------------------------------------------------------------------------------------
# operator application + norms  (SolutionsManagers.py:49, 56-62)
# --------------------------------------------------------------------------------------
def stencil_apply(g: Geometry, a: np.ndarray, X: np.ndarray) -> np.ndarray:
    """Y[k] = A(a) X[k] for X of shape (K, dim), without forming A."""
    diag, east, north = stencil_arrays(g, a)
    X = np.asarray(X, dtype=np.float64).reshape(-1, g.nr, g.nc)
    Y = diag[None] * X
    Y[:, :, :-1] += east[None] * X[:, :, 1:]
    Y[:, :, 1:] += east[None] * X[:, :, :-1]
    Y[:, :-1, :] += north[None] * X[:, 1:, :]
    Y[:, 1:, :] += north[None] * X[:, :-1, :]
    return Y.reshape(-1, g.dim)


def H10norm(g: Geometry, solutions) -> np.ndarray:
    """``sqrt(u^T A_1 u)`` with ``A_1 = sum_pq A_pq`` (:49, :58) = unit-coefficient stencil."""
    U = np.asarray(solutions, dtype=np.float64).reshape(-1, g.dim)
    AU = stencil_apply(g, np.ones((g.nrb, g.ncb)), U)
    return np.sqrt(np.einsum("ki,ki->k", U, AU))


def l2norm(solutions) -> np.ndarray:
    """(:60-62)."""
    return np.sqrt(np.sum(np.square(solutions), axis=1))


# --------------------------------------------------------------------------------------
# projectors  (SolutionsManagers.py:88-139)
# --------------------------------------------------------------------------------------
def _block_onehots(g: Geometry):
    for p in range(g.nrb):
        for q in range(g.ncb):
            e = np.zeros((g.nrb, g.ncb))
            e[p, q] = 1.0
            yield p, q, e


def reduced_tensor(g: Geometry, C: np.ndarray) -> np.ndarray:
    """``A_kl[p,q] = C A_pq C^T`` (:93-101 / :125-133), shape (nrb,ncb,n,n)."""
    C = np.asarray(C, dtype=np.float64).reshape(-1, g.dim)
    n = C.shape[0]
    out = np.zeros((g.nrb, g.ncb, n, n))
    for p, q, e in _block_onehots(g):
        out[p, q] = stencil_apply(g, e, C) @ C.T
    return out


def generate_fm_solutions(g: Geometry, a, C, method="lsq") -> np.ndarray:
    """Galerkin ROM (:88-106)."""
    a = np.asarray(a)
    if len(C) == 0:
        return np.zeros((len(a), g.dim))  # (:89-91)
    C = np.asarray(C, dtype=np.float64).reshape(-1, g.dim)
    A_kl = reduced_tensor(g, C)
    B_k = C @ load_vector(g)  # (:103)
    c = np.array([galerkin_dense(ai, B_k, A_kl, method) for ai in a])  # (:104-105)
    return c @ C  # (:106)


def project_solutions(g: Geometry, solutions, C, method="lsq") -> np.ndarray:
    """H^1_0-orthogonal projection (:108-139)."""
    U = np.asarray(solutions, dtype=np.float64).reshape(-1, g.dim)
    if len(C) == 0:
        return np.zeros((len(U), g.dim))  # (:109-111)
    C = np.asarray(C, dtype=np.float64).reshape(-1, g.dim)
    ones = np.ones((g.nrb, g.ncb))
    # B_km (:113-124): sum over blocks of C A_pq U^T  ==  C A_1 U^T, accumulated per block
    B_km = np.zeros((C.shape[0], U.shape[0]))
    for p, q, e in _block_onehots(g):
        B_km += stencil_apply(g, e, C) @ U.T
    A_kl = reduced_tensor(g, C)
    c = np.array([galerkin_dense(ones, b, A_kl, method) for b in B_km.T])  # (:135-138)
    return c @ C  # (:139)


# --------------------------------------------------------------------------------------
# reduced-basis builders  (ReducedBasis.py)
# --------------------------------------------------------------------------------------
def get_high_contrast_coefficient(a):
    """Largest block coefficient of every parameter (:14-15)."""
    return np.asarray([np.asarray(ai).max() for ai in a], dtype=np.float64)


def orthonormalize_base(rb):
    """Euclidean (not H^1_0) thin QR of the basis vectors, returned as rows (:18-21)."""
    Q = np.linalg.qr(np.asarray(rb, dtype=np.float64).T)[0]
    return Q.T


def sort_orthogonalize_base(a_selected, rb):
    """(:24-29).  The reference permutes the rows by ``argsort(1/a)`` and then permutes the
    already-permuted rows *again* before the QR; the span is unaffected, the row order of the
    orthonormal basis is.  Restated as one composed permutation ``perm[perm]``."""
    a_selected = np.asarray(a_selected, dtype=np.float64)
    perm = np.argsort(1.0 / a_selected)
    twice = perm[perm]
    return a_selected[perm], orthonormalize_base(np.asarray(rb)[twice, :])


def greedy_build(g: Geometry, n, solutions2train, a2train, solutions2train_h1norm,
                 greedy_for=GREEDY_FOR_GALERKIN, method="lsq", return_errors=False):
    """Strong greedy in relative H^1_0 error, ``ReducedBasisGreedy.build`` (:112-139).

    Iteration 0 has an empty basis (approximation = 0, all relative errors = 1, argmax = 0).
    Returns (raw basis rows in pick order, picked parameters, pick indices[, max errors]).
    """
    U = np.asarray(solutions2train, dtype=np.float64)
    a2train = np.asarray(a2train)
    contrast = get_high_contrast_coefficient(a2train)
    if greedy_for not in (GREEDY_FOR_H10, GREEDY_FOR_GALERKIN):
        raise Exception(f"Not implemented greedy for {greedy_for}, "
                        f"should be one of [{GREEDY_FOR_H10}, {GREEDY_FOR_GALERKIN}]")
    picks, errs = [], []
    C_orth = np.empty((0, 0))
    for _ in range(n):
        if greedy_for == GREEDY_FOR_H10:
            approx = project_solutions(g, U, C_orth, method)  # (:122)
        else:
            approx = generate_fm_solutions(g, a2train, C_orth, method)  # (:124)
        rel = H10norm(g, approx - U) / solutions2train_h1norm  # (:129)
        ix = int(np.argmax(rel))
        picks.append(ix)
        errs.append(float(rel[ix]))
        _, C_orth = sort_orthogonalize_base(contrast[picks], U[picks])  # (:135-136)
    basis = U[picks]
    a = [a2train[i] for i in picks]
    return (basis, a, picks, errs) if return_errors else (basis, a, picks)


def split_inf_solutions(solutions2train, a2train, only_one_block=True):
    """``get_inf_solutions_starting_basis`` (:142-150): peel off the snapshots that have
    blocks exactly equal to INFINIT_A (exactly one such block, or any, per the flag)."""
    a2train = np.asarray(a2train)
    solutions2train = np.asarray(solutions2train)
    n_inf = (a2train == INFINIT_A).reshape(len(a2train), -1).sum(axis=1)
    chosen = (n_inf == 1) if only_one_block else (n_inf != 0)
    return solutions2train[chosen], a2train[chosen], solutions2train[~chosen], a2train[~chosen]


def get_starting_basis(solutions2train, a2train, add_inf_solutions=True):
    """(:153-164): both branches drop the INFINIT_A snapshots from the pool; only
    ``add_inf_solutions=True`` keeps them as the fixed leading basis."""
    lead, lead_a, pool, pool_a = split_inf_solutions(solutions2train, a2train, only_one_block=False)
    if not add_inf_solutions:
        lead = np.empty((0, pool.shape[1]))
        lead_a = np.empty((0,) + pool_a.shape[1:])
    return lead, lead_a, pool, pool_a


def pca_components(X: np.ndarray, n: int):
    """Deterministic equivalent of ``PCA(n_components=n).fit(X).components_`` (:196).

    Mean-centred thin SVD; rows = right singular vectors with scikit-learn >= 1.5's ``svd_flip``
    convention (``u_based_decision=False``: the largest-|.| entry of every component row is
    positive).  scikit-learn may pick a randomized solver at large sizes, so at
    those sizes parity is asserted on the subspace / singular values, not on signed vectors.
    Returns (components (n,dim), singular_values (n,)).
    """
    X = np.asarray(X, dtype=np.float64)
    Xc = X - X.mean(axis=0)
    _, S, Vt = np.linalg.svd(Xc, full_matrices=False)
    piv = np.argmax(np.abs(Vt), axis=1)
    signs = np.sign(Vt[np.arange(Vt.shape[0]), piv])
    signs[signs == 0] = 1.0
    Vt = Vt * signs[:, None]
    return Vt[:n], S[:n]


def pca_build(n, solutions2train, a2train, add_inf_solutions=True):
    """``ReducedBasisPCA.build`` (:189-200). Returns (basis (n,dim), a)."""
    basis, a, sol, a2 = get_starting_basis(np.asarray(solutions2train), np.asarray(a2train), add_inf_solutions)
    comps, _ = pca_components(sol, n)
    return np.vstack((basis, comps))[:n], np.vstack((a, a2))[:n]


def random_build(n, solutions2train, a2train, add_inf_solutions=True, seed=42):
    """``ReducedBasisRandom.build`` (:173-180)."""
    basis, a, sol, a2 = get_starting_basis(np.asarray(solutions2train), np.asarray(a2train), add_inf_solutions)
    np.random.seed(seed)
    chosen_ix = np.random.choice(len(sol), size=n, replace=False)
    return np.vstack((basis, sol[chosen_ix]))[:n], np.vstack((a, a2[chosen_ix]))[:n]


# --------------------------------------------------------------------------------------
# point evaluation (SolutionsManagers.py:221-244) -- "next" row (f-1); kept for fixtures
# --------------------------------------------------------------------------------------
def evaluate_solutions(g: Geometry, points, solutions) -> np.ndarray:
    """P1 interpolation on the SW-NE split triangulation (:221-244), vectorised.

    points (m,2) as (x,y); returns (n_solutions, m).  ``searchsorted(...)-1`` is the cell whose
    *right* edge is the first grid line >= the coordinate, exactly as the reference.
    """
    P = np.asarray(points, dtype=np.float64).reshape(-1, 2)
    V = np.zeros((len(np.atleast_2d(solutions)), g.nr_cells, g.nc_cells))
    V[:, 1:-1, 1:-1] = np.asarray(solutions, dtype=np.float64).reshape(-1, g.nr, g.nc)
    ix = np.searchsorted(g.points_c, P[:, 0]) - 1
    iy = np.searchsorted(g.points_r, P[:, 1]) - 1
    tx = (P[:, 0] - g.points_c[ix]) / (g.points_c[ix + 1] - g.points_c[ix])
    ty = (P[:, 1] - g.points_r[iy]) / (g.points_r[iy + 1] - g.points_r[iy])
    v00, v10 = V[:, iy, ix], V[:, iy, ix + 1]          # value at (x-index, y-index)
    v01, v11 = V[:, iy + 1, ix], V[:, iy + 1, ix + 1]
    lower = (1 - tx - ty) * v00 + tx * v10 + ty * v01
    upper = (tx + ty - 1) * v11 + (1 - tx) * v01 + (1 - ty) * v10
    return np.where((tx + ty) < 1, lower, upper)


# --------------------------------------------------------------------------------------
# experiment-level helpers (src/experiments/HighContrast.py:59-64) used by the harness
# --------------------------------------------------------------------------------------
def get_full_a(a_per_block, blocks_geometry, high_contrast_blocks):
    """Expand per-group coefficients to per-block ones; ungrouped blocks stay 1 (:59-64)."""
    a_per_block = np.asarray(a_per_block, dtype=np.float64)
    full = np.ones((a_per_block.shape[0],) + tuple(blocks_geometry))
    for group, members in enumerate(high_contrast_blocks):
        for (p, q) in members:
            full[:, p, q] = a_per_block[:, group]
    return full
