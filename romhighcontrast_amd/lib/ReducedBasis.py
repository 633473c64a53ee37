"""Host-side mirror of the reference's ``src/lib/ReducedBasis.py`` driving the HIP kernels.

Same public names (constants, helper functions, ``BaseReducedBasis``, ``ReducedBasisGreedy``,
``ReducedBasisRandom``, ``ReducedBasisPCA``) and the same ``build(n, sm, solutions2train, a2train,
solutions2train_h1norm, **kwargs)`` contract; citations are reference file:line.

The arithmetic runs on the GPU: Euclidean orthonormalisation is a re-orthogonalised Gram-Schmidt
(two passes of MFMA GEMMs per vector), the greedy sweep keeps the training set, the approximations
and the residual norms in HBM, and the PCA is an MFMA Gram matrix + a small eigenproblem.
Orthonormal bases are defined up to the sign of each vector (NumPy's Householder QR at :19 may
return negative diagonals in R); every consumer in the reference is sign-invariant.
"""
from __future__ import annotations

import functools
import os
from logging import warning
from typing import List

import numpy as np

from .. import _ffi
from .Estimators import EstimatorInv, EstimatorLinear
from .SolutionsManagers import DeviceArray, SolutionsManager, _as_device

INFINIT_A = 1e10  # (:11)

GREEDY_FOR_H10 = r"$H^1_0$"  # (:101)
GREEDY_FOR_GALERKIN = "galerkin"  # (:102)


def get_high_contrast_coefficient(a):
    """(:14-15) largest block coefficient of each parameter."""
    return np.array([np.max(coefs, axis=(-1, -2)) for coefs in a])


def _orthonormalize_device(ctx: _ffi.Context, X: DeviceArray) -> DeviceArray:
    """Rows of X -> Euclidean-orthonormal rows spanning the same nested subspaces (CGS2)."""
    n, dim = X.rows, X.dim
    Q = ctx.alloc(max(n * dim, 1))
    if n == 0:
        return DeviceArray(Q, 0, dim)
    Q.copy_from(X.buf, n * dim)
    h = ctx.alloc(max(n, 1))
    for j in range(n):
        off = j * dim
        if j > 0:
            for _ in range(2):  # "twice is enough"
                ctx.gemm_nt(j, 1, dim, Q, 0, dim, Q, off, dim, h, 0, 1)  # h = Q[:j] v
                ctx.gemm_nn(1, dim, j, h, 0, j, Q, 0, dim, Q, off, dim, alpha=-1.0, beta=1.0)  # v -= h^T Q[:j]
        nrm = float(ctx.l2norm(Q, j, 1, dim)[0])
        if nrm > 1e-300:
            Q.scale(1.0 / nrm, offset=off, n=dim)
        else:
            Q.fill(0.0, offset=off, n=dim)
    return DeviceArray(Q, n, dim)


def _append_orthonormal_row(ctx: _ffi.Context, Q: _ffi.Buffer, k: int, src: _ffi.Buffer, src_row: int, dim: int):
    """Q[k] <- row `src_row` of `src`, orthonormalised against the orthonormal rows Q[0:k] (CGS2, the inner step
    of _orthonormalize_device)."""
    off = k * dim
    Q.copy_from(src, dim, dst_off=off, src_off=src_row * dim)
    if k > 0:
        h = ctx.alloc(k)
        for _ in range(2):  # "twice is enough"
            ctx.gemm_nt(k, 1, dim, Q, 0, dim, Q, off, dim, h, 0, 1)
            ctx.gemm_nn(1, dim, k, h, 0, k, Q, 0, dim, Q, off, dim, alpha=-1.0, beta=1.0)
    nrm = float(ctx.l2norm(Q, k, 1, dim)[0])
    if nrm > 1e-300:
        Q.scale(1.0 / nrm, offset=off, n=dim)
    else:
        Q.fill(0.0, offset=off, n=dim)


def _whiten_rows_device(ctx: _ffi.Context, X: DeviceArray, rel_tol=1e-13) -> DeviceArray:
    """Orthonormal rows spanning the numerical row space of X (which may be rank deficient: sketches of a deflated block
    are): two rounds of {b x b Gram on MFMA, symmetric eigen-decomposition of that tiny matrix on the host,
    X <- Lambda^-1/2 Q^T X as a GEMM}, directions whose singular value is below ``rel_tol`` of the largest are dropped in
    the first round.  Returns r <= b rows.  (CholeskyQR2 with the Cholesky factor replaced by an eigen-decomposition:
    no breakdown on dependent rows, and no fall-back to the row-by-row Gram-Schmidt, which costs a host round trip per
    row.)"""
    b, dim = X.rows, X.dim
    cur, r = X.buf, b
    for rnd in range(2):
        Gb = ctx.alloc(r * r)
        ctx.gram(r, dim, cur, 0, dim, Gb, 0, r)
        Gh = Gb.download(r * r, shape=(r, r))
        lam, Q = np.linalg.eigh(0.5 * (Gh + Gh.T))
        keep = lam > (rel_tol ** 2 if rnd == 0 else 1e-8) * lam[-1]
        lam, Q = lam[keep][::-1], Q[:, keep][:, ::-1]
        rn = int(keep.sum())
        T = np.ascontiguousarray((Q / np.sqrt(lam)).T)       # (rn, r): rows_new = Lambda^-1/2 Q^T rows
        nxt = ctx.alloc(max(rn * dim, 1))
        if rn:
            ctx.gemm_nn(rn, dim, r, ctx.upload(T), 0, r, cur, 0, dim, nxt, 0, dim)
        cur, r = nxt, rn
        if r == 0:
            break
    return DeviceArray(cur, r, dim)


def _cholqr2_device(ctx: _ffi.Context, X: DeviceArray) -> DeviceArray:
    """Orthonormalise rows that are already well conditioned (kappa << 1e7): CholeskyQR2.
    Two rounds of {b x b Gram on MFMA, Cholesky of that tiny matrix on the host, X <- R^-T X as a GEMM}.
    Used for power-iteration bases and lifted POD modes; ill-conditioned snapshot bases go through
    the re-orthogonalised Gram-Schmidt above."""
    b, dim = X.rows, X.dim
    cur = X.buf
    for _ in range(2):
        Gb = ctx.alloc(b * b)
        ctx.gram(b, dim, cur, 0, dim, Gb, 0, b)
        Gh = Gb.download(b * b, shape=(b, b))
        R = np.linalg.cholesky(0.5 * (Gh + Gh.T))          # Gh = R R^T (lower)
        Rinv = np.linalg.inv(R)                              # rows_new = R^-1 rows
        nxt = ctx.alloc(b * dim)
        ctx.gemm_nn(b, dim, b, ctx.upload(np.ascontiguousarray(Rinv)), 0, b, cur, 0, dim, nxt, 0, dim)
        cur = nxt
    return DeviceArray(cur, b, dim)


def orthonormalize_base(rb):
    """(:18-21) Euclidean orthonormalisation of the basis rows (thin QR of ``rb.T``)."""
    if isinstance(rb, DeviceArray):
        return _orthonormalize_device(_ffi.get_context(), rb)
    rb = np.asarray(rb, dtype=np.float64)
    if rb.size == 0:
        return rb.reshape(0, rb.shape[-1] if rb.ndim == 2 else 0)
    ctx = _ffi.get_context()
    return _orthonormalize_device(ctx, _as_device(ctx, rb, rb.shape[-1])).numpy()


def sort_orthogonalize_base(a_selected, rb):
    """(:24-29): order by ``argsort(1/a)``; like the reference the row permutation is applied
    twice before the orthonormalisation."""
    a_selected = np.asarray(a_selected, dtype=np.float64)
    order = np.argsort(1 / a_selected)
    rb = np.asarray(rb)[order, :]
    return a_selected[order], orthonormalize_base(rb[order, :])


class BaseReducedBasis:
    """Container of a reduced basis (rows of ``basis``) and the parameters it came from, with the online
    operations of the reference (:32-98).  Pure host object: picklable, no device state."""

    def __init__(self):
        self.basis = None
        self.a = None
        self.inverse_parameter_estimator = None
        self.linear_parameter_estimator = None

    def build(self, **kwargs):
        raise Exception("Not implemented.")  # (:39-40)

    def set(self, basis, a):
        """(:42-46) install a basis and the two parameter estimators built on its parameters."""
        self.basis, self.a = basis, a
        self.inverse_parameter_estimator = EstimatorInv(a)
        self.linear_parameter_estimator = EstimatorLinear(a)

    # -- sizes -----------------------------------------------------------------------------------------
    @property
    def dim(self):
        return np.shape(self.basis)[0]

    @property
    def ambient_space_dim(self):
        return np.shape(self.basis)[1]

    def __str__(self):
        return type(self).__name__

    def __getitem__(self, item):
        """(:88-92) sub-basis with the same slicing applied to vectors and parameters."""
        sub = BaseReducedBasis()
        sub.set(basis=self.basis[item], a=self.a[item])
        return sub

    # -- online stage: thin wrappers over the solutions manager (GPU) ---------------------------------------
    def forward_modeling(self, sm: SolutionsManager, a: np.ndarray):
        """(:59-60) Galerkin reduced-order solutions for the parameters ``a``."""
        return sm.generate_fm_solutions(a=a, coefficients_rom=self.basis)

    def projection(self, sm: SolutionsManager, true_solutions: np.ndarray):
        """(:62-63) H^1_0-orthogonal projections of ``true_solutions``."""
        return sm.project_solutions(true_solutions, self.basis)

    def state_estimation(self, sm: SolutionsManager, measurement_points: np.ndarray, measurements: np.ndarray,
                         return_coefs=False):
        """(:65-70) fit the basis coefficients to point measurements: the basis is evaluated at the
        measurement points on the device, the (points x n) least-squares problem is solved on the host."""
        E = sm.evaluate_solutions(measurement_points, self.basis)          # (n, points)
        c, *_ = np.linalg.lstsq(E.T, measurements.T, rcond=-1)              # (n, n_measured_states)
        estimates = c.T @ np.array(self.basis)
        return (c, estimates) if return_coefs else estimates

    def parameter_estimation_inverse(self, c):
        """(:72-78) harmonic-mean style estimate from the state-estimation coefficients."""
        return self.inverse_parameter_estimator.estimate_parameter(c_values=c)

    def parameter_estimation_linear(self, c):
        """(:80-86) linear estimate from the state-estimation coefficients."""
        return self.linear_parameter_estimator.estimate_parameter(c_values=c)

    def orthonormalize(self):
        """(:94-98) replace the basis by its contrast-sorted Euclidean-orthonormal version."""
        vectors = np.reshape(self.basis, (-1, self.ambient_space_dim))
        self.basis = sort_orthogonalize_base(get_high_contrast_coefficient(self.a), vectors)[1]


class ReducedBasisGreedy(BaseReducedBasis):
    """Strong greedy in relative H^1_0 error (:105-139)."""

    def __init__(self, greedy_for=GREEDY_FOR_GALERKIN):
        self.greedy_for = greedy_for
        self.name = "Greedy " + self.greedy_for
        self.linestyle = "solid" if greedy_for == GREEDY_FOR_H10 else "dashed"
        super().__init__()

    def build(self, n: int, sm: SolutionsManager, solutions2train, a2train: List[np.ndarray] = (()),
              solutions2train_h1norm=1, **kwargs):
        if self.greedy_for not in (GREEDY_FOR_H10, GREEDY_FOR_GALERKIN):
            raise Exception(f"Not implemented greedy for {self.greedy_for}, "
                            f"should be one of [{GREEDY_FOR_H10}, {GREEDY_FOR_GALERKIN}]")
        from ..factored import FactoredSnapshots, greedy_factored
        if isinstance(solutions2train, FactoredSnapshots):
            # training block in factored form: the same greedy in coordinates of the interface vectors
            a2train = np.asarray(a2train)
            self.picks, self.max_errors = greedy_factored(solutions2train, a2train, n,
                                                          self.greedy_for == GREEDY_FOR_GALERKIN,
                                                          solutions2train_h1norm)
            basis = solutions2train.take(self.picks).rows().numpy()
            super().set(basis=basis, a=[a2train[i] for i in self.picks])
            return self
        ctx = sm._ctx
        dim = sm.vspace_dim
        a2train = np.asarray(a2train)
        high_contrast_a = get_high_contrast_coefficient(a2train)
        U = _as_device(ctx, solutions2train, dim)  # training set stays in HBM for the whole build
        picks: List[int] = []
        self.max_errors = []
        # The reference re-orthonormalises the contrast-sorted picks from scratch in every iteration (:135-136).
        # Both approximations below depend on the SPAN of the basis only, so the orthonormal basis is grown by one
        # vector per iteration instead (re-orthogonalised Gram-Schmidt of the new pick against the rows already
        # there): n instead of n^2/2 vector steps.  ROMHC_GREEDY_RESORT=1 keeps the from-scratch variant (A/B).
        resort = bool(os.environ.get("ROMHC_GREEDY_RESORT"))
        Q = None if resort else ctx.alloc(max(n * dim, 1))
        Ahat = None  # reduced tensor of the rows of Q, grown with them (Galerkin mode)
        for _ in range(n):
            if picks and resort:
                contrast = np.ravel(high_contrast_a[picks])
                order = np.argsort(1 / contrast)
                rows = np.asarray(picks)[order][order]  # the reference permutes twice (:27-28)
                sel = ctx.alloc(len(picks) * dim).gather_rows_from(U.buf, rows, dim)
                C_orth = _orthonormalize_device(ctx, DeviceArray(sel, len(picks), dim))
            elif picks:
                _append_orthonormal_row(ctx, Q, len(picks) - 1, U.buf, picks[-1], dim)
                C_orth = DeviceArray(Q, len(picks), dim)
            else:
                C_orth = np.empty((0, 0))
            if self.greedy_for == GREEDY_FOR_H10:
                approx = sm.project_solutions_device(U, C_orth)  # (:122)
            elif picks and not resort:
                Ahat = sm._reduced_tensor_grow(C_orth, Ahat)
                approx = sm.generate_fm_solutions_device(a2train, C_orth, reduced_tensor=Ahat)  # (:124)
            else:
                approx = sm.generate_fm_solutions_device(a2train, C_orth)  # (:124)
            rel = sm.H10norm_diff(approx, U) / solutions2train_h1norm  # (:129)
            ix = int(np.argmax(rel))
            self.max_errors.append(float(rel[ix]))
            picks.append(ix)
        self.picks = list(picks)
        if isinstance(solutions2train, DeviceArray):
            basis = ctx.alloc(max(len(picks) * dim, 1)).gather_rows_from(U.buf, np.asarray(picks), dim)
            basis = DeviceArray(basis, len(picks), dim).numpy()
        else:
            basis = np.asarray(solutions2train)[picks].reshape(len(picks), -1)
        super().set(basis=basis, a=[a2train[i] for i in picks])  # raw snapshots in pick order (:138)
        return self


def get_inf_solutions_starting_basis(solutions2train, a2train, only_one_block=True):
    """(:142-150) peel off the snapshots that have blocks exactly at INFINIT_A (exactly one such block, or
    any number, per the flag).  Returns (chosen solutions, chosen a, remaining solutions, remaining a)."""
    a2train, solutions2train = np.asarray(a2train), np.asarray(solutions2train)
    n_inf = (a2train == INFINIT_A).reshape(len(a2train), -1).sum(axis=1)
    chosen = (n_inf == 1) if only_one_block else (n_inf > 0)
    return solutions2train[chosen], a2train[chosen], solutions2train[~chosen], a2train[~chosen]


def get_starting_basis(solutions2train, a2train, add_inf_solutions=True):
    """(:153-164) the INFINIT_A snapshots always leave the pool; they lead the basis only on request."""
    lead, lead_a, pool, pool_a = get_inf_solutions_starting_basis(solutions2train, a2train, only_one_block=False)
    if not add_inf_solutions:
        lead, lead_a = np.empty((0, pool.shape[1])), np.empty((0,) + pool_a.shape[1:])
    return lead, lead_a, pool, pool_a


class ReducedBasisRandom(BaseReducedBasis):
    """(:167-180) seeded random snapshots behind the optional INFINIT_A lead -- host indexing only."""

    def __init__(self, add_inf_solutions=True):
        self.add_inf_solutions = add_inf_solutions
        self.name = "Random" + (r" $\infty$" if add_inf_solutions else "")
        super().__init__()

    def build(self, n: int, sm: SolutionsManager, solutions2train, a2train: List[np.ndarray] = (()),
              solutions2train_h1norm=1, seed=42, **kwargs):
        lead, lead_a, pool, pool_a = get_starting_basis(solutions2train, a2train, self.add_inf_solutions)
        np.random.seed(seed)
        picked = np.random.choice(len(pool), size=n, replace=False)
        self.set(basis=np.vstack((lead, pool[picked]))[:n], a=np.vstack((lead_a, pool_a[picked]))[:n])
        return self


@functools.lru_cache(maxsize=8)
def _start_block(b: int, M: int, seed: int) -> np.ndarray:
    """Gaussian start block of the subspace iteration (cached: it only depends on its shape and the seed)."""
    blk = np.random.default_rng(seed).standard_normal((b, M))
    blk.setflags(write=False)
    return blk


def _top_eigenpairs_device(ctx: _ffi.Context, G: _ffi.Buffer, M: int, nev: int, oversample=12, tol=2e-14,
                           max_iter=30, seed=0, accept=1e-13):
    """Leading ``nev`` eigenpairs of the symmetric PSD matrix G (M x M, device) by orthogonal
    (subspace) iteration with Rayleigh-Ritz acceleration.  Every M-sized operation is an MFMA GEMM
    on the device (``Z = Y G``, ``H = Z Y^T``, the rotations, the residuals, the re-orthogonalised
    Gram-Schmidt); the host only sees (nev+p) x (nev+p) projected matrices and nev+p norms.
    Stops when every wanted residual ``||G y_i - theta_i y_i||`` is below ``tol * theta_0`` or has
    stopped decreasing (fp64 floor).
    Returns (theta (nev,), W DeviceArray (nev, M) with orthonormal rows = eigenvectors).
    """
    b = int(min(M, nev + oversample))
    Y = _cholqr2_device(ctx, DeviceArray(ctx.upload(_start_block(b, M, seed)), b, M))  # Gaussian rows: kappa ~ 1
    Z, H = ctx.alloc(b * M), ctx.alloc(b * b)
    Zr, Yr = ctx.alloc(b * M), ctx.alloc(b * M)
    best, stall = np.inf, 0
    for it in range(max_iter):
        ctx.gemm_nt(b, M, M, Y.buf, 0, M, G, 0, M, Z, 0, M)          # Z = Y G   (G symmetric)
        ctx.gemm_nt(b, b, M, Z, 0, M, Y.buf, 0, M, H, 0, b)          # H = Y G Y^T
        Hh = H.download(b * b, shape=(b, b))
        theta, S = np.linalg.eigh(0.5 * (Hh + Hh.T))                  # (b x b) projected problem
        theta, S = theta[::-1], S[:, ::-1]
        St = ctx.upload(np.ascontiguousarray(S.T))
        ctx.gemm_nn(b, M, b, St, 0, b, Y.buf, 0, M, Yr, 0, M)         # Ritz vectors  S^T Y
        if b == M:                                                    # full space: exact after one Ritz step
            return theta[:nev], DeviceArray(Yr, nev, M)
        ctx.gemm_nn(b, M, b, St, 0, b, Z, 0, M, Zr, 0, M)             # G applied to them: S^T (Y G)
        # residuals  Zr - diag(theta) Yr, row norms on the device
        Res = ctx.alloc(b * M).copy_from(Zr, b * M)
        ctx.gemm_nn(b, M, b, ctx.upload(np.diag(-theta)), 0, b, Yr, 0, M, Res, 0, M, alpha=1.0, beta=1.0)
        res = ctx.l2norm(Res, 0, nev, M)
        # pairs whose eigenvalue sits at the fp64 noise floor of G (theta_i < 1e-13 theta_0) cannot be
        # resolved from the Gram matrix at all (pod_modes deflates and retries for those)
        # (only the pairs the caller will accept -- theta_i > accept * theta_0 -- have to converge)
        resolvable = theta[:nev] > accept * abs(theta[0])
        worst = float(res[resolvable].max()) / max(abs(theta[0]), 1e-300) if resolvable.any() else 0.0
        _top_eigenpairs_device.last_iterations = it + 1
        _top_eigenpairs_device.total_iterations = getattr(_top_eigenpairs_device, 'total_iterations', 0) + 1
        if worst < 0.7 * best:
            best, stall = worst, 0
        else:
            stall += 1
        if worst <= tol or stall >= 3 or it == max_iter - 1:
            return theta[:nev], DeviceArray(Yr, nev, M)
        # next basis: the rotated power step G y_i / theta_i for the resolvable pairs (nearly orthonormal
        # rows), the Ritz vector y_i itself where theta_i sits at the noise floor; then CholeskyQR2
        ok = theta > 1e-13 * abs(theta[0])
        Zs = ctx.alloc(b * M)
        ctx.gemm_nn(b, M, b, ctx.upload(np.diag(np.where(ok, 1.0 / np.where(ok, theta, 1.0), 0.0))), 0, b, Zr, 0, M,
                    Zs, 0, M)
        ctx.gemm_nn(b, M, b, ctx.upload(np.diag(np.where(ok, 0.0, 1.0))), 0, b, Yr, 0, M, Zs, 0, M, alpha=1.0, beta=1.0)
        try:
            Y = _cholqr2_device(ctx, DeviceArray(Zs, b, M))
        except np.linalg.LinAlgError:                                # rank-deficient noise directions
            Y = _orthonormalize_device(ctx, DeviceArray(Zs, b, M))
    raise AssertionError("unreachable")


def _transpose_device(ctx: _ffi.Context, T: _ffi.Buffer, rows: int, cols: int) -> _ffi.Buffer:
    """(cols, rows) transpose of the small row-major (rows, cols) matrix T, on the device (an MFMA GEMM against I)."""
    out = ctx.alloc(rows * cols)
    ctx.gemm_nt(cols, rows, cols, ctx.upload(np.eye(cols)), 0, cols, T, 0, cols, out, 0, rows)
    return out


def _orthonormalize_against(ctx: _ffi.Context, V: _ffi.Buffer, found: int, take: int, dim: int):
    """Rows V[found : found + take] <- orthonormal and orthogonal to the orthonormal rows V[0 : found] (block CGS2 against
    the old rows, CholeskyQR2 among the new ones)."""
    new_off = found * dim
    if found:
        C = ctx.alloc(take * found)
        for _ in range(2):
            ctx.gemm_nt(take, found, dim, V, new_off, dim, V, 0, dim, C, 0, found)
            ctx.gemm_nn(take, dim, found, C, 0, found, V, 0, dim, V, new_off, dim, alpha=-1.0, beta=1.0)
    tmp = ctx.alloc(take * dim).copy_from(V, take * dim, src_off=new_off)
    try:
        Q = _cholqr2_device(ctx, DeviceArray(tmp, take, dim))
    except np.linalg.LinAlgError:
        Q = _orthonormalize_device(ctx, DeviceArray(tmp, take, dim))  # (dependent rows: row by row, zero rows for the dependent ones)
    V.copy_from(Q.buf, take * dim, dst_off=new_off)


def _sketched_modes(ctx: _ffi.Context, X: _ffi.Buffer, M: int, dim: int, k: int, oversample=8, power=1, seed=1):
    """Leading ``k`` right singular vectors / singular values of the (M, dim) block X by a randomised range finder with
    power iterations: every big operation is a thin GEMM (2 b M dim flops, b = k + oversample) instead of the
    2 M^2 dim of a Gram matrix.  Used for the DEFLATED remainder of a snapshot block, whose spectrum falls off
    geometrically (the error of the range finder is (sigma_{b+1} / sigma_k)^(2 power + 1)).  The small factor
    T = X Q^T (M x b) goes to the host for a LAPACK SVD -- no Gram matrix of it, so nothing is squared here.
    Returns (V DeviceArray (k', dim) orthonormal rows, sigma (k',)), k' <= k."""
    b = int(min(M, dim, k + oversample))
    Om = ctx.upload(_start_block(b, M, seed))
    Y = ctx.alloc(b * dim)
    ctx.gemm_nn(b, dim, M, Om, 0, M, X, 0, dim, Y, 0, dim)                       # Y = Omega X
    flops = 2.0 * b * M * dim
    for it in range(power + 1):
        Q = _whiten_rows_device(ctx, DeviceArray(Y, b, dim))                      # (the remainder may have rank < b)
        b = Q.rows
        if b == 0:
            _sketched_modes.last_flops = flops
            return DeviceArray(ctx.alloc(1), 0, dim), np.zeros(0)
        T = ctx.alloc(M * b)
        ctx.gemm_nt(M, b, dim, X, 0, dim, Q.buf, 0, dim, T, 0, b)                 # T = X Q^T   (M, b)
        flops += 2.0 * b * M * dim + 4.0 * b * b * dim
        if it == power:
            break
        Tt = _transpose_device(ctx, T, M, b)
        Y = ctx.alloc(b * dim)
        ctx.gemm_nn(b, dim, M, Tt, 0, M, X, 0, dim, Y, 0, dim)                   # Y = T^T X = Q X^T X
        flops += 2.0 * b * M * dim
    # X ~ T Q: the right singular vectors of the small factor rotate Q into the modes
    Th = T.download(M * b, shape=(M, b))
    ss, Rt = np.linalg.svd(Th, full_matrices=False)[1:]
    k = min(k, b)
    V = ctx.alloc(k * dim)
    ctx.gemm_nn(k, dim, b, ctx.upload(np.ascontiguousarray(Rt[:k])), 0, b, Q.buf, 0, dim, V, 0, dim)  # modes = R^T Q
    flops += 2.0 * k * b * dim
    _sketched_modes.last_flops = flops
    return DeviceArray(V, k, dim), ss[:k]


GRAM_ACCEPT = 1e-10    # eigenvalues of a Gram matrix are taken down to this fraction of its largest one ...
SKETCH_ACCEPT = 1e-6   # ... singular values of a sketch down to this fraction of its largest one
NOISE_FLOOR = 1e-13    # modes below this fraction of sigma_1 are fp64 noise of the snapshots themselves


def pod_modes(ctx: _ffi.Context, X: DeviceArray, n: int, center=True, passes=6):
    """Leading ``n`` right singular vectors / singular values of the (M, dim) snapshot block.

    Pass 1: MFMA Gram matrix ``G = Xc Xc^T`` (lower tiles + mirror) -> leading eigenpairs of the M x M matrix by
    subspace iteration on the device -> lift ``V = S^-1 W^T Xc``.  The Gram matrix squares the condition number: its
    eigenvectors carry an error of ~eps (sigma_1 / sigma_k)^2, so only the modes with lambda_k > 1e-8 lambda_1
    (sigma_k > 1e-4 sigma_1) are taken from it.  The rest comes from the DEFLATED block (accepted modes projected out
    of X) -- not through another Gram matrix, which would cost as much as the first, but through a randomised range
    finder on the remainder (``_sketched_modes``: a handful of thin GEMMs and a LAPACK SVD of an M x b factor), each
    round reaching 6 orders of magnitude further down, until the request is filled or the spectrum has reached the
    fp64 noise of the snapshots (1e-13 sigma_1).  A Rayleigh-Ritz step on the collected subspace (SVD of the M x n
    coefficient matrix X V^T, which the deflations have produced on the way) orders the modes and fixes the singular
    values.  What is still missing then does not exist in the data; like LAPACK / scikit-learn, which return SOME
    orthonormal directions there, the basis is completed with orthonormalised random directions (singular value 0),
    so the rows returned are always orthonormal.
    Rows follow scikit-learn's ``svd_flip(u_based_decision=False)`` sign convention (the PCA call at
    src/lib/ReducedBasis.py:196).  X is overwritten (centred, and deflated by all accepted modes but the last batch).  ``pod_modes.last_info`` holds the flop
    accounting of the call (useful = symmetric half of one Gram + lift; executed = what ran).
    """
    M, dim = X.rows, X.dim
    n = min(n, M, dim)
    if center:
        ctx.center_rows(X.buf, M, dim, ctx.alloc(dim))
    V = ctx.alloc(max(n * dim, 1))
    B = ctx.alloc(max(M * n, 1))  # coefficients X V^T of the accepted modes, (M, n) row-major, filled batch by batch
    sig = np.zeros(n)
    found = 0
    executed = 0.0
    info = {"gram_passes": 0, "sketch_passes": 0, "completed_modes": 0}

    def deflate(lo, take, last=False):
        """coefficients of the modes V[lo:lo+take] into B[:, lo:lo+take], and those modes out of X (not when nothing
        reads X afterwards: ``last``)"""
        nonlocal executed
        Y = ctx.alloc(M * take)
        ctx.gemm_nt(M, take, dim, X.buf, 0, dim, V, lo * dim, dim, Y, 0, take)
        if not last:
            ctx.gemm_nn(M, dim, take, Y, 0, take, V, lo * dim, dim, X.buf, 0, dim, alpha=-1.0, beta=1.0)
        # B[:, lo:lo+take] = Y  (strided destination: one small GEMM against the identity)
        ctx.gemm_nn(M, take, take, Y, 0, take, ctx.upload(np.eye(take)), 0, take, B, lo, n)
        executed += (2.0 if last else 4.0) * take * M * dim

    sigma_1 = 0.0
    if n > 0:
        G = ctx.alloc(M * M)
        ctx.gram(M, dim, X.buf, 0, dim, G, 0, M)
        info["gram_passes"] = 1
        pod_modes.last_gram_passes = 1
        executed += float(M) * (M + 1) * dim
        lam, W = _top_eigenpairs_device(ctx, G, M, n, accept=GRAM_ACCEPT)
        del G
        lam = np.maximum(lam, 0.0)
        sigma_1 = float(np.sqrt(lam[0])) if len(lam) else 0.0
        take = 0
        while take < n and take < len(lam) and lam[take] > GRAM_ACCEPT * lam[0] and lam[take] > 0:
            take += 1
        if take:
            s = np.sqrt(lam[:take])
            ctx.rows_scale(W.buf, take, M, 1.0 / s)
            ctx.gemm_nn(take, dim, M, W.buf, 0, M, X.buf, 0, dim, V, 0, dim)   # V = S^-1 W^T Xc
            executed += 2.0 * take * M * dim
            _orthonormalize_against(ctx, V, 0, take, dim)
            deflate(0, take, last=take >= n or passes <= 1)
            found = take
    for p in range(1, passes):
        if found >= n or found == 0:
            break
        Vs, ss = _sketched_modes(ctx, X.buf, M, dim, n - found, seed=p)
        executed += _sketched_modes.last_flops
        info["sketch_passes"] += 1
        take = 0
        while take < len(ss) and found + take < n and ss[take] > SKETCH_ACCEPT * ss[0] and ss[take] > NOISE_FLOOR * sigma_1:
            take += 1
        if take == 0:
            break
        V.copy_from(Vs.buf, take * dim, dst_off=found * dim)
        _orthonormalize_against(ctx, V, found, take, dim)
        at_floor = take < len(ss) and ss[take] <= NOISE_FLOOR * sigma_1
        deflate(found, take, last=at_floor or found + take >= n or p == passes - 1)
        found += take
        if at_floor:
            break  # the spectrum has reached the noise floor: nothing left to find
    if found:
        # Rayleigh-Ritz on the collected subspace: X ~ B V  ->  SVD of B orders / rotates the modes
        Bh = B.download(M * n, shape=(M, n))[:, :found]
        s, Rt = np.linalg.svd(Bh, full_matrices=False)[1:]
        Vr = ctx.alloc(found * dim)
        ctx.gemm_nn(found, dim, found, ctx.upload(np.ascontiguousarray(Rt)), 0, found, V, 0, dim, Vr, 0, dim)
        V.copy_from(Vr, found * dim)
        executed += 2.0 * found * found * dim
        sig[:found] = s
    if found < n:
        # complete the basis: random directions orthonormalised against the modes (CGS2); they carry no variance
        rest = n - found
        # (any linearly independent directions do; uniform float32 draws are 4x cheaper on the host than normal fp64 ones)
        fill = ctx.upload(np.random.default_rng(found).random((rest, dim), dtype=np.float32).astype(np.float64) - 0.5)
        V.copy_from(fill, rest * dim, dst_off=found * dim)
        _orthonormalize_against(ctx, V, found, rest, dim)
        info["completed_modes"] = rest
        warning(f"POD: {rest} of the {n} requested modes lie below the fp64 noise floor of the snapshot block "
                f"(sigma < {NOISE_FLOOR:g} sigma_1); completed with orthonormal directions of zero singular value")
    info.update(resolved_modes=found, executed_flops=executed,
                useful_flops=float(M) * (M + 1) * dim + 2.0 * n * M * dim)
    pod_modes.last_info = info
    pod_modes.resolved = found
    if n == 0:
        return np.zeros((0, dim)), sig
    ctx.rows_sign_flip(V, n, dim)  # svd_flip(u_based_decision=False)
    return V.download(n * dim, shape=(n, dim)), sig


class ReducedBasisPCA(BaseReducedBasis):
    """(:183-200) mean-centred PCA of the training snapshots (INFINIT_A ones peeled off first)."""

    def __init__(self, add_inf_solutions=True):
        self.add_inf_solutions = add_inf_solutions
        self.name = "PCA" + (r" $\infty$" if add_inf_solutions else "")
        super().__init__()

    def build(self, n: int, sm: SolutionsManager, solutions2train, a2train: List[np.ndarray] = (()),
              solutions2train_h1norm=1, add_inf_solutions=True, seed=42, **kwargs):
        from ..factored import FactoredSnapshots, pod_modes_factored
        if isinstance(solutions2train, FactoredSnapshots):
            # the training block in factored form (e.g. gathered from several GPUs): the same peel-off of the
            # INFINIT_A snapshots by index, POD on the interface vectors, only the basis rows are materialised
            fs, a2train = solutions2train, np.asarray(a2train)
            has_inf = (a2train == INFINIT_A).reshape(len(a2train), -1).any(axis=1)
            lead_idx = np.flatnonzero(has_inf) if self.add_inf_solutions else np.zeros(0, dtype=np.int64)
            pool = fs.take(np.flatnonzero(~has_inf))
            comps, sigma = pod_modes_factored(pool, n)
            self.singular_values_ = sigma
            self.resolved_modes_ = pod_modes_factored.last_info.get("resolved_modes", n)
            lead = fs.take(lead_idx).rows().numpy() if lead_idx.size else np.empty((0, sm.vspace_dim))
            super().set(basis=np.vstack((lead, comps))[:n], a=np.vstack((a2train[lead_idx], a2train[~has_inf]))[:n])
            return self
        if isinstance(solutions2train, DeviceArray):
            solutions2train = solutions2train.numpy()
        basis, a, solutions2train, a2train = get_starting_basis(solutions2train, a2train, self.add_inf_solutions)
        ctx = sm._ctx
        X = _as_device(ctx, np.array(solutions2train, dtype=np.float64), sm.vspace_dim)  # private copy
        comps, sigma = pod_modes(ctx, X, n, center=True)
        self.singular_values_ = sigma
        self.resolved_modes_ = pod_modes.resolved  # modes above the fp64 noise floor of the block (the rest: see pod_modes)
        super().set(basis=np.vstack((basis, comps))[:n], a=np.vstack((a, a2train))[:n])
        warning("PCA method has not been adapted for inverse parameter estimation, the a coefficients are not correct.")
        return self
