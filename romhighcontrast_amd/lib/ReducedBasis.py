"""Host-side mirror of the reference's ``src/lib/ReducedBasis.py`` driving the HIP kernels.

Same public names (constants, helper functions, ``BaseReducedBasis``, ``ReducedBasisGreedy``,
``ReducedBasisRandom``, ``ReducedBasisPCA``) and the same ``build(n, sm, solutions2train, a2train,
solutions2train_h1norm, **kwargs)`` contract; citations are reference file:line.

The arithmetic runs on the GPU: Euclidean orthonormalisation is a re-orthogonalised Gram-Schmidt
(two passes of MFMA GEMMs per vector), the greedy sweep keeps the training set, the approximations
and the residual norms in HBM, and the PCA is a few passes of thin MFMA products over the block (rom_pod).
Orthonormal bases are defined up to the sign of each vector (NumPy's Householder QR at :19 may
return negative diagonals in R); every consumer in the reference is sign-invariant.
"""
from __future__ import annotations

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
    """Rows of X -> Euclidean-orthonormal rows spanning the same nested subspaces (rom_orthonormalize_rows: device
    CGS2, no host round trip per row)."""
    n, dim = X.rows, X.dim
    n = min(n, dim)   # (more rows than dimensions: the thin QR of the reference, np.linalg.qr(rb.T) at :19, returns dim of them)
    Q = ctx.alloc(max(n * dim, 1))
    if n:
        ctx.orthonormalize_rows(X.buf, n, dim, Q)
    return DeviceArray(Q, n, dim)


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
        U = _as_device(ctx, solutions2train, dim)  # training set stays in HBM for the whole build
        M = U.rows
        fs = getattr(U, "factored", None)
        if fs is None and isinstance(solutions2train, np.ndarray) and hasattr(sm, "factored_of_host_rows"):
            # the reference's own pattern: solutions = sm.generate_solutions(a) -> a host array -> build(n, sm, solutions, ...).
            # The manager kept the interface vectors of the arrays it returned; if this is one of them and nobody has written
            # into it (checked bit for bit on the device), the build runs on them
            fs = sm.factored_of_host_rows(solutions2train, U)
        if fs is not None and fs.M == M and (self.greedy_for == GREEDY_FOR_H10 or kwargs.get("galerkin_on_interface_vectors", True)):
            # A block that sm.generate_solutions_device has just produced carries its interface vectors: the greedy runs on
            # them (rom_greedy_factored: M x ~300 numbers per pass instead of M x dim; same picks; H^1_0 curves within 1e-12 of
            # the row path's and 3e-13 of the reference arithmetic's on the same rows: tests/test_gpu_parity.py, C4).  In
            # Galerkin mode two exact fp64 routes differ by contrast x eps; against the 80-bit truth of the reduced systems
            # (tests/referee.py, contrast 1e8) the factored form -- its quadratic forms w_i^T S_b w_j in compensated
            # arithmetic since round 5 -- is 4.5e-10 off, the reference's own arithmetic 6.5e-10, the row form 1.9e-10:
            # build(..., galerkin_on_interface_vectors=False) keeps the rows (65 ms instead of 8 at C4, n = 50).
            self.picks, self.max_errors = greedy_factored(fs, a2train, n, self.greedy_for == GREEDY_FOR_GALERKIN,
                                                          solutions2train_h1norm)
            basis = ctx.alloc(max(len(self.picks) * dim, 1)).gather_rows_from(U.buf, np.asarray(self.picks), dim)
            super().set(basis=DeviceArray(basis, len(self.picks), dim).numpy(), a=[a2train[i] for i in self.picks])
            return self
        # One C call (rom_greedy): the reference re-orthonormalises the contrast-sorted picks from scratch in every
        # iteration (:135-136) and recomputes the approximations of all snapshots (:122/:124); both depend on the SPAN of
        # the basis only, which the library carries as an A_1-orthonormal basis with the projection residuals updated in
        # place -- picks and error curve are the reference's (tests: fixture g6, composition with the projectors).
        galerkin = self.greedy_for == GREEDY_FOR_GALERKIN
        a_dev = ctx.upload(np.ascontiguousarray(a2train, dtype=np.float64).reshape(M, -1)) if galerkin else None
        picks, self.max_errors = sm._fem.greedy(U.buf, M, a_dev, solutions2train_h1norm, galerkin, n)
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


_pod_warned = set()


def warn_completed_modes(info, n, rel_floor):
    """One warning per process and message when a POD call completed modes the data do not determine (pod_modes and
    pod_modes_factored alike), with the reason the C call reported."""
    floor = max(rel_floor, 1e-13)
    if info["stop_reason"] == "budget":
        msg = (f"POD: {info['completed_modes']} of the {n} requested modes were NOT found although the spectrum had not "
               f"reached the floor ({floor:g} sigma_1): a sketch pass accepted nothing; completed with orthonormal "
               "directions of zero singular value")
    else:
        msg = (f"POD: {info['completed_modes']} of the {n} requested modes lie below the floor of the snapshot block (sigma < "
               f"{floor:g} sigma_1" + ("" if rel_floor > 1e-13 else ": fp64 noise of the data") + "); completed with "
               "orthonormal directions of zero singular value")
    if msg not in _pod_warned:  # (once per process and message: a bench calls this a dozen times on the same block)
        _pod_warned.add(msg)
        warning(msg)


def pod_modes(ctx: _ffi.Context, X: DeviceArray, n: int, center=True, rel_floor=0.0, download=True):
    """Leading ``n`` right singular vectors / singular values of the (M, dim) snapshot block: one C call (rom_pod).

    Randomised range-finder passes over the implicitly deflated block (four thin GEMMs each, one power step, the rows
    orthonormalised on both sides of it: a pass resolves modes over seven orders of magnitude to LAPACK's own bound
    eps sigma_1 / sigma) with a convergence rule per pass; a spectrum that decays too slowly for that -- the first pass
    says so -- gets its leading modes from the MFMA Gram matrix ``G = Xc Xc^T`` instead (eigenpairs iterated in M space,
    modes with lambda_k > 1e-10 lambda_1) and the passes go on below; until the request is filled or the spectrum has
    reached the fp64 noise of the snapshots (1e-13 sigma_1); a Rayleigh-Ritz step over the collected modes orders them.  What is still missing
    then does not exist in the data; like LAPACK / scikit-learn, which return SOME orthonormal directions there, the
    basis is completed with orthonormalised pseudo-random directions (singular value 0, seeded by the number of
    resolved modes: deterministic), so the rows returned are always orthonormal.  Rows follow scikit-learn's
    ``svd_flip(u_based_decision=False)`` sign convention (the PCA call at src/lib/ReducedBasis.py:196).  X is
    overwritten.  ``pod_modes.last_info``: flop accounting and pass counts of the call.  ``rel_floor``: do not look for
    modes below that fraction of sigma_1 (rom_pod_ex; default: the fp64 noise floor of the block, 1e-13).
    ``download=False`` returns the modes as a DeviceArray."""
    M, dim = X.rows, X.dim
    n = min(n, M, dim)
    V = ctx.alloc(max(n * dim, 1))
    X.factored = None  # (the rows are about to be centred in place: they stop being the image of their interface vectors)
    try:
        sig, info = ctx.pod(X.buf, M, dim, n, V, center=center, rel_floor=rel_floor)
    except _ffi.RomLibraryError as e:
        if "NaN / Inf" in str(e) or "rescale the block" in str(e):   # (scikit-learn's PCA raises ValueError on such input)
            raise ValueError(str(e)) from None
        raise
    pod_modes.last_info = info
    pod_modes.resolved = info["resolved_modes"]
    if info["completed_modes"]:
        warn_completed_modes(info, n, rel_floor)
    elif info["stop_reason"] == "budget":
        warning("POD: the eigenpairs of the Gram matrix did not converge (a spectrum so flat that the subspace iteration "
                f"stalls, with more than 1024 snapshots: M = {M}); the trailing modes of the request are approximate")
    if n == 0:
        return (np.zeros((0, dim)) if download else DeviceArray(V, 0, dim)), sig
    return (V.download(n * dim, shape=(n, dim)) if download else DeviceArray(V, n, dim)), sig


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
            # the training block is resident in HBM: the INFINIT_A snapshots are peeled off BY INDEX on the device
            # (get_starting_basis, :153-164, does it on host rows), the pool is gathered into the private copy that rom_pod
            # overwrites, and only the basis rows ever cross PCIe
            Ud, a2 = solutions2train, np.asarray(a2train)
            if getattr(Ud, "factored", None) is not None and Ud.factored.M == Ud.rows:
                # fresh from sm.generate_solutions_device: the block's interface vectors are at hand -- the PCA of the rows IS
                # the PCA of their energy coordinates (rom_pod_factored: same modes, a matrix dim / ~300 times narrower)
                return self.build(n, sm, Ud.factored, a2, solutions2train_h1norm, **kwargs)
            ctx, dim = sm._ctx, Ud.dim
            has_inf = (a2 == INFINIT_A).reshape(len(a2), -1).sum(axis=1) > 0
            pool_idx = np.flatnonzero(~has_inf)
            lead_idx = np.flatnonzero(has_inf) if self.add_inf_solutions else np.zeros(0, dtype=np.int64)
            X = ctx.alloc(max(len(pool_idx) * dim, 1)).gather_rows_from(Ud.buf, pool_idx, dim)
            comps, sigma = pod_modes(ctx, DeviceArray(X, len(pool_idx), dim), n, center=True)
            self.singular_values_ = sigma
            self.resolved_modes_ = pod_modes.resolved
            if lead_idx.size:
                lead = DeviceArray(ctx.alloc(lead_idx.size * dim).gather_rows_from(Ud.buf, lead_idx, dim), lead_idx.size, dim).numpy()
            else:
                lead = np.empty((0, dim))
            super().set(basis=np.vstack((lead, comps))[:n], a=np.vstack((a2[lead_idx].reshape((-1,) + a2.shape[1:]), a2[pool_idx]))[:n])
            warning("PCA method has not been adapted for inverse parameter estimation, the a coefficients are not correct.")
            return self
        basis, a, solutions2train, a2train = get_starting_basis(solutions2train, a2train, self.add_inf_solutions)
        ctx = sm._ctx
        X = _as_device(ctx, np.array(solutions2train, dtype=np.float64), sm.vspace_dim)  # private copy
        comps, sigma = pod_modes(ctx, X, n, center=True)
        self.singular_values_ = sigma
        self.resolved_modes_ = pod_modes.resolved  # modes above the fp64 noise floor of the block (the rest: see pod_modes)
        super().set(basis=np.vstack((basis, comps))[:n], a=np.vstack((a, a2train))[:n])
        warning("PCA method has not been adapted for inverse parameter estimation, the a coefficients are not correct.")
        return self
