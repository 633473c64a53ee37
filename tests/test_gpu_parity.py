"""Parity of the HIP path (through the reference-shaped Python API -> ctypes -> C-ABI) against the
committed golden fixtures (reference outputs) and the pinned CPU oracle.  Needs an MI355X.

Tolerances (relative H^1_0 norm of the difference unless stated):
  * snapshots: 1e-11 (observed 1e-15 .. 4e-13; the substructured direct solve and LAPACK/SuperLU
    are both backward stable, kappa(A) <= 1e9 on these inputs);
  * rows with an interior ("floating") block that dominates its neighbours (INFINIT_A = 1e10 in fixture g4, 1e8 in C4
    row 5): the plateau level of that block is only determined to ~contrast * eps * N by ANY fp64 direct solver.  An
    extended-precision referee (tests/golden/make_referee.py: SuperLU + iterative refinement with long-double
    edge-form residuals) gives the truth: at (3,3)/N=11 the reference's lsq is 7.2e-6 and its lsqsparse 1.3e-6 from
    it, at (4,4)/N=8 both are 5e-6 from it while agreeing with each other to 2.6e-7, at C4 row 5 SuperLU is 6.8e-5
    from it -- and the GPU 9.4e-9 (profiles/r03_referee_floating_rows.txt).  Those rows are compared with the TRUTH:
    bound = FLOATING_FACTOR x the worse of the reference's two solvers on the same row; every other row keeps 1e-11;
  * reduced-basis relative errors: |err_gpu - err_ref| <= 1e-10 (BASELINE.json target).
"""
import os

import numpy as np
import pytest

from conftest import load_golden, observed
from oracle import rom_oracle as ro

pytestmark = pytest.mark.gpu

SNAP_TOL = 1e-11
FLOATING_FACTOR = 2.0   # floating rows: GPU-vs-truth <= this x max(reference lsq, lsqsparse vs truth)


@pytest.fixture(scope="module")
def api():
    from src.lib import SolutionsManagers as SM
    from src.lib import ReducedBasis as RB
    return SM, RB


def relh10(g, U, Uref):
    return ro.H10norm(g, np.asarray(U) - Uref) / ro.H10norm(g, Uref)


def test_native_library_is_the_one_running(api):
    from romhighcontrast_amd import _ffi
    ctx = _ffi.get_context()
    assert "gfx950" in ctx.device_name()
    assert open("/proc/self/maps").read().count("libromhc.so") > 0


def test_g1_snapshots_norms_eval(api):
    SM, _ = api
    z = load_golden("g1_basic.npz")
    sm = SM.SolutionsManagerFEM(tuple(z["blocks"]), int(z["N"]))
    g = ro.Geometry(tuple(z["blocks"]), int(z["N"]))
    assert sm.vspace_dim == 361 and str(sm) == "SolutionsManagerFEM"
    assert np.array_equal(sm.B_total, z["B_total"])
    assert np.array_equal(sm.points_c, z["points_c"]) and np.array_equal(sm.points_r, z["points_r"])
    U = sm.generate_solutions(z["a"])
    assert U.shape == z["U"].shape
    observed("g1: snapshots vs reference (rel H10)", relh10(g, U, z["U"]), SNAP_TOL)
    # integer-typed parameters are accepted (InverseProblemPipeline.ipynb cell 17)
    Ui = sm.generate_solutions(np.array([[[1, 2], [3, 4]]]))
    assert relh10(g, Ui, z["U"][2:3]).max() < SNAP_TOL
    np.testing.assert_allclose(sm.H10norm(z["U"]), z["H10"], rtol=1e-13)
    np.testing.assert_allclose(sm.l2norm(z["U"]), z["l2"], rtol=1e-13)
    np.testing.assert_allclose(SM.SolutionsManager.l2norm(z["U"]), z["l2"], rtol=1e-13)
    np.testing.assert_allclose(sm.evaluate_solutions(z["points"], z["U"]), z["evals"], rtol=1e-12, atol=1e-15)
    d, e, n = sm.stencil_arrays(z["a"][2:3])
    assert np.array_equal(d[0], z["diag"]) and np.array_equal(e[0], z["east"]) and np.array_equal(n[0], z["north"])
    # the lazily materialised dense tensor equals the reference's
    A = np.einsum("pqij,pq->ij", sm.A_preassembled, z["a"][2])
    assert np.array_equal(A, ro.assemble_dense(g, z["a"][2]))
    assert np.count_nonzero(sm.A_preassembled4h1_norm) == 1729


def test_g2_config_c1(api):
    SM, _ = api
    z = load_golden("g2_c1.npz")
    g = ro.Geometry(tuple(z["blocks"]), int(z["N"]))
    for method in ("lsq", "lsqsparse"):
        sm = SM.SolutionsManagerFEM(tuple(z["blocks"]), int(z["N"]), method=method)
        U = sm.generate_solutions(z["a"])
        observed(f"g2/C1 method={method}: snapshots vs reference lsq (rel H10)", relh10(g, U, z["U_lsq"]), SNAP_TOL)
        observed(f"g2/C1 method={method}: snapshots vs reference lsqsparse (rel H10)", relh10(g, U, z["U_lsqsparse"]), SNAP_TOL)
    np.testing.assert_allclose(sm.H10norm(U), z["H10"], rtol=1e-12)


@pytest.mark.parametrize("name", ["r23", "r32"])
def test_g3_rectangular(api, name):
    SM, _ = api
    z = load_golden("g3_rect.npz")
    blocks, N = tuple(z[f"{name}_blocks"]), int(z[f"{name}_N"])
    sm = SM.SolutionsManagerFEM(blocks, N)
    g = ro.Geometry(blocks, N)
    observed(f"g3/{name}: snapshots vs reference (rel H10)", relh10(g, sm.generate_solutions(z[f"{name}_a"]), z[f"{name}_U"]), SNAP_TOL)
    np.testing.assert_allclose(sm.H10norm(z[f"{name}_U"]), z[f"{name}_H10"], rtol=1e-13)
    d, e, n = sm.stencil_arrays(z[f"{name}_a"][:1])
    assert np.array_equal(d[0], z[f"{name}_diag"]) and np.array_equal(n[0], z[f"{name}_north"])


@pytest.mark.parametrize("name", ["b22", "b33", "b44"])
def test_g4_high_contrast(api, name):
    SM, _ = api
    z = load_golden("g4_contrast.npz")
    blocks, N = tuple(z[f"{name}_blocks"]), int(z[f"{name}_N"])
    sm = SM.SolutionsManagerFEM(blocks, N)
    g = ro.Geometry(blocks, N)
    U = sm.generate_solutions(z[f"{name}_a"])
    err = relh10(g, U, z[f"{name}_U"])
    self_gap = relh10(g, z[f"{name}_U_lsqsparse"], z[f"{name}_U"])
    assert self_gap[:7].max() < 1e-13  # the reference agrees with itself except on the floating-block row
    floating = self_gap > 1e-9
    # (the message keeps everything needed to read a failure from the log alone: DESIGN.md section 9)
    observed(f"g4/{name}: snapshots vs reference, ordinary rows (rel H10)", err[~floating], SNAP_TOL, detail=(
        f"rows with NaN {np.flatnonzero(np.isnan(U).any(axis=1)).tolist()}, with Inf {np.flatnonzero(np.isinf(U).any(axis=1)).tolist()}, "
        f"all-zero rows {np.flatnonzero(~U.any(axis=1)).tolist()}, max |U| per row {np.abs(U).max(axis=1).tolist()}"))
    if floating.any():
        # the floating row against the extended-precision truth, relative to how close the reference itself gets
        ref = load_golden("referee_g4_floating.npz")
        row = int(ref[f"{name}_row"])
        assert floating.sum() == 1 and floating[row] and np.array_equal(ref[f"{name}_a"], z[f"{name}_a"][row])
        truth = ref[f"{name}_truth"][None]
        e_gpu = float(relh10(g, U[row:row + 1], truth)[0])
        e_ref = max(float(ref[f"{name}_err_ref_lsq_vs_truth"]), float(ref[f"{name}_err_ref_lsqsparse_vs_truth"]))
        observed(f"g4/{name}: floating row {row} vs long-double truth (reference's solvers: {e_ref:.2e})", e_gpu,
                 FLOATING_FACTOR * e_ref)


def test_g5_projectors(api):
    SM, _ = api
    z = load_golden("g5_projectors.npz")
    sm = SM.SolutionsManagerFEM(tuple(z["blocks"]), int(z["N"]))
    scale = np.abs(z["U"]).max()
    for tag in ("0", "1", "5", "10", "snap"):
        C = z["C" + tag] if tag != "snap" else z["Csnap"]
        pk, fk = ("proj" + tag, "fm" + tag) if tag != "snap" else ("proj_snap", "fm_snap")
        observed(f"g5: project_solutions, basis {tag} (abs / max|U|)", np.abs(sm.project_solutions(z["U"], C) - z[pk]) / scale, 1e-11)
        observed(f"g5: generate_fm_solutions, basis {tag} (abs / max|U|)", np.abs(sm.generate_fm_solutions(z["a"], C) - z[fk]) / scale, 1e-11)
    # empty inputs
    assert sm.project_solutions(z["U"], np.empty((0, 0))).shape == z["U"].shape
    assert sm.generate_solutions(np.empty((0, 2, 2))).shape == (0, sm.vspace_dim)


def test_orthonormalize_base_matches_numpy_qr_up_to_sign(api):
    _, RB = api
    z = load_golden("g5_projectors.npz")
    Q = RB.orthonormalize_base(z["U"][:4])
    ref = z["Csnap"]  # np.linalg.qr in the reference
    signs = np.sign(np.sum(Q * ref, axis=1))
    np.testing.assert_allclose(Q * signs[:, None], ref, atol=1e-11)
    np.testing.assert_allclose(Q @ Q.T, np.eye(4), atol=1e-14)


@pytest.mark.parametrize("tag,mode_name", [("h10", "GREEDY_FOR_H10"), ("gal", "GREEDY_FOR_GALERKIN")])
def test_g6_greedy(api, tag, mode_name):
    SM, RB = api
    z = load_golden("g6_greedy.npz")
    sm = SM.SolutionsManagerFEM(tuple(z["blocks"]), int(z["N"]))
    n = int(z["n"])
    h1 = sm.H10norm(z["U"])
    np.testing.assert_allclose(h1, z["h1"], rtol=1e-13)
    rb = RB.ReducedBasisGreedy(greedy_for=getattr(RB, mode_name)).build(
        n=n, sm=sm, solutions2train=z["U"], a2train=z["a"], solutions2train_h1norm=h1)
    assert rb.picks[0] == 0 and rb.max_errors[0] == 1.0
    assert rb.picks == list(z[f"{tag}_picks"])  # max errors are >> roundoff for all 6 picks
    assert np.array_equal(rb.basis, z["U"][rb.picks])
    assert rb.name == "Greedy " + getattr(RB, mode_name)
    for m in range(1, n + 1):
        sub = rb[:m]
        sub.orthonormalize()
        ep = sm.H10norm(sub.projection(sm, z["U"]) - z["U"]) / h1
        ef = sm.H10norm(sub.forward_modeling(sm, z["a"]) - z["U"]) / h1
        # BASELINE: reduced-basis H1 errors within 1e-10 of the reference's
        observed(f"g6/{tag}: projection errors of rb[:{m}] vs reference", np.abs(ep - z[f"{tag}_errs_proj"][m - 1]), 1e-10)
        observed(f"g6/{tag}: Galerkin errors of rb[:{m}] vs reference", np.abs(ef - z[f"{tag}_errs_fm"][m - 1]), 1e-10)
    rb.orthonormalize()
    ref = z[f"{tag}_basis"]
    signs = np.sign(np.sum(rb.basis * ref, axis=1))
    np.testing.assert_allclose(rb.basis * signs[:, None], ref, atol=1e-9)
    with pytest.raises(Exception, match="Not implemented greedy for"):
        RB.ReducedBasisGreedy(greedy_for="x").build(n=1, sm=sm, solutions2train=z["U"], a2train=z["a"])


@pytest.mark.parametrize("mode_name", ["GREEDY_FOR_H10", "GREEDY_FOR_GALERKIN"])
def test_greedy_call_matches_its_definition(api, mode_name):
    """rom_greedy carries the span of the picks as an A_1-orthonormal basis and updates the projection residuals of the
    training block by one vector per iteration; the reference re-orthonormalises the contrast-sorted picks from scratch
    and recomputes every approximation (src/lib/ReducedBasis.py:120-136).  Iteration by iteration, the error vector the
    reference's definition gives -- the standalone projectors on the orthonormalised picks, H10norm of the differences
    -- must have its maximum where the call picked, with the value the call reports."""
    SM, RB = api
    blocks, N, M, n = (2, 3), 12, 60, 10
    sm = SM.SolutionsManagerFEM(blocks, N)
    a = 10.0 ** np.random.default_rng(5).uniform(0, 4, size=(M,) + blocks)
    a[0] = 1.0
    U = sm.generate_solutions(a)
    h1 = sm.H10norm(U)
    mode = getattr(RB, mode_name)
    rb = RB.ReducedBasisGreedy(greedy_for=mode).build(n=n, sm=sm, solutions2train=U, a2train=a, solutions2train_h1norm=h1)
    assert rb.picks[0] == 0 and rb.max_errors[0] == 1.0 and len(set(rb.picks)) == n
    for i in range(1, n):
        sub = rb[:i]
        sub.orthonormalize()                                   # (:135-136)
        approx = sub.projection(sm, U) if mode == RB.GREEDY_FOR_H10 else sub.forward_modeling(sm, a)   # (:122 / :124)
        rel = sm.H10norm(approx - U) / h1                       # (:129)
        observed(f"greedy definition/{mode_name}: |max error - reported| at iteration {i}", abs(rel.max() - rb.max_errors[i]), 1e-10)
        assert rel[rb.picks[i]] >= rel.max() * (1 - 1e-9), (i, rb.picks[i], int(np.argmax(rel)))
    # a normalisation other than the snapshots' own norms: the first pick is the largest ||u|| / h1, not index 0
    h1b = np.ones(M)
    rb1 = RB.ReducedBasisGreedy(greedy_for=mode).build(n=2, sm=sm, solutions2train=U, a2train=a, solutions2train_h1norm=h1b)
    assert rb1.picks[0] == int(np.argmax(h1)) and abs(rb1.max_errors[0] - h1.max()) <= 1e-13 * h1.max()
    # a training set with a duplicated snapshot: once the errors reach roundoff the duplicate may be picked; the build
    # must stay finite and the reduced systems positive definite
    U2, a2 = np.vstack((U[:6], U[2:3])), np.concatenate((a[:6], a[2:3]))
    rb2 = RB.ReducedBasisGreedy(greedy_for=mode).build(n=7, sm=sm, solutions2train=U2, a2train=a2, solutions2train_h1norm=sm.H10norm(U2))
    assert np.all(np.isfinite(rb2.max_errors)) and len(rb2.picks) == 7


def test_small_symmetric_eigensolver_vs_lapack():
    """The one-workgroup Jacobi kernel behind the whitening / Rayleigh-Ritz steps of rom_pod and rom_greedy, against
    numpy.linalg.eigh: random symmetric matrices in LDS (n <= 96) and in the global workspace (n > 96), and a graded
    positive definite matrix whose small eigenvalues must come out to high RELATIVE accuracy."""
    from romhighcontrast_amd import _ffi
    ctx = _ffi.get_context()
    rng = np.random.default_rng(3)
    for n in (1, 2, 5, 61, 62, 96, 97, 130):
        A = rng.standard_normal((n, n))
        A = A + A.T
        lam, T = ctx.small_eig(A, mode=0, gram_like=False)
        ref = np.linalg.eigh(A)[0][::-1]
        observed(f"small_eig n={n}: eigenvalues vs LAPACK (abs / ||A||)", np.abs(lam - ref) / np.abs(ref).max(), 1e-13)  # ~ n eps x sweeps
        observed(f"small_eig n={n}: orthonormality of the eigenvector rows", np.abs(T @ T.T - np.eye(n)), 1e-13)
        observed(f"small_eig n={n}: residual T A T^T - diag (abs / ||A||)", np.abs(T @ A @ T.T - np.diag(lam)) / np.abs(ref).max(), 1e-13)
    # graded: A = D B D with B well conditioned, D over 16 orders of magnitude
    n = 40
    B = rng.standard_normal((n, 3 * n))
    B = B @ B.T / (3 * n)
    d = 10.0 ** np.linspace(0, -8, n)
    A = B * np.outer(d, d)
    lam, T = ctx.small_eig(A, mode=0)
    Bl = B.astype(np.longdouble) * np.outer(d, d).astype(np.longdouble)
    # reference for the small eigenvalues: eigh of the scaled problem is useless (absolute accuracy only), so check the
    # defining relations in long double: Rayleigh quotients of the returned vectors and their relative residuals
    Tl = T.astype(np.longdouble)
    rq = np.einsum("ij,jk,ik->i", Tl, Bl, Tl)
    observed("small_eig graded: eigenvalues vs Rayleigh quotients (relative)", np.abs(rq - lam) / np.abs(rq), 1e-12)
    assert lam.min() > 0 and lam.max() / lam.min() > 1e14
    # whitening and symmetric inverse square root
    X = rng.standard_normal((12, 300))
    X[7] = X[3] + 1e-4 * rng.standard_normal(300)  # nearly dependent row: lambda_min / lambda_max ~ 1e-9 in the Gram matrix
    G = X @ X.T
    _, Tw = ctx.small_eig(G, mode=1, rel_tol=1e-26)
    Q = Tw @ X
    observed("small_eig whitening: (T X)(T X)^T - I (one pass, kappa(G) ~ 1e9)", np.abs(Q @ Q.T - np.eye(12)), 1e-6)
    X[7] = X[3]                                     # exactly dependent: one direction is dropped (a zero row)
    _, Tw = ctx.small_eig(X @ X.T, mode=1, rel_tol=1e-13)
    Q = Tw @ X
    gram = Q @ Q.T
    assert np.abs(gram - np.diag(np.diag(gram))).max() < 1e-10 and sorted(np.round(np.diag(gram), 8))[:1] == [0.0]
    assert np.allclose(sorted(np.diag(gram))[1:], 1.0, atol=1e-10)
    # pivoted-Cholesky whitening (mode 3: what rom_pod's orthonormalisations use): ill-conditioned, rank-deficient and plain blocks
    for b, m, kind in ((24, 500, "decay"), (62, 1024, "decay"), (62, 1024, "plain"), (30, 400, "rank10"), (96, 2000, "plain"), (1, 9, "plain")):
        Xb = rng.standard_normal((b, m))
        if kind == "decay":
            Xb = (10.0 ** -np.linspace(0, 7, b))[:, None] * np.linalg.qr(Xb.T)[0].T + 1e-3 * (10.0 ** -np.linspace(0, 7, b))[:, None] * rng.standard_normal((b, m)) / np.sqrt(m)
            Xb = np.linalg.qr(rng.standard_normal((b, b)))[0] @ Xb     # mixed rows: kappa(X) = 1e7, kappa(Gram) = 1e14
        if kind == "rank10":
            Xb = rng.standard_normal((b, 10)) @ rng.standard_normal((10, m))
        Tc = None
        Zb = Xb
        for rnd in range(2):                                            # two rounds, as the library runs them
            lam_c, Tc = ctx.small_eig(Zb @ Zb.T, mode=3, rel_tol=1e-26 if rnd == 0 else 1e-8)
            Zb = Tc @ Zb
        r = int((np.abs(Zb).max(axis=1) > 0).sum())
        gram = Zb @ Zb.T
        observed(f"pivoted-Cholesky whitening b={b} {kind}: orthonormality of the {r} rows kept (2 rounds)", np.abs(gram[:r, :r] - np.eye(r)), 1e-10)
        assert not Zb[r:].any() and (r == b if kind != "rank10" else r == 10), (b, kind, r)
        # the rows kept span the row space: projecting X onto them loses nothing above the rank threshold
        lost = np.linalg.norm(Xb - (Xb @ Zb[:r].T) @ Zb[:r], axis=1) / np.linalg.norm(Xb, axis=1).max()
        observed(f"pivoted-Cholesky whitening b={b} {kind}: rows of X outside the span (relative to the largest row)", lost, 1e-6 if kind == "decay" else 1e-10)
    # Newton-Schulz path (rows orthogonal up to a moderate defect): a Gaussian block, and rotated orthonormal rows + 5 % noise
    for b, m, noise in ((62, 1024, None), (40, 300, 0.05), (96, 4000, None)):
        Yb = rng.standard_normal((b, m)) if noise is None else np.linalg.qr(rng.standard_normal((m, b)))[0].T + noise * rng.standard_normal((b, m)) / np.sqrt(m)
        for md in (1, 2):
            _, Tn = ctx.small_eig(Yb @ Yb.T, mode=md, rel_tol=1e-30)
            Zb = Tn @ Yb
            observed(f"small_eig Newton-Schulz / fallback b={b} mode={md}: orthonormality", np.abs(Zb @ Zb.T - np.eye(b)), 1e-12)
    Y = rng.standard_normal((9, 200))
    Y = np.linalg.qr(Y.T)[0].T + 1e-3 * rng.standard_normal((9, 200))
    _, Tl_ = ctx.small_eig(Y @ Y.T, mode=2, rel_tol=1e-30)
    Z = Tl_ @ Y
    observed("small_eig Loewdin: orthonormality", np.abs(Z @ Z.T - np.eye(9)), 1e-12)
    assert np.abs(Tl_ - Tl_.T).max() < 1e-13 and np.abs(Z - Y).max() < 2e-2   # symmetric; rows stay close to what they were


def test_g7_pca_random(api):
    SM, RB = api
    z = load_golden("g7_pca_random.npz")
    sm = SM.SolutionsManagerFEM(tuple(z["blocks"]), int(z["N"]))
    n = int(z["n"])
    for flag in (1, 0):
        rb = RB.ReducedBasisPCA(add_inf_solutions=bool(flag)).build(n=n, sm=sm, solutions2train=z["U"], a2train=z["a"])
        ref = z[f"pca_basis_{flag}"]
        assert rb.basis.shape == ref.shape
        np.testing.assert_allclose(rb.basis, ref, atol=2e-9)  # signed modes (sigma_5/sigma_1 ~ 1e-4 here)
        np.testing.assert_allclose(np.array(rb.a), z[f"pca_a_{flag}"])
        np.testing.assert_allclose(rb.singular_values_[:n - (2 if flag else 0)],
                                   z["sigma"][:n - (2 if flag else 0)], rtol=1e-9)
        rr = RB.ReducedBasisRandom(add_inf_solutions=bool(flag)).build(n=n, sm=sm, solutions2train=z["U"], a2train=z["a"])
        assert np.array_equal(rr.basis, z[f"rnd_basis_{flag}"]) and np.array_equal(np.array(rr.a), z[f"rnd_a_{flag}"])


@pytest.mark.parametrize("M,dim", [(1, 7), (5, 300), (37, 1025), (130, 4097)])
def test_row_helpers_of_the_pca(M, dim):
    """Centring (sklearn PCA.fit, src/lib/ReducedBasis.py:196), 1/sigma row scaling and svd_flip(u_based_decision=False)
    on the device against NumPy."""
    from romhighcontrast_amd import _ffi
    ctx = _ffi.get_context()
    rng = np.random.default_rng(M * dim)
    X = rng.standard_normal((M, dim)) + 3.0
    Xd, mean = ctx.upload(X), ctx.alloc(dim)
    ctx.center_rows(Xd, M, dim, mean)
    np.testing.assert_allclose(mean.download(dim), X.mean(axis=0), rtol=0, atol=1e-14)
    Xc = X - X.mean(axis=0)
    np.testing.assert_allclose(Xd.download(shape=(M, dim)), Xc, rtol=0, atol=2e-14)
    fac = rng.uniform(-2, 2, size=M)
    fac[0] = 0.0
    Xd = ctx.upload(X)
    ctx.rows_scale(Xd, M, dim, fac)
    assert np.array_equal(Xd.download(shape=(M, dim)), X * fac[:, None])
    # sign convention: the entry of largest magnitude of every row becomes positive (first one on ties)
    Y = X - 3.0
    if dim > 20:
        Y[0, 5] = -9.0
        Y[0, 17] = 9.0   # tie in magnitude: np.argmax takes index 5 -> the row is negated
    Yd = ctx.upload(Y)
    ctx.rows_sign_flip(Yd, M, dim)
    piv = np.argmax(np.abs(Y), axis=1)
    sg = np.sign(Y[np.arange(M), piv])
    assert np.array_equal(Yd.download(shape=(M, dim)), Y * sg[:, None])
    # only a sub-range of rows
    if M > 2:
        Yd = ctx.upload(Y)
        ctx.rows_sign_flip(Yd, M - 2, dim, row0=1)
        ref = Y.copy()
        ref[1:M - 1] *= sg[1:M - 1, None]
        assert np.array_equal(Yd.download(shape=(M, dim)), ref)


def test_large_downloads_through_pinned_pool():
    """Downloads of a size that keeps coming back land in page-locked arrays from libromhc's pool: same numbers,
    ordinary writable NumPy arrays, a view keeps its block alive, blocks are recycled."""
    import gc
    from romhighcontrast_amd import _ffi
    ctx = _ffi.get_context()
    n = (12 << 20) // 8 + 3
    x = np.random.default_rng(0).standard_normal(n)
    b = ctx.upload(x)
    outs = [b.download() for _ in range(5)]  # the third request on may pin
    for o in outs:
        assert o.dtype == np.float64 and o.flags.writeable and np.array_equal(o, x)
    view = outs[-1][5:1000]
    keep = view.copy()
    del outs, o
    gc.collect()
    again = [b.download(shape=(1, n)) for _ in range(3)]  # reuses pooled blocks; must not disturb the live view
    assert np.array_equal(view, keep) and all(np.array_equal(a[0], x) for a in again)
    again[0][0, :4] = 7.0  # writable, and private to this array
    assert np.array_equal(again[1][0], x)


def test_pod_subspace_iteration_vs_oracle(api):
    """POD with M > n + oversampling, so that the device subspace iteration (not the one-step full-space
    Ritz solve) produces the modes; compared with the oracle's LAPACK SVD."""
    SM, RB = api
    sm = SolutionsManagerFEM_cached(SM, (2, 2), 8)
    M, n = 96, 6
    a = 10.0 ** np.random.default_rng(11).uniform(0, 2, size=(M, 2, 2))
    U = sm.generate_solutions(a)
    rb = RB.ReducedBasisPCA(add_inf_solutions=False).build(n=n, sm=sm, solutions2train=U, a2train=a)
    comps, sigma = ro.pca_components(U, n)
    np.testing.assert_allclose(rb.singular_values_, sigma, rtol=1e-9)
    np.testing.assert_allclose(rb.basis, comps, atol=1e-8)   # sigma_6 / sigma_1 ~ 1e-4: gaps are small
    np.testing.assert_allclose(rb.basis @ rb.basis.T, np.eye(n), atol=1e-13)
    # subspace (projector) agreement, the sign- and rotation-free statement
    P1, P2 = rb.basis.T @ rb.basis, comps.T @ comps
    assert np.abs(P1 - P2).max() < 1e-9


def test_pod_small_modes_vs_lapack(api):
    """The modes below the reach of the Gram matrix (sigma < 1e-4 sigma_1: deflation + sketches + Rayleigh-Ritz in
    pod_modes) against LAPACK's SVD of the same centred rows (the call inside scikit-learn's PCA,
    src/lib/ReducedBasis.py:196): singular values over ten orders of magnitude, the subspaces of the leading modes,
    orthonormal rows throughout -- from snapshot rows and from the factored block."""
    SM, RB = api
    from romhighcontrast_amd import factored
    sm = SM.SolutionsManagerFEM((2, 2), 32)
    ctx, fem, dim = sm._ctx, sm._fem, sm.vspace_dim
    M, n = 200, 30
    a = 10.0 ** np.random.default_rng(4).uniform(0, 2, size=(M, 2, 2))
    Ud = sm.generate_solutions_device(a)
    Uh = Ud.numpy()
    _, sv, Vt = np.linalg.svd(Uh - Uh.mean(axis=0), full_matrices=False)
    X = ctx.alloc(M * dim).copy_from(Ud.buf, M * dim)
    comps, sig = RB.pod_modes(ctx, SM.DeviceArray(X, M, dim), n)
    Yf = ctx.alloc(M * fem.reduced_stride)
    fem.solve_reduced(ctx.upload(a.reshape(M, -1)), M, Yf)
    ctx.solve_status()
    comps_f, sig_f = factored.pod_modes_factored(factored.FactoredSnapshots(sm, Yf, M), n)
    real = sv[:n] > 1e-11 * sv[0]          # what the fp64 snapshots determine
    assert real.sum() >= 20 and sv[:n][real][-1] < 1e-8 * sv[0]     # (the test does reach far below the Gram floor)
    for name, c, s_ in (("rows", comps, sig), ("factored", comps_f, sig_f)):
        np.testing.assert_allclose(s_[real], sv[:n][real], rtol=1e-4, atol=1e-13 * sv[0], err_msg=name)
        lead = sv[:n] > 1e-6 * sv[0]
        np.testing.assert_allclose(s_[lead], sv[:n][lead], rtol=1e-9, err_msg=name)
        assert np.abs(c @ c.T - np.eye(n)).max() < 1e-9, name
        k = int(real.sum())
        P1, P2 = c[:k].T @ c[:k], Vt[:k].T @ Vt[:k]                 # projectors onto the span of the determined modes
        assert np.abs(P1 - P2).max() < 1e-5, (name, np.abs(P1 - P2).max())
        k = int(lead.sum())
        assert np.abs(c[:k].T @ c[:k] - Vt[:k].T @ Vt[:k]).max() < 1e-8, name


def _block_with_known_svd(s, M, dim, seed, noise=1e-15):
    """(M, dim) block with singular values s (then a flat floor at `noise`) and known right singular vectors."""
    rng = np.random.default_rng(seed)
    Q1, _ = np.linalg.qr(rng.standard_normal((M, M)))
    Q2, _ = np.linalg.qr(rng.standard_normal((dim, M)))
    full = np.concatenate([s, noise * rng.uniform(0.3, 1.0, M - len(s))])
    return (Q1 * full) @ Q2.T, Q2[:, :len(s)].T


@pytest.mark.parametrize("per_decade", [1, 3])
def test_pod_passes_resolve_seven_orders_each(api, per_decade):
    """Round 5: rom_pod has no Gram stage on a fast-decaying spectrum -- two sketch passes cover 1 ... 2e-13 sigma_1 (the
    coefficient rows of a pass are orthonormalised in M space before its second product, rom_pod.hip sketch_pass).  A
    block with a KNOWN SVD, one mode per 1 / per_decade decades over 12.7 orders: every singular value, and every mode's
    angle to the true one against LAPACK's own bound eps sigma_1 / sigma (the gaps are of the order of sigma)."""
    SM, RB = api
    from romhighcontrast_amd import _ffi
    ctx = _ffi.get_context()
    s = np.concatenate([10.0 ** -np.arange(0, 12.5, 1.0 / per_decade), [2e-13]])
    n = len(s)
    Xh, Vtrue = _block_with_known_svd(s, 512, 6000, 5)
    comps, sig = RB.pod_modes(ctx, SM.DeviceArray(ctx.upload(Xh), 512, 6000), n, center=False)
    info = RB.pod_modes.last_info
    assert info["gram_passes"] == 0 and info["sketch_passes"] == 2 and info["resolved_modes"] == n, info
    noise = 1.1e-16 / s
    observed(f"POD, {per_decade} mode(s) per decade over 12.7 orders: singular values, |error| / (sigma + 50 eps sigma_1)",
             np.abs(sig - s) / (s + 50 * 1.1e-16), 1e-5)
    observed("  ... singular values above 1e-6 sigma_1 (relative)", np.abs(sig / s - 1)[s > 1e-6], 1e-10)
    ang = np.array([np.linalg.norm(comps[i] - (comps[i] @ Vtrue[i]) * Vtrue[i]) for i in range(n)])
    observed("  ... angle of every mode to the true one / (LAPACK's bound eps sigma_1 / sigma + 1e-14)", ang / (noise + 1e-14), 3.0)
    observed("  ... orthonormality of the rows", np.abs(comps @ comps.T - np.eye(n)), 1e-13)


def test_pod_slow_decay_takes_the_gram_route(api):
    """A spectrum of one decade per 12 modes: the first sketch pass cannot separate its modes from what lies beyond its 32
    rows ((sigma_33 / sigma_k)^3 is not small), the convergence rule of rom_pod_ex says so and the leading modes come from
    the Gram matrix instead -- iterated to convergence in M space.  50 modes against the known SVD."""
    SM, RB = api
    from romhighcontrast_amd import _ffi
    ctx = _ffi.get_context()
    M, dim, n = 512, 6000, 50
    s_all = 10.0 ** (-np.arange(M) / 12.0)
    Xh, Vtrue = _block_with_known_svd(s_all, M, dim, 7)
    comps, sig = RB.pod_modes(ctx, SM.DeviceArray(ctx.upload(Xh), M, dim), n, center=False)
    info = RB.pod_modes.last_info
    assert info["gram_passes"] == 1 and info["resolved_modes"] == n, info
    s = s_all[:n]
    observed("POD, slow decay (a decade per 12 modes), 50 modes: singular values (relative)", np.abs(sig / s - 1), 1e-9)
    ang = np.array([np.linalg.norm(comps[i] - (comps[i] @ Vtrue[i]) * Vtrue[i]) for i in range(n)])
    # (modes lifted from eigenvectors of the Gram matrix, iterated to a residual of 2e-14 lambda_1: the angle is that residual
    # over the gap of the SQUARED values, 0.32 sigma^2 here -- six digits short of LAPACK at the last mode, as in round 4)
    observed("  ... angle of every mode to the true one / (2e-14 (sigma_1 / sigma)^2 / 0.32 + 1e-13)",
             ang / (2e-14 / (0.32 * s ** 2) + 1e-13), 1.0)
    observed("  ... orthonormality of the rows", np.abs(comps @ comps.T - np.eye(n)), 1e-13)


def test_orthonormalize_base_more_rows_than_dimensions(api):
    """(tests/dev/gpu_api_fuzz.py, round 5) The reference's thin QR, np.linalg.qr(rb.T) at src/lib/ReducedBasis.py:19,
    returns min(rows, dim) orthonormal rows; so does the device call."""
    SM, RB = api
    rng = np.random.default_rng(4)
    C = rng.standard_normal((12, 4))
    Q, Qo = RB.orthonormalize_base(C), ro.orthonormalize_base(C)
    assert Q.shape == Qo.shape == (4, 4)
    sgn = np.sign(np.sum(Q * Qo, axis=1))
    assert np.abs(Q * sgn[:, None] - Qo).max() < 1e-13


def test_host_rows_from_generate_solutions_take_the_interface_vector_route(api, monkeypatch):
    """The reference's own pattern -- solutions = sm.generate_solutions(a) (a host array), then
    ReducedBasisGreedy(...).build(n, sm, solutions, a, h1), src/lib/ReducedBasis.py:112 -- runs on the interface vectors the
    manager kept for the array it returned, after a bit-for-bit check on the device that the rows are still their image
    (rom_buf_equal); an array the caller has written into, a copy, or a slice takes the row route."""
    SM, RB = api
    from romhighcontrast_amd import factored
    sm = SM.SolutionsManagerFEM((2, 2), 24)
    if not sm._fem.expansion_is_linear:
        pytest.skip("this geometry does not keep interface vectors")
    a = 10.0 ** np.random.default_rng(8).uniform(0, 2, size=(40, 2, 2))
    U = sm.generate_solutions(a)
    h1 = sm.H10norm(U)
    calls = []
    real = factored.greedy_factored
    monkeypatch.setattr(factored, "greedy_factored", lambda *args, **kw: (calls.append(1), real(*args, **kw))[1])   # (build imports it per call)
    for mode in (RB.GREEDY_FOR_H10, RB.GREEDY_FOR_GALERKIN):
        calls.clear()
        rb_f = RB.ReducedBasisGreedy(mode).build(8, sm, U, a, h1)          # the array generate_solutions returned
        assert len(calls) == 1, mode
        rb_c = RB.ReducedBasisGreedy(mode).build(8, sm, U.copy(), a, h1)   # a copy: not that array
        rb_s = RB.ReducedBasisGreedy(mode).build(8, sm, U[:30], a[:30], h1[:30])   # a slice
        assert len(calls) == 1, mode
        assert rb_f.picks == rb_c.picks
        np.testing.assert_allclose(rb_f.max_errors, rb_c.max_errors, rtol=1e-8, atol=1e-11)
        assert np.array_equal(rb_f.basis, rb_c.basis)
    # written into: the check on the device sees it
    U[3, 5] += 1e-9
    calls.clear()
    rb_w = RB.ReducedBasisGreedy(RB.GREEDY_FOR_H10).build(8, sm, U, a, sm.H10norm(U))
    assert len(calls) == 0
    assert rb_w.picks[0] == 0


def test_a_reported_failure_is_not_reported_again(api):
    """Found by tests/dev/gpu_api_fuzz.py in round 5: a projection onto dependent basis rows raises LinAlgError -- like
    scipy.linalg.solve(assume_a='pos') in the reference (src/lib/SolutionsManagers.py:28) -- and the NEXT call on the
    context, a sweep with perfectly good parameters, raised it again: the device status word was read but not cleared
    (read_status, csrc/rom_basis_int.h).  A failure is reported once."""
    SM, RB = api
    sm = SolutionsManagerFEM_cached(SM, (2, 2), 8)
    a = 10.0 ** np.random.default_rng(2).uniform(0, 2, size=(6, 2, 2))
    U = sm.generate_solutions(a)
    C = np.vstack([U[0], U[1], U[0] + U[1]])          # dependent rows: C A_1 C^T is singular
    for call in (lambda: sm.project_solutions(U, C), lambda: sm.generate_fm_solutions(a, C)):
        with pytest.raises(np.linalg.LinAlgError):
            call()
        U2 = sm.generate_solutions(a)                 # (raised "interface matrix not positive definite" before the fix)
        assert np.array_equal(U2, U)
        P = sm.project_solutions(U, U[:2])
        assert np.all(np.isfinite(P))


def test_pod_says_so_when_the_block_has_no_svd_in_fp64(api):
    """NaN / Inf entries (scikit-learn's PCA, src/lib/ReducedBasis.py:196, raises ValueError on those) and entries whose
    squares leave the range of fp64: rom_pod used to return zeros and 'floor' for them (tools/dev/pod_nan.py, round 5);
    now the call says what is wrong (LAPACK would rescale such a block; this library asks the caller to).  A block of
    zeros is still a block of zeros; blocks scaled by 1e60 / 1e-60 work."""
    SM, RB = api
    from romhighcontrast_amd import _ffi
    ctx = _ffi.get_context()
    rng = np.random.default_rng(0)
    X0 = rng.standard_normal((60, 900))
    for bad in (np.nan, np.inf, -np.inf):
        X = X0.copy()
        X[3, 7] = bad
        with pytest.raises(ValueError, match="NaN / Inf"):
            RB.pod_modes(ctx, SM.DeviceArray(ctx.upload(X), 60, 900), 5)
    for scale in (1e200, 1e-200):
        with pytest.raises(ValueError, match="rescale"):
            RB.pod_modes(ctx, SM.DeviceArray(ctx.upload(X0 * scale), 60, 900), 5, center=False)
    comps, sig = RB.pod_modes(ctx, SM.DeviceArray(ctx.upload(np.zeros((60, 900))), 60, 900), 5)
    assert np.all(sig == 0) and np.abs(comps @ comps.T - np.eye(5)).max() < 1e-13
    sv = np.linalg.svd(X0, compute_uv=False)
    for scale in (1e100, 1e-100):   # (squares fine, fourth powers not: the small eigenproblems would stop rotating)
        with pytest.raises(ValueError, match="rescale"):
            RB.pod_modes(ctx, SM.DeviceArray(ctx.upload(X0 * scale), 60, 900), 5, center=False)
    for scale in (1e60, 1e-60):
        comps, sig = RB.pod_modes(ctx, SM.DeviceArray(ctx.upload(X0 * scale), 60, 900), 5, center=False)
        np.testing.assert_allclose(sig / scale, sv[:5], rtol=1e-9)   # (a flat spectrum: the Gram route, diagonalised whole)
    # the context is usable afterwards
    comps, sig = RB.pod_modes(ctx, SM.DeviceArray(ctx.upload(X0), 60, 900), 5, center=False)
    np.testing.assert_allclose(sig, sv[:5], rtol=1e-9)


def test_pod_awkward_blocks(api):
    """Blocks the sketch passes could stumble over: a mean a million times the variation (the first product of the first
    pass runs on the uncentred block -- its last row is the mean, rom_pod.hip kp_zero_sum_rows), clusters of ten equal
    singular values, rank-deficient data with more modes asked than exist, two and three rows, as many modes as rows."""
    SM, RB = api
    from romhighcontrast_amd import _ffi
    ctx = _ffi.get_context()
    rng = np.random.default_rng(3)
    Q1, _ = np.linalg.qr(rng.standard_normal((300, 300)))
    Q2, _ = np.linalg.qr(rng.standard_normal((4000, 300)))

    def pod(Xh, n, center):
        comps, sig = RB.pod_modes(ctx, SM.DeviceArray(ctx.upload(Xh), *Xh.shape), n, center=center)
        return comps, sig, dict(RB.pod_modes.last_info)

    s = 10.0 ** -np.arange(0, 6, 0.5)
    base = (Q1[:, :len(s)] * s) @ Q2[:, :len(s)].T
    for scale in (1.0, 1e3, 1e6):
        Xh = base + scale * rng.standard_normal(4000)[None, :]
        _, sv, Vt = np.linalg.svd(Xh - Xh.mean(axis=0), full_matrices=False)
        comps, sig, info = pod(Xh, len(s), True)
        # what the stored numbers determine: eps x the size of the UNCENTRED block
        noise = 1.1e-16 * np.linalg.norm(Xh, 2)
        observed(f"POD, mean {scale:g} x the variation: |sigma - LAPACK's| / (1e-12 sigma + 100 eps ||X||_2)",
                 np.abs(sig - sv[:len(s)]) / (1e-12 * sv[:len(s)] + 100 * noise), 1.0)
        k = int((sv[:len(s)] > 1e4 * noise).sum())
        observed("  ... projector onto the modes above 1e4 eps ||X||_2 vs LAPACK", np.abs(comps[:k].T @ comps[:k] - Vt[:k].T @ Vt[:k]), 1e-9)
        observed("  ... orthonormality", np.abs(comps @ comps.T - np.eye(len(s))), 1e-13)
    cl = np.repeat([1.0, 1e-3, 1e-6, 1e-9], 10)
    comps, sig, info = pod((Q1[:, :40] * cl) @ Q2[:, :40].T, 40, False)
    observed("POD, four clusters of ten equal singular values: values (relative)", np.abs(sig / cl - 1), 1e-6)
    for c in range(4):
        Vc = Q2[:, 10 * c:10 * c + 10]
        observed(f"  ... projector onto cluster {c} / (eps sigma_1 / sigma + 1e-15)",
                 np.abs(comps[10 * c:10 * c + 10].T @ comps[10 * c:10 * c + 10] - Vc @ Vc.T) / (1.1e-16 / cl[10 * c] + 1e-15), 1.0)
    comps, sig, info = pod((Q1[:, :5] * [1, .5, .1, .01, .001]) @ Q2[:, :5].T, 10, False)
    assert info["resolved_modes"] == 5 and info["completed_modes"] == 5 and info["stop_reason"] == "floor", info
    observed("POD, rank 5 with 10 modes asked: the five values (relative)", np.abs(sig[:5] / [1, .5, .1, .01, .001] - 1), 1e-12)
    assert np.all(sig[5:] == 0.0)
    observed("  ... orthonormality of resolved + completed rows", np.abs(comps @ comps.T - np.eye(10)), 1e-13)
    for M_, n_, c_ in ((2, 2, True), (3, 3, True), (24, 24, False)):
        Xh = rng.standard_normal((M_, 500))
        sv = np.linalg.svd(Xh - Xh.mean(axis=0) if c_ else Xh, compute_uv=False)
        comps, sig, info = pod(Xh, n_, c_)
        r = M_ - 1 if c_ else M_
        observed(f"POD, {M_} rows, {n_} modes{' (centred)' if c_ else ''}: values vs LAPACK (relative)", np.abs(sig[:r] / sv[:r] - 1), 1e-12)
        observed("  ... orthonormality", np.abs(comps @ comps.T - np.eye(n_)), 1e-13)


def test_pod_fuzz_vs_lapack(api):
    """rom_pod against numpy.linalg.svd on random shapes (2 ... 400 rows, 2 ... 5000 columns), requests (a few modes ... all of
    them), centring and spectra: geometric decay at a random rate, a plateau and a cliff of nine orders, independent random rows
    (a flat spectrum: the Gram route, whose subspace iteration stalls there -- the whole Gram matrix is then diagonalised by
    the grid-wide Jacobi, rom_basis.hip jacobi_grid), exact low rank.  Every singular value within 1e-7 relative + 50 eps ||X||_2
    (what the stored numbers determine) of LAPACK's; a mode the POD completes with sigma = 0 must be below its floor there."""
    SM, RB = api
    from romhighcontrast_amd import _ffi
    ctx = _ffi.get_context()
    rng = np.random.default_rng(20251005)
    worst_v, worst_o = 0.0, 0.0
    for case in range(60):
        M = int(rng.integers(2, 400))
        dim = int(rng.integers(2, 5000))
        r = min(M, dim)
        n = int(rng.integers(1, r + 1)) if rng.random() < 0.3 else int(rng.integers(1, min(r, 60) + 1))
        center = bool(rng.integers(0, 2))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            s = 10.0 ** (-np.arange(r) * rng.uniform(0.02, 1.5))
        elif kind == 1:
            k = int(rng.integers(1, r + 1))
            s = np.concatenate([np.ones(k), 1e-9 * np.ones(r - k)])
        elif kind == 3:
            k = int(rng.integers(1, min(r, 12) + 1))
            s = np.concatenate([10.0 ** -rng.uniform(0, 6, k), np.zeros(r - k)])
        if kind == 2:
            Xh = rng.standard_normal((M, dim))
        else:
            Q1, _ = np.linalg.qr(rng.standard_normal((M, r)))
            Q2, _ = np.linalg.qr(rng.standard_normal((dim, r)))
            Xh = (Q1 * s) @ Q2.T
        if center:
            Xh = Xh + rng.uniform(0, 3) * rng.standard_normal(dim)[None, :]
        sv = np.linalg.svd(Xh - Xh.mean(axis=0) if center else Xh, compute_uv=False)
        comps, sig = RB.pod_modes(ctx, SM.DeviceArray(ctx.upload(Xh), M, dim), n, center=center)
        info = RB.pod_modes.last_info
        noise = 50 * 1.1e-16 * np.linalg.norm(Xh, 2)
        m = min(n, len(sv))
        err = np.abs(sig[:m] - sv[:m]) / (1e-7 * sv[:m] + noise)
        completed_ok = (sig[:m] == 0) & (sv[:m] <= 2e-13 * sv[0] + noise)
        err = np.where(completed_ok, 0.0, err)
        assert err.max() <= 1.0, (case, M, dim, n, center, kind, int(err.argmax()), float(sv[err.argmax()]), float(sig[err.argmax()]), info)
        orth = np.abs(comps @ comps.T - np.eye(n)).max()
        assert orth < 1e-12, (case, M, dim, n, center, kind, orth, info)
        worst_v, worst_o = max(worst_v, float(err.max())), max(worst_o, float(orth))
    observed("POD fuzz, 60 random blocks: |sigma - LAPACK's| / (1e-7 sigma + 50 eps ||X||_2), worst", worst_v, 1.0)
    observed("POD fuzz: orthonormality of the rows, worst", worst_o, 1e-12)


def test_pod_slowly_decaying_spectrum_many_modes(api):
    """a request of hundreds of modes from a spectrum that decays slowly -- 320 modes over 12 orders of
    magnitude, 195 of them below the reach of the Gram matrix -- must be FILLED by the sketch passes (the round-3 loop gave
    up after 12 passes of 16 modes and completed the rest with sigma = 0 under a warning that blamed the data), every
    singular value against LAPACK, and the call must say why it stopped.  The floor of rom_pod_ex cuts the same request
    short on purpose: the modes above it are unchanged, the rest is completed and the stop reason says 'floor'."""
    SM, RB = api
    from romhighcontrast_amd import _ffi
    ctx = _ffi.get_context()
    rng = np.random.default_rng(11)
    M, dim, n = 512, 3000, 320
    Q1, _ = np.linalg.qr(rng.standard_normal((M, M)))
    Q2, _ = np.linalg.qr(rng.standard_normal((dim, M)))
    s = 10.0 ** (-np.arange(M) / 27.0)
    Xh = (Q1 * s) @ Q2.T
    sv = np.linalg.svd(Xh, compute_uv=False)
    X = ctx.upload(Xh)
    comps, sig = RB.pod_modes(ctx, SM.DeviceArray(X, M, dim), n, center=False)
    info = RB.pod_modes.last_info
    assert info["resolved_modes"] == n and info["completed_modes"] == 0 and info["stop_reason"] == "filled", info
    assert info["sketch_passes"] >= 3, info
    # (a singular value is determined to ~50 eps sigma_1 absolutely -- LAPACK's own error and the rounding of the test matrix:
    # 7e-3 of the smallest one requested)
    observed("POD, 320 modes over 12 orders: singular values vs LAPACK (relative, beyond 1e-14 sigma_1)",
             np.maximum(np.abs(sig - sv[:n]) - 1e-14 * sv[0], 0.0) / sv[:n], 1e-3)
    lead = sv[:n] > 1e-6 * sv[0]
    observed("POD, 320 modes: singular values above 1e-6 sigma_1 (relative)", np.abs(sig[lead] / sv[:n][lead] - 1), 1e-9)
    observed("POD, 320 modes: orthonormality of the rows", np.abs(comps @ comps.T - np.eye(n)), 1e-9)
    # the same request with a floor at 1e-8 sigma_1
    X = ctx.upload(Xh)
    comps_f, sig_f = RB.pod_modes(ctx, SM.DeviceArray(X, M, dim), n, center=False, rel_floor=1e-8)
    info_f = RB.pod_modes.last_info
    k = info_f["resolved_modes"]
    above = int((sv > 1e-8 * sv[0]).sum())
    assert info_f["stop_reason"] == "floor" and info_f["completed_modes"] == n - k and above - 2 <= k <= above + 2, (info_f, above)
    assert info_f["sketch_passes"] < info["sketch_passes"]
    # (the modes a pass accepts last are its least converged ones -- the power step weighs a direction with sigma^3 -- and a
    # later pass would have refined them through the final Rayleigh-Ritz step: next to the cut the values are good to 1e-2)
    far = sv[:k] > 1e-6 * sv[0]
    observed("POD with rel_floor = 1e-8: singular values above 1e-6 sigma_1 vs LAPACK (relative)", np.abs(sig_f[:k][far] / sv[:k][far] - 1), 1e-9)
    observed("POD with rel_floor = 1e-8: singular values between the floor and 1e-6 sigma_1 (relative)",
             np.abs(sig_f[:k][~far] / sv[:k][~far] - 1), 5e-2)
    assert np.all(sig_f[k:] == 0.0)
    observed("POD with rel_floor = 1e-8: orthonormality of resolved + completed rows", np.abs(comps_f @ comps_f.T - np.eye(n)), 1e-9)


def test_pca_class_from_a_device_block_stays_on_the_device(api):
    """ReducedBasisPCA.build (src/lib/ReducedBasis.py:189-200) on a DeviceArray at the size of config C3 (8192 x 65 025,
    4.3 GB): the INFINIT_A snapshots are peeled off by index on the device, the pool is gathered into rom_pod's private
    copy, and only basis rows cross PCIe -- the bytes that leave the device through Buffer.download are counted.  The
    basis equals the one from the building blocks (pod_modes on the gathered pool behind the peeled rows)."""
    SM, RB = api
    from romhighcontrast_amd import _ffi
    sm = SM.SolutionsManagerFEM((2, 2), 128)
    ctx, dim = sm._ctx, sm.vspace_dim
    M, n = 8192, 50
    a = 10.0 ** np.random.default_rng(20240807).uniform(0, 2, size=(M, 2, 2))
    inf_rows = [5, 4000, 8191]
    for k, r in enumerate(inf_rows):
        a[r].flat[k] = RB.INFINIT_A
    Uf = sm.generate_solutions_device(a)                     # fresh from the sweep: carries its interface vectors
    assert Uf.factored is not None and Uf.factored.M == M
    Ud = SM.DeviceArray(Uf.buf, M, dim)                      # the same rows as a plain device block: the row route
    before = _ffi.D2H_BYTES[0]
    rb = RB.ReducedBasisPCA(add_inf_solutions=True).build(n, sm, Ud, a)
    moved = _ffi.D2H_BYTES[0] - before
    assert moved <= (n + len(inf_rows) + 2) * dim * 8, f"{moved} bytes left the device; the block is {M * dim * 8}"
    assert rb.basis.shape == (n, dim) and np.array_equal(np.asarray(rb.a[:3]), a[inf_rows])
    # the block that remembers its interface vectors takes the factored route (rom_pod_factored): same peel-off, same modes
    before = _ffi.D2H_BYTES[0]
    rbf = RB.ReducedBasisPCA(add_inf_solutions=True).build(n, sm, Uf, a)
    moved = _ffi.D2H_BYTES[0] - before
    assert moved <= (n + len(inf_rows) + 2) * dim * 8, f"{moved} bytes left the device (factored route)"
    assert np.array_equal(rbf.basis[:3], rb.basis[:3]) and np.array_equal(np.asarray(rbf.a[:3]), a[inf_rows])
    big = rb.singular_values_ > 1e-6 * rb.singular_values_[0]
    observed("PCA class on a fresh device block (factored route) vs the row route: singular values > 1e-6 sigma_1 (relative)",
             np.abs(rbf.singular_values_[big] / rb.singular_values_[big] - 1), 1e-7)
    kb = min(int(big.sum()), n - 3)
    observed("PCA class on a fresh device block (factored route) vs the row route: |<mode, mode>| - 1 for those modes",
             np.abs(np.abs(np.sum(rbf.basis[3:3 + kb] * rb.basis[3:3 + kb], axis=1)) - 1), 1e-6)
    # the same from the building blocks
    pool_idx = np.setdiff1d(np.arange(M), inf_rows)
    X = ctx.alloc(len(pool_idx) * dim).gather_rows_from(Ud.buf, pool_idx, dim)
    comps, sig = RB.pod_modes(ctx, SM.DeviceArray(X, len(pool_idx), dim), n)
    lead = SM.DeviceArray(ctx.alloc(3 * dim).gather_rows_from(Ud.buf, np.array(inf_rows), dim), 3, dim).numpy()
    assert np.array_equal(rb.basis, np.vstack((lead, comps))[:n])
    assert np.array_equal(rb.singular_values_, sig)
    # and the peeled rows are what the reference's own peel-off takes from host rows (a small subsample of the block)
    sub = np.array(sorted(set(inf_rows) | set(range(0, M, 1024))))
    lead_h, lead_a, pool_h, pool_a = RB.get_starting_basis(SM.DeviceArray(ctx.alloc(len(sub) * dim).gather_rows_from(Ud.buf, sub, dim),
                                                                       len(sub), dim).numpy(), a[sub], True)
    assert np.array_equal(lead_h, lead) and np.array_equal(lead_a, a[inf_rows])


def SolutionsManagerFEM_cached(SM, blocks, N):
    return SM.SolutionsManagerFEM(blocks, N)


def test_g9_n32(api):
    SM, _ = api
    z = load_golden("g9_n32.npz")
    sm = SM.SolutionsManagerFEM(tuple(z["blocks"]), int(z["N"]))
    U = sm.generate_solutions(z["a"])
    np.testing.assert_allclose(sm.H10norm(U), z["H10"], rtol=1e-12)
    np.testing.assert_allclose(sm.l2norm(U), z["l2"], rtol=1e-12)
    np.testing.assert_allclose(U.sum(axis=1), z["sums"], rtol=1e-11)
    np.testing.assert_allclose(U[:, z["probe"]], z["U_probe"], rtol=1e-10)


@pytest.mark.parametrize("blocks,N", [((1, 1), 8), ((1, 3), 7), ((2, 2), 2), ((2, 2), 3), ((2, 2), 65), ((2, 2), 70),
                                      ((3, 3), 40), ((5, 2), 9), ((2, 2), 129)])
def test_ragged_geometries_vs_oracle(api, blocks, N):
    """Tile padding edge cases: edges shorter / longer than one 64-tile, one-node edges, strips."""
    SM, _ = api
    sm = SM.SolutionsManagerFEM(blocks, N)
    g = ro.Geometry(blocks, N)
    a = 10.0 ** np.random.default_rng(N).uniform(0, 3, size=(3,) + blocks)
    U = sm.generate_solutions(a)
    assert relh10(g, U, ro.generate_solutions(g, a)).max() < SNAP_TOL


@pytest.mark.parametrize("seed", range(12))
def test_seeded_random_geometries_vs_oracle(api, seed):
    """Geometry fuzz: block grid, resolution, batch size and contrast drawn from a seeded generator -- strips, single
    blocks, one-node edges, grids with several reduced tiles -- snapshots and H10 norms against the SuperLU oracle."""
    SM, _ = api
    rng = np.random.default_rng(1000 + seed)
    blocks = (int(rng.integers(1, 5)), int(rng.integers(1, 5)))
    N = int(rng.integers(2, 49))
    if blocks[0] * blocks[1] * N * N > 12000:  # keep the oracle's sparse solves short
        N = max(2, int((12000 / (blocks[0] * blocks[1])) ** 0.5))
    M = int(rng.integers(1, 40))
    decades = float(rng.uniform(0.5, 4.0))
    a = 10.0 ** rng.uniform(0, decades, size=(M,) + blocks)
    sm = SM.SolutionsManagerFEM(blocks, N)
    g = ro.Geometry(blocks, N)
    U = sm.generate_solutions(a)
    Uo = ro.generate_solutions(g, a)
    tag = f"fuzz {seed}: {blocks[0]}x{blocks[1]} N={N} M={M} contrast 1e{decades:.1f}"
    observed(f"{tag}: snapshots vs oracle (rel H10)", relh10(g, U, Uo), SNAP_TOL)
    observed(f"{tag}: H10 norms vs oracle (relative)", np.abs(sm.H10norm(U) / ro.H10norm(g, Uo) - 1.0), SNAP_TOL)


def test_workspace_chunking_and_streams(api):
    """A small factor-workspace budget forces the sweep through several chunks (and the multi-stream
    sub-batch path); results must be bit-identical to the single-chunk run."""
    SM, _ = api
    sm = SM.SolutionsManagerFEM((2, 2), 20)
    a = 10.0 ** np.random.default_rng(2).uniform(0, 3, size=(300, 2, 2))
    ref = sm.generate_solutions(a)
    ctx = sm._ctx
    per_sys = 2 * 4096 * 8 + 2 * 640 * 8  # factor bytes of one system at this size (one tile + inverse + vectors)
    ctx.set_workspace_limit(70 * per_sys)
    try:
        sm2 = SM.SolutionsManagerFEM((2, 2), 20)
        assert np.array_equal(sm2.generate_solutions(a), ref)
    finally:
        ctx.set_workspace_limit(24 << 30)


@pytest.mark.parametrize("blocks,N", [((1, 2), 6), ((2, 1), 9), ((1, 4), 5), ((3, 3), 20), ((4, 4), 6), ((2, 2), 64)])
def test_closed_form_edge_elimination_geometries(api, blocks, N):
    """Geometries that exercise the closed-form elimination: every edge eliminated ((1,2)), chains,
    cross points whose hosting edge is not a block neighbour of the eliminated edge (3x3, 4x4)."""
    SM, _ = api
    sm = SM.SolutionsManagerFEM(blocks, N)
    g = ro.Geometry(blocks, N)
    a = 10.0 ** np.random.default_rng(N + blocks[0]).uniform(0, 4, size=(4,) + blocks)
    assert relh10(g, sm.generate_solutions(a), ro.generate_solutions(g, a)).max() < SNAP_TOL


def test_error_behaviour(api):
    SM, _ = api
    from scipy.linalg import LinAlgError
    sm = SM.SolutionsManagerFEM((2, 2), 4, method="ridge")
    with pytest.raises(Exception, match="Method ridge Not implemented."):
        sm.generate_solutions(np.ones((1, 2, 2)))
    sm = SM.SolutionsManagerFEM((2, 2), 4)
    with pytest.raises(LinAlgError):  # scipy posv raises LinAlgError on a non-SPD matrix
        sm.generate_solutions(np.array([[[1.0, -1.0], [1.0, 1.0]]]))
    with pytest.raises(Exception, match="Not implemented."):
        sm.generate_riesz([[0.0, 0.0]], norm="h10")
    assert sm.generate_riesz([[0.1, 0.2], [0.3, -0.4]], norm="l2").shape == (2, sm.vspace_dim)  # (m, N) as the reference docstring says
    np.testing.assert_allclose(SM.galerkin(np.ones((1, 1)), np.array([1.0, 2.0]), np.array([[[[2.0, 0.0], [0.0, 4.0]]]])),
                               [0.5, 0.5])


def test_basis_stage_calls_edge_cases(api):
    """The single-call operations of the basis stage (rom_project_h10, rom_galerkin_rom, rom_greedy, rom_pod,
    rom_orthonormalize_rows) on the inputs the reference's loops meet at their edges: empty bases and sweeps, one snapshot,
    more modes / basis vectors requested than there are snapshots, dependent basis rows, bad arguments."""
    SM, RB = api
    from scipy.linalg import LinAlgError
    from romhighcontrast_amd import _ffi
    sm = SM.SolutionsManagerFEM((2, 2), 6)
    ctx, fem, dim = sm._ctx, sm._fem, sm.vspace_dim
    g = ro.Geometry((2, 2), 6)
    M = 9
    a = 10.0 ** np.random.default_rng(2).uniform(0, 3, size=(M, 2, 2))
    U = sm.generate_solutions(a)
    h1 = sm.H10norm(U)
    # projectors: empty basis -> zeros (src/lib/SolutionsManagers.py:89-91,109-111); empty sweep -> (0, dim)
    assert not sm.project_solutions(U, np.empty((0, 0))).any() and not sm.generate_fm_solutions(a, np.empty((0, 0))).any()
    assert sm.project_solutions(np.empty((0, dim)), U[:2]).shape == (0, dim)
    assert sm.generate_fm_solutions(np.empty((0, 2, 2)), U[:2]).shape == (0, dim)
    # a basis that contains the snapshot reproduces it; against the oracle for a generic basis
    C = ro.orthonormalize_base(U[:4])
    observed("edge cases: projection of a basis member onto its basis (rel H10)", relh10(g, sm.project_solutions(U[:4], C), U[:4]), 1e-11)
    observed("edge cases: project_solutions vs oracle (abs / max|U|)", np.abs(sm.project_solutions(U, C) - ro.project_solutions(g, U, C)) / np.abs(U).max(), 1e-12)
    observed("edge cases: generate_fm_solutions vs oracle (abs / max|U|)", np.abs(sm.generate_fm_solutions(a, C) - ro.generate_fm_solutions(g, a, C)) / np.abs(U).max(), 1e-12)
    # dependent basis rows: the reference's posv raises LinAlgError on the singular reduced matrix
    with pytest.raises(LinAlgError):
        sm.project_solutions(U, np.vstack((U[:2], U[:1])))
    # greedy: one snapshot; more vectors than snapshots (duplicates at roundoff, finite errors); h1norm broadcast from a scalar
    for mode in (RB.GREEDY_FOR_H10, RB.GREEDY_FOR_GALERKIN):
        rb = RB.ReducedBasisGreedy(mode).build(1, sm, U[:1], a[:1], h1[:1])
        assert rb.picks == [0] and rb.max_errors == [1.0] and np.array_equal(rb.basis, U[:1])
        rb = RB.ReducedBasisGreedy(mode).build(M + 2, sm, U, a, h1)
        assert len(rb.picks) == M + 2 and sorted(set(rb.picks[:4])) == sorted(rb.picks[:4]) and np.all(np.isfinite(rb.max_errors))
        assert max(rb.max_errors[M:]) < 1e-6      # the nine snapshots span themselves
        rb = RB.ReducedBasisGreedy(mode).build(3, sm, U, a, 1)
        assert rb.picks[0] == int(np.argmax(h1))
    # POD: more modes than the data determine -> completed, orthonormal; one snapshot; uncentred; sigma vs LAPACK
    X = ctx.upload(U)
    comps, sig = RB.pod_modes(ctx, SM.DeviceArray(X, M, dim), M)
    sv = np.linalg.svd(U - U.mean(axis=0), compute_uv=False)
    observed("edge cases: POD singular values of 9 snapshots vs LAPACK (relative to sigma_1)", np.abs(sig - np.where(sv > 1e-13 * sv[0], sv, 0.0))[:M - 1] / sv[0], 1e-12)
    observed("edge cases: POD with as many modes as snapshots, orthonormality", np.abs(comps @ comps.T - np.eye(M)), 1e-12)
    comps1, sig1 = RB.pod_modes(ctx, SM.DeviceArray(ctx.upload(U[:1]), 1, dim), 1)
    assert comps1.shape == (1, dim) and sig1[0] == 0.0 and abs(np.linalg.norm(comps1) - 1) < 1e-12     # centred single row: no variance
    comps2, sig2 = RB.pod_modes(ctx, SM.DeviceArray(ctx.upload(U), M, dim), 2, center=False)
    sv2, Vt2 = np.linalg.svd(U, full_matrices=False)[1:]
    observed("edge cases: uncentred POD, leading singular values vs LAPACK (relative)", np.abs(sig2 - sv2[:2]) / sv2[:2], 1e-10)
    # orthonormalize: zero rows in, zero rows out; nothing at all
    Z = np.vstack((U[:2], np.zeros((1, dim))))
    Q = RB.orthonormalize_base(Z)
    assert not Q[2].any() and np.abs(Q[:2] @ Q[:2].T - np.eye(2)).max() < 1e-14
    assert RB.orthonormalize_base(np.empty((0, dim))).shape == (0, dim)
    # the C entries refuse bad arguments with a message instead of faulting
    lib = ctx.lib
    assert lib.rom_pod(ctx.h, X.h, 0, M, dim, M + 1, 1, X.h, 0, None, None) != 0 and b"rom_pod" in lib.rom_last_error()
    assert lib.rom_greedy(fem.h, X.h, 0, M, None, h1.ctypes.data, 1, 2, None, None) != 0    # Galerkin mode without parameters
    assert lib.rom_greedy(fem.h, X.h, 0, M + 5, None, h1.ctypes.data, 0, 2, h1.ctypes.data, h1.ctypes.data) != 0   # rows out of range
    assert lib.rom_project_h10(fem.h, X.h, 0, M, X.h, 5, M, X.h, 0) != 0                        # basis rows out of range


@pytest.mark.parametrize("n", [1, 64, 88, 89, 100, 140, 141, 150, 260])
def test_reduced_solves_of_any_size(api, n):
    """galerkin() takes any n in the reference (src/lib/SolutionsManagers.py:17-40): the batched reduced solve keeps the
    matrix in LDS up to n = 140 (64 KB default up to 88, the CU's 160 KB beyond) and in global memory after that."""
    SM, _ = api
    from romhighcontrast_amd import _ffi
    ctx = _ffi.get_context()
    rng = np.random.default_rng(n)
    kb, M = 4, 5
    F = rng.standard_normal((kb, n, n + 3))
    Ahat = np.einsum("bik,bjk->bij", F, F) / (n + 3) + 0.05 * np.eye(n)      # kb SPD matrices
    w = rng.uniform(0.5, 3.0, size=(M, kb))
    for per_system in (False, True):
        rhs = rng.standard_normal((M, n) if per_system else n)
        c = ctx.alloc(M * n)
        ctx.reduced_solve_batch(n, kb, M, ctx.upload(Ahat), ctx.upload(w), ctx.upload(rhs), per_system, c)
        got = c.download(shape=(M, n))
        for m in range(M):
            ref = np.linalg.solve(np.einsum("b,bij->ij", w[m], Ahat), rhs[m] if per_system else rhs)
            np.testing.assert_allclose(got[m], ref, rtol=0, atol=1e-11 * np.abs(ref).max())
    x = SM.galerkin(w[0].reshape(2, 2), rhs[0], Ahat.reshape(2, 2, n, n))
    np.testing.assert_allclose(x, np.linalg.solve(np.einsum("b,bij->ij", w[0], Ahat), rhs[0]), rtol=0,
                               atol=1e-11 * np.abs(x).max())


def test_projectors_with_a_100_vector_basis(api):
    """forward_modeling / projection with more basis vectors than one 64 KB LDS matrix holds (the notebooks use
    100-vector bases), against the oracle (src/lib/SolutionsManagers.py:88-139)."""
    SM, RB = api
    sm = SM.SolutionsManagerFEM((2, 2), 8)
    g = ro.Geometry((2, 2), 8)
    rng = np.random.default_rng(100)
    a = 10.0 ** rng.uniform(0, 2, size=(7, 2, 2))
    U = sm.generate_solutions(a)
    C = RB.orthonormalize_base(rng.standard_normal((100, sm.vspace_dim)))
    scale = np.abs(U).max()
    np.testing.assert_allclose(sm.project_solutions(U, C), ro.project_solutions(g, U, C), atol=1e-10 * scale)
    np.testing.assert_allclose(sm.generate_fm_solutions(a, C), ro.generate_fm_solutions(g, a, C), atol=1e-10 * scale)


@pytest.mark.parametrize("blocks,N,K", [((2, 2), 10, 3), ((2, 3), 5, 4), ((3, 2), 4, 1), ((1, 1), 8, 2), ((2, 2), 129, 2),
                                        ((3, 3), 90, 1), ((1, 4), 70, 3)])
def test_stencil_apply_vs_oracle(api, blocks, N, K):
    """rom_stencil_apply (the `C A_pq` contractions, src/lib/SolutionsManagers.py:93-101, and A_1 of :49) against the
    oracle's sparse matrix: unit coefficient and block coefficients, slabs / columns that do not fill a workgroup."""
    from romhighcontrast_amd import _ffi
    ctx = _ffi.get_context()
    fem = _ffi.Fem(ctx, blocks[0], blocks[1], N)
    g = ro.Geometry(blocks, N)
    rng = np.random.default_rng(N + K)
    X = rng.standard_normal((K, fem.dim))
    a = 10.0 ** rng.uniform(0, 3, size=blocks)
    Xd, Yd = ctx.upload(X), ctx.alloc(K * fem.dim)
    for coef in (None, a):
        fem.stencil_apply(Xd, K, Yd, a_one=None if coef is None else coef.ravel())
        ref = ro.stencil_apply(g, np.ones(blocks) if coef is None else coef, X)
        np.testing.assert_allclose(Yd.download(shape=(K, fem.dim)), ref, rtol=0, atol=1e-13 * np.abs(ref).max())


def test_full_size_c2_properties(api):
    """BASELINE config C2 ((2,2)/N=128, 1024-parameter sweep): size-independent checks on the device."""
    SM, _ = api
    from romhighcontrast_amd import _ffi
    sm = SM.SolutionsManagerFEM((2, 2), 128)
    ctx, fem, dim = sm._ctx, sm._fem, sm.vspace_dim
    assert dim == 65025
    M = 1024
    a = 10.0 ** np.random.default_rng(20240807).uniform(0, 2, size=(M, 2, 2))
    Ud = sm.generate_solutions_device(a)
    # residual: A(a_m) u_m - B = 0 for sampled rows, through the independent stencil kernel
    Y = ctx.alloc(dim)
    for m in (0, 1, 511, 1023):
        row = _ffi.Buffer(ctx, dim).copy_from(Ud.buf, dim, 0, m * dim)
        fem.stencil_apply(row, 1, Y, a_one=a[m].ravel())
        r = Y.download(dim) - sm.B_total
        u = row.download(dim)
        # ||r||_inf relative to ||diag(A) u||_inf
        observed(f"C2: stencil residual of row {m}, ||A u - B||_inf / ||diag(A) u||_inf", np.abs(r).max() / (np.abs(u).max() * 4 * a[m].max()), 1e-11)
    # linearity in 1/a: u(2a) = u(a)/2 exactly up to roundoff
    U2 = sm.generate_solutions_device(2.0 * a[:8]).numpy()
    U1 = np.stack([Ud.buf.download(dim, offset=i * dim) for i in range(8)])
    observed("C2: homogeneity, max |2 u(2a) - u(a)| / max |u|", np.max(np.abs(2.0 * U2 - U1)) / np.abs(U1).max(), 1e-12)
    # symmetry: swapping the two block columns mirrors the solution left-right
    am = a[:4][:, :, ::-1]
    Um = sm.generate_solutions(am).reshape(4, 255, 255)[:, :, ::-1].reshape(4, -1)
    observed("C2: mirror symmetry, max |u(a mirrored) mirrored - u(a)| / max |u|", np.max(np.abs(Um - U1[:4])) / np.abs(U1).max(), 1e-11)
    # three rows against the SuperLU oracle
    g = ro.Geometry((2, 2), 128)
    idx = [0, 511, 1023]
    Uo = ro.generate_solutions(g, a[idx])
    Ug = np.stack([Ud.buf.download(dim, offset=i * dim) for i in idx])
    observed("C2: rows 0, 511, 1023 vs SuperLU oracle (rel H10)", relh10(g, Ug, Uo), SNAP_TOL)
    # norms on the device agree with the oracle's on the same vectors
    observed("C2: H10 norms on the device vs the oracle's on the same rows (relative)", np.abs(sm.H10norm(Ud)[idx] / ro.H10norm(g, Ug) - 1), 1e-12)


def test_full_size_c3_workload(api):
    """BASELINE config C3 on one GPU: the 8192-parameter sweep of C2's geometry ((2,2)/N=128) through the library's
    sharded-sweep entry (RcclSweep with world = 1: the same code path every rank of the 8-GPU job runs, gathered block in
    factored form), a 50-mode POD of the 4.26 GB block from its interface vectors AND from the materialised rows (rom_pod),
    against each other and -- on a 256-row subsample -- against numpy.linalg.svd of the centred rows (the call inside
    scikit-learn's PCA, src/lib/ReducedBasis.py:196); three rows against the SuperLU oracle."""
    SM, RB = api
    import bench
    from romhighcontrast_amd import factored, sweep
    blocks, N, M, r = (2, 2), 128, 8192, 50
    sm = SM.SolutionsManagerFEM(blocks, N)
    ctx, fem, dim = sm._ctx, sm._fem, sm.vspace_dim
    a = bench.workload_parameters("c2", blocks, M)
    assert np.array_equal(a[:1024], bench.workload_parameters("c2", blocks, 1024))   # rank 0's shard is config C2's sweep
    fs = sweep.RcclSweep(sm, 0, 1).generate_factored(a)
    assert fs.M == M
    Ud = fs.rows()
    assert Ud.shape == (M, dim)
    # the factored block reproduces a plain sweep bit for bit (rows of three "ranks")
    for lo in (0, 3 * 1024 + 5, 7 * 1024):
        chk = sm.generate_solutions_device(a[lo:lo + 4])
        assert np.array_equal(chk.numpy(), Ud.buf.download(4 * dim, offset=lo * dim, shape=(4, dim)))
    g = ro.Geometry(blocks, N)
    idx = [1024, 4097, M - 1]
    Uo = ro.generate_solutions(g, a[idx])
    Ug = np.stack([Ud.buf.download(dim, offset=i * dim) for i in idx])
    observed("C3: rows 1024, 4097, 8191 vs SuperLU oracle (rel H10)", relh10(g, Ug, Uo), SNAP_TOL)
    h1 = sm.H10norm(Ud)
    observed("C3: H10 norms from the interface vectors vs stencil norms (relative)",
             np.abs(factored.h10norm_factored(fs) - h1) / h1, 1e-10)
    # POD: factored vs rows
    modes_f, sig_f = factored.pod_modes_factored(fs, r)
    X = ctx.alloc(M * dim).copy_from(Ud.buf, M * dim)
    modes_r, sig_r = RB.pod_modes(ctx, SM.DeviceArray(X, M, dim), r)
    info = dict(RB.pod_modes.last_info)
    del X
    assert info["gram_passes"] == 0 and info["sketch_passes"] <= 3 and info["resolved_modes"] >= 25
    big = sig_r > 1e-6 * sig_r[0]
    assert big.sum() >= 12
    observed("C3 POD: singular values > 1e-6 sigma_1, rows vs factored (relative)", np.abs(sig_f[big] - sig_r[big]) / sig_r[big], 1e-7)
    observed("C3 POD: |<mode_rows, mode_factored>| - 1 for those modes", np.abs(np.abs(np.sum(modes_f[big] * modes_r[big], axis=1)) - 1.0), 1e-6)
    observed("C3 POD: orthonormality of all 50 rows (rows path)", np.abs(modes_r @ modes_r.T - np.eye(r)), 1e-9)
    observed("C3 POD: orthonormality of all 50 rows (factored path)", np.abs(modes_f @ modes_f.T - np.eye(r)), 1e-9)
    # a 256-row subsample against LAPACK on the host
    Ms = 256
    Xs = Ud.buf.download(Ms * dim, shape=(Ms, dim))
    _, sv, Vt = np.linalg.svd(Xs - Xs.mean(axis=0), full_matrices=False)
    Xd = ctx.alloc(Ms * dim).copy_from(Ud.buf, Ms * dim)
    comps, sig_s = RB.pod_modes(ctx, SM.DeviceArray(Xd, Ms, dim), 30)
    keep = sv[:30] > 1e-7 * sv[0]
    observed("C3 POD (256-row subsample): singular values > 1e-7 sigma_1 vs LAPACK (relative)",
             np.abs(sig_s[keep] - sv[:30][keep]) / sv[:30][keep], 1e-7)
    lead = sv[:30] > 1e-4 * sv[0]
    k = int(lead.sum())
    observed("C3 POD (256-row subsample): projector onto the modes > 1e-4 sigma_1 vs LAPACK",
             np.abs(comps[:k].T[:2000] @ comps[:k][:, :2000] - Vt[:k].T[:2000] @ Vt[:k][:, :2000]), 1e-8)
    # the FACTORED builders on the same subsample against LAPACK / the oracle directly, not via the row path
    fss = fs.take(np.arange(Ms))
    comps_f, sig_fs = factored.pod_modes_factored(fss, 30)
    observed("C3 POD from interface vectors (256-row subsample): singular values > 1e-7 sigma_1 vs LAPACK (relative)",
             np.abs(sig_fs[keep] - sv[:30][keep]) / sv[:30][keep], 1e-7)
    observed("C3 POD from interface vectors (256-row subsample): projector onto the modes > 1e-4 sigma_1 vs LAPACK",
             np.abs(comps_f[:k].T[:2000] @ comps_f[:k][:, :2000] - Vt[:k].T[:2000] @ Vt[:k][:, :2000]), 1e-8)
    observed("C3: H10 norms from interface vectors (256 rows) vs the oracle's norms of the downloaded rows (relative)",
             np.abs(factored.h10norm_factored(fss) / ro.H10norm(g, Xs) - 1), 1e-11)


def _oracle_rows(args):
    """(worker of a spawned process pool: SuperLU solves of the oracle, ~1 s each at 512 x 512)"""
    blocks, N, rows = args
    g = ro.Geometry(blocks, N)
    B = ro.load_vector(g)
    return np.stack([ro.solve_one(g, a, B, "lsqsparse") for a in rows])


def _refereed_rows(args):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from referee import referee
    blocks, N, rows = args
    out = [referee(blocks, N, a, verbose=False) for a in rows]
    return np.stack([o[2] for o in out]), np.stack([o[1] for o in out])


def refereed_rows_parallel(blocks, N, a, workers=2):
    """(SuperLU rows, refined-truth rows) of the parameters `a`, one factorisation each, in FRESH processes (spawn)."""
    import multiprocessing as mp
    workers = max(1, min(workers, len(a)))
    with mp.get_context("spawn").Pool(workers) as pool:
        parts = pool.map(_refereed_rows, [(blocks, N, a[i::workers]) for i in range(workers)])
    slu, tru = np.empty((len(a), parts[0][0].shape[1])), np.empty((len(a), parts[0][0].shape[1]))
    for i, (p0, p1) in enumerate(parts):
        slu[i::workers], tru[i::workers] = p0, p1
    return slu, tru


def oracle_sweep_parallel(blocks, N, a, workers=8):
    """The oracle's generate_solutions over a pool of FRESH processes (spawn: this process has initialised the GPU,
    it must not fork)."""
    import multiprocessing as mp
    workers = max(1, min(workers, len(a)))
    chunks = [(blocks, N, a[i::workers]) for i in range(workers)]
    with mp.get_context("spawn").Pool(workers) as pool:
        parts = pool.map(_oracle_rows, chunks)
    out = np.empty((len(a), parts[0].shape[1]))
    for i, part in enumerate(parts):
        out[i::workers] = part
    return out


def test_full_size_c4_workload(api):
    """BASELINE config C4 at its stated workload ((3,3)/N=171: 512 x 512 unknowns, contrast up to 1e8, the 1024-row
    training set of SURVEY 8d, greedy reduced basis to n = 50, src/lib/ReducedBasis.py:112-139):
      * residual of sampled rows through the independent stencil kernel; the first 64 rows (which hold the limit
        solutions) against the SuperLU oracle;
      * greedy to n = 50 in both modes on snapshot rows and on the factored block: identical picks, equal error curves;
      * the greedy on the 64-row subsample against the ORACLE's greedy on the oracle's own snapshots (the oracle is
        ~1 s per solve here, so it is the oracle that is subsampled, not the GPU): same picks while the errors are
        above roundoff, error curves within 1e-8."""
    SM, RB = api
    import bench
    from romhighcontrast_amd import _ffi, factored
    blocks, N, M, n = (3, 3), 171, 1024, 50
    sm = SM.SolutionsManagerFEM(blocks, N)
    ctx, fem, dim = sm._ctx, sm._fem, sm.vspace_dim
    assert dim == 262144
    a = bench.workload_parameters("c4", blocks, M)
    assert np.all(a[0] == 1) and np.all(a[10] == 1e8) and a[11:].max() <= 1e8
    Ud = sm.generate_solutions_device(a)
    Y = ctx.alloc(dim)
    for m in (0, 3, 5, 10, 500, M - 1):
        row = _ffi.Buffer(ctx, dim).copy_from(Ud.buf, dim, 0, m * dim)
        fem.stencil_apply(row, 1, Y, a_one=a[m].ravel())
        res = Y.download(dim) - sm.B_total
        u = row.download(dim)
        observed(f"C4: stencil residual of row {m} (|r|_inf / (4 a_max |u|_inf))", np.abs(res).max() / (np.abs(u).max() * 4 * a[m].max()), 1e-11)
    g = ro.Geometry(blocks, N)
    Ms = 64
    Uo = oracle_sweep_parallel(blocks, N, a[:Ms])
    Ug = Ud.buf.download(Ms * dim, shape=(Ms, dim))
    # A block that touches no Dirichlet side and dominates all its neighbours "floats": its plateau level is fixed by
    # fluxes that are 1e-8 of the matrix entries (row 5: the centre block at 1e8 among ones), and an fp64 direct solver
    # gets it to ~contrast * eps * N only.  The extended-precision referee (tests/golden/make_referee.py) says who is
    # right there: SuperLU -- the oracle -- is 6.84e-5 from the truth, the GPU 9.4e-9
    # (profiles/r03_referee_floating_rows.txt).  So row 5 is compared with the TRUTH, and must be at least as close to
    # it as the oracle is; every other row is compared with the oracle at 1e-11.
    centre = a[:Ms, 1, 1]
    others = np.delete(a[:Ms].reshape(Ms, 9), 4, axis=1).max(axis=1)
    floating = centre >= 1e4 * others
    assert floating[5] and floating.sum() == 1
    ref = load_golden("referee_c4_row5.npz")
    assert np.array_equal(ref["a"], a[5])
    truth = ref["truth"][None]
    e_oracle = float(relh10(g, Uo[5:6], truth)[0])
    assert abs(e_oracle - float(ref["err_superlu_vs_truth"])) < 0.05 * e_oracle   # the referee's record of the oracle's error
    e_gpu = observed(f"C4: floating row 5 vs long-double truth (SuperLU oracle: {e_oracle:.2e})",
                     relh10(g, Ug[5:6], truth), max(e_oracle, 1e-11))
    observed("C4: floating row 5 vs long-double truth, against contrast * eps * N = 1e8 x 1.1e-16 x 513", e_gpu, 1e8 * 1.1e-16 * 513)
    Uo[5] = truth[0]                                   # from here on the oracle block holds the refereed row
    err = relh10(g, Ug, Uo)
    observed("C4: first 64 training rows vs SuperLU oracle, ordinary rows (rel H10)", err[~floating], SNAP_TOL)
    h1 = sm.H10norm(Ud)
    h1o = ro.H10norm(g, Uo)
    observed("C4: H10 norms of the first 64 rows vs oracle, ordinary rows (relative)", (np.abs(h1[:Ms] - h1o) / h1o)[~floating], 1e-11)
    observed("C4: H10 norm of the floating row 5 vs the truth's (relative)", (np.abs(h1[:Ms] - h1o) / h1o)[floating], 2 * e_gpu)
    # greedy n = 50 on the full training set: rows vs factored block
    Yf = ctx.alloc(M * fem.reduced_stride)
    fem.solve_reduced(ctx.upload(a.reshape(M, -1)), M, Yf)
    ctx.solve_status()
    fs = factored.FactoredSnapshots(sm, Yf, M)
    np.testing.assert_allclose(factored.h10norm_factored(fs), h1, rtol=1e-10)
    for mode in (RB.GREEDY_FOR_H10, RB.GREEDY_FOR_GALERKIN):
        rb_r = RB.ReducedBasisGreedy(mode).build(n, sm, SM.DeviceArray(Ud.buf, M, dim), a, h1)   # (a plain block: the row route)
        rb_f = RB.ReducedBasisGreedy(mode).build(n, sm, fs, a, h1)
        assert rb_r.picks[0] == 0 and rb_r.max_errors[0] == 1.0 and len(rb_r.picks) == n
        er, ef = np.array(rb_r.max_errors), np.array(rb_f.max_errors)
        # picks are compared while the selection is not a coin toss between near-equal errors (relative gap of the two
        # error curves); the curves themselves agree throughout (the factored build works in energy coordinates of the
        # interface vectors: a different route to the same numbers, at kappa(A) up to 1e13)
        # H^1_0 mode: both routes are orthogonal projections (sums of squares): 1e-12.  Galerkin mode: the error of snapshot m
        # comes from the reduced system (C A(a_m) C^T) c = C B, whose condition number reaches the contrast of the training
        # set, 1e8: two exact routes to it differ by contrast x eps = 2.2e-8 in the coefficients (observed 6e-9 ... 1.1e-8
        # over rounds 3-4, moving with the rounding of the block forms S_b = B^T A_b B) -- bound 4 x contrast x eps
        bound_rf = 1e-12 if mode == RB.GREEDY_FOR_H10 else 4 * 1e8 * 2.2e-16
        observed(f"C4 greedy n=50 {mode}: error curve, rows vs factored block (bound {bound_rf:.1e})", np.abs(er - ef), bound_rf)
        same = [p == q for p, q in zip(rb_r.picks, rb_f.picks)]
        assert sum(same) >= n - 2, (mode, rb_r.picks, rb_f.picks)
        # the worst-case error decays (nine free blocks at contrast 1e8: slowly); the H^1_0 projection error never grows
        # (the Galerkin solution is not the best approximation in this norm: its curve may go up a little)
        assert er[-1] < 0.2 * er[1], (mode, er)
        if mode == RB.GREEDY_FOR_H10:
            assert all(e2 <= e1 * (1 + 1e-9) for e1, e2 in zip(er, er[1:])), (mode, er)
        assert rb_r.basis.shape == (n, dim)
        # the plain call on a block that carries its interface vectors (what sm.generate_solutions_device returns) takes the
        # factored route by itself in both modes (Galerkin since the quadratic forms are compensated: the referee below);
        # galerkin_on_interface_vectors=False keeps the rows -- same picks and curves as the explicit calls
        Ucarry = SM.DeviceArray(Ud.buf, M, dim, factored=fs)
        rb_c = RB.ReducedBasisGreedy(mode).build(n, sm, Ucarry, a, h1)
        assert rb_c.picks == rb_f.picks and rb_c.max_errors == rb_f.max_errors, mode
        if mode == RB.GREEDY_FOR_GALERKIN:
            rb_o = RB.ReducedBasisGreedy(mode).build(n, sm, Ucarry, a, h1, galerkin_on_interface_vectors=False)
            assert rb_o.picks == rb_r.picks and rb_o.max_errors == rb_r.max_errors
        assert np.array_equal(rb_c.basis, rb_r.basis if rb_c.picks == rb_r.picks else rb_c.basis)
    # Greedy on a 256-row subsample (the seeded limit rows + 245 random ones), n = 24 -- past the 0.99 plateau of the first
    # iterations -- three routes end to end on the SAME rows: the oracle's greedy (NumPy: src/lib/ReducedBasis.py:112-139 on
    # the downloaded GPU rows, which the 64-row comparison above holds to the oracle's), rom_greedy on rows, and
    # rom_greedy_factored on the interface vectors (the factored builders against the oracle
    # directly, not against the row path).  In Galerkin mode the error of a parameter comes from a reduced system whose
    # condition number reaches the contrast, where two exact fp64 routes differ by contrast x eps; tests/referee.py's
    # 80-bit Galerkin truth on the exact span of the picked rows says how far EACH route is from the numbers themselves.
    import referee
    Ms2, ns = 256, 24
    U2 = Ud.buf.download(Ms2 * dim, shape=(Ms2, dim))
    h2, h2o = h1[:Ms2], ro.H10norm(g, U2)   # (each side divides by ITS OWN norms: the first errors are exactly 1.0, pick 0)
    sub = SM.DeviceArray(_ffi.Buffer(ctx, Ms2 * dim).copy_from(Ud.buf, Ms2 * dim), Ms2, dim)
    fs2 = fs.take(np.arange(Ms2))
    for mode, omode in ((RB.GREEDY_FOR_H10, ro.GREEDY_FOR_H10), (RB.GREEDY_FOR_GALERKIN, ro.GREEDY_FOR_GALERKIN)):
        rb = RB.ReducedBasisGreedy(mode).build(ns, sm, sub, a[:Ms2], h2)
        rbf = RB.ReducedBasisGreedy(mode).build(ns, sm, fs2, a[:Ms2], h2)
        _, _, picks_o, errs_o = ro.greedy_build(g, ns, U2, a[:Ms2], h2o, greedy_for=omode, method="lsq", return_errors=True)
        errs_o = np.array(errs_o)
        er, ef = np.array(rb.max_errors), np.array(rbf.max_errors)
        assert er[-1] < 0.9, (mode, er)   # (the comparison reaches the informative part of the curve)
        ok = errs_o > 1e-9
        for name, picks in (("rows", rb.picks), ("factored", rbf.picks)):
            assert [p for p, k in zip(picks, ok) if k] == [p for p, k in zip(picks_o, ok) if k], (mode, name, picks, picks_o)
        if mode == RB.GREEDY_FOR_H10:
            observed(f"C4 greedy on 256 rows {mode}, n = {ns}: rows route vs the oracle's greedy (BASELINE: 1e-10)", np.abs(er - errs_o), 1e-10)
            observed(f"C4 greedy on 256 rows {mode}, n = {ns}: FACTORED route vs the oracle's greedy (BASELINE: 1e-10)", np.abs(ef - errs_o), 1e-10)
        else:
            # entry j of a curve = the worst error with the first j picks (entry 0: the empty basis, 1.0)
            truth = referee.galerkin_truth_nested(g, a[:Ms2], U2[rb.picks[:ns - 1]], U2, range(1, ns))
            et = np.array([1.0] + [truth[j].max() for j in range(1, ns)])
            d_o, d_r, d_f = np.abs(errs_o - et), np.abs(er - et), np.abs(ef - et)
            # The oracle (= the reference's arithmetic: fp64 QR of the picked rows, dense solves) sets the scale: where the worst
            # parameter of an iteration has a reduced system of condition number ~ contrast, the reference's own number is
            # contrast x eps x (a few) from the truth, so "within 1e-10 of the reference" cannot be asked of any other exact
            # route.  A GPU route passes when its curve is within the BASELINE bar of the TRUTH or no further from it than
            # 4 x the worst distance of the oracle's curve.
            bar = max(1e-10, 4 * d_o.max())
            observed(f"C4 greedy on 256 rows {mode}, n = {ns}: (for the record) the ORACLE's curve vs the 80-bit truth", d_o, 1e-7)
            observed(f"C4 greedy on 256 rows {mode}, n = {ns}: rows route vs the 80-bit truth (bound max(1e-10, 4 x oracle's) = {bar:.1e})", d_r, bar)
            observed(f"C4 greedy on 256 rows {mode}, n = {ns}: FACTORED route vs the 80-bit truth (same bound)", d_f, bar)


def test_full_size_c5_workload(api):
    """BASELINE config C5 at its stated workload ((4,4)/N=256: 1024 x 1024 cells, dim 1 046 529, 4096-parameter sweep =
    a 34.3 GB snapshot block, POD to 50 modes, src/lib/ReducedBasis.py:189-200).  The oracle cannot sweep this
    (25 s per solve), so: residual through the independent stencil kernel, homogeneity u(c a) = u(a) / c, the
    unit-coefficient solution's symmetry, two rows against SuperLU; POD singular values of the full block from rows
    against the factored form, and -- on a 256-row subsample -- against numpy.linalg.svd of the downloaded rows."""
    SM, RB = api
    import bench
    from romhighcontrast_amd import _ffi, factored
    blocks, N, M, r = (4, 4), 256, 4096, 50
    sm = SM.SolutionsManagerFEM(blocks, N)
    ctx, fem, dim = sm._ctx, sm._fem, sm.vspace_dim
    assert dim == 1023 * 1023
    a = bench.workload_parameters("c5", blocks, M)
    a[0] = 1.0
    a[1] = 7.0 * a[2]
    Ud = sm.generate_solutions_device(a)
    Y = ctx.alloc(dim)
    for m in (0, 2, 2048, M - 1):
        row = _ffi.Buffer(ctx, dim).copy_from(Ud.buf, dim, 0, m * dim)
        fem.stencil_apply(row, 1, Y, a_one=a[m].ravel())
        res = Y.download(dim) - sm.B_total
        u = row.download(dim)
        observed(f"C5: stencil residual of row {m}, ||A u - B||_inf / ||diag(A) u||_inf", np.abs(res).max() / (np.abs(u).max() * 4 * a[m].max()), 1e-11)
    u0 = Ud.buf.download(dim, offset=0).reshape(1023, 1023)
    observed("C5: symmetries of the unit-coefficient solution (up-down, transpose), relative to max u",
             max(np.abs(u0 - u0[::-1, :]).max(), np.abs(u0 - u0.T).max()) / u0.max(), 1e-12)
    u1, u2 = Ud.buf.download(dim, offset=dim), Ud.buf.download(dim, offset=2 * dim)
    observed("C5: homogeneity, max |7 u(7a) - u(a)| / max |u|", np.abs(7.0 * u1 - u2).max() / np.abs(u2).max(), 1e-12)
    g = ro.Geometry(blocks, N)
    idx = [5, M - 1]
    Uo, Ut = refereed_rows_parallel(blocks, N, a[idx], workers=2)
    Ug = np.stack([Ud.buf.download(dim, offset=i * dim) for i in idx])
    observed("C5: rows 5, 4095 vs SuperLU oracle (rel H10)", relh10(g, Ug, Uo), SNAP_TOL)
    # (how much of that distance is the oracle's own: SuperLU + refinement with long-double edge-form residuals = the truth)
    observed("C5: rows 5, 4095 vs the refined truth (rel H10)", relh10(g, Ug, Ut), SNAP_TOL)
    observed("C5: (for the record) SuperLU oracle vs the refined truth on those rows (rel H10)", relh10(g, Uo, Ut), 1e-9)
    h1 = sm.H10norm(Ud)
    assert np.all(h1 > 0)
    observed("C5: homogeneity of the H10 norms, |h1(a) / h1(7a) - 7|", abs(h1[2] / h1[1] - 7.0), 1e-11)
    observed("C5: H10 norms on the device vs the oracle's on the same rows (relative)", np.abs(h1[idx] / ro.H10norm(g, Ug) - 1), 1e-11)
    # POD of the full block: rows vs interface vectors
    Yf = ctx.alloc(M * fem.reduced_stride)
    fem.solve_reduced(ctx.upload(a.reshape(M, -1)), M, Yf)
    ctx.solve_status()
    fs = factored.FactoredSnapshots(sm, Yf, M)
    modes_f, sig_f = factored.pod_modes_factored(fs, r)
    X = ctx.alloc(M * dim).copy_from(Ud.buf, M * dim)
    modes_r, sig_r = RB.pod_modes(ctx, SM.DeviceArray(X, M, dim), r)
    del X
    big = sig_r > 1e-6 * sig_r[0]
    assert big.sum() >= 20
    observed("C5 POD: singular values > 1e-6 sigma_1, rows vs factored (relative)", np.abs(sig_f[big] / sig_r[big] - 1), 1e-7)
    observed("C5 POD: |<mode_rows, mode_factored>| - 1 for those modes", np.abs(np.abs(np.sum(modes_f[big] * modes_r[big], axis=1)) - 1.0), 1e-6)
    observed("C5 POD: orthonormality of those rows", np.abs(modes_r[big] @ modes_r[big].T - np.eye(int(big.sum()))), 1e-9)
    # a 256-row subsample against LAPACK on the host (the SVD inside sklearn's PCA, src/lib/ReducedBasis.py:196)
    Ms = 256
    Xs = Ud.buf.download(Ms * dim, shape=(Ms, dim))
    sv = np.linalg.svd(Xs - Xs.mean(axis=0), full_matrices=False)[1]
    Xd = ctx.alloc(Ms * dim).copy_from(Ud.buf, Ms * dim)
    _, sig_s = RB.pod_modes(ctx, SM.DeviceArray(Xd, Ms, dim), 30)
    keep = sv[:30] > 1e-7 * sv[0]
    observed("C5 POD (256-row subsample): singular values > 1e-7 sigma_1 vs LAPACK (relative)", np.abs(sig_s[keep] / sv[:30][keep] - 1), 1e-7)
    # the FACTORED builders on the same subsample against LAPACK / the oracle directly
    fss = fs.take(np.arange(Ms))
    _, sig_fs = factored.pod_modes_factored(fss, 30)
    observed("C5 POD from interface vectors (256-row subsample): singular values > 1e-7 sigma_1 vs LAPACK (relative)",
             np.abs(sig_fs[keep] / sv[:30][keep] - 1), 1e-7)
    observed("C5: H10 norms from interface vectors (256 rows) vs the oracle's norms of the downloaded rows (relative)",
             np.abs(factored.h10norm_factored(fss) / ro.H10norm(g, Xs) - 1), 1e-11)


@pytest.mark.parametrize("blocks,N", [((2, 2), 128), ((2, 2), 100), ((2, 2), 64)])
def test_single_tile_solve_does_not_depend_on_its_batch(api, blocks, N):
    """The single-tile reduced solve (k_solve1) runs four systems per workgroup, one per wave, and its waves share the
    assembly: a system's interface vector -- hence its snapshot row -- must not depend on which systems share its
    workgroup, on its place in it, or on waves of the last workgroup having no system.  Batches of 1, 2, 3, 5, 7 systems
    and a permuted batch against one sweep of all of them, bit for bit; the rows also against the oracle."""
    from romhighcontrast_amd import _ffi
    ctx = _ffi.get_context()
    fem = _ffi.Fem(ctx, blocks[0], blocks[1], N)
    if fem.n_tiles != 1:
        pytest.skip("this geometry does not take the single-tile path")
    k = blocks[0] * blocks[1]
    M = 18
    a = 10.0 ** np.random.default_rng(N).uniform(0, 4, size=(M, k))
    U = ctx.alloc(M * fem.dim)
    fem.solve_batch(ctx.upload(a), M, U)
    ref = U.download(shape=(M, fem.dim))
    lo = 0
    for n in (1, 2, 3, 5, 7):
        Us = ctx.alloc(n * fem.dim)
        fem.solve_batch(ctx.upload(a[lo:lo + n]), n, Us)
        assert np.array_equal(Us.download(shape=(n, fem.dim)), ref[lo:lo + n]), f"batch of {n} systems from row {lo}"
        lo += n
    perm = np.random.default_rng(1).permutation(M)
    Up = ctx.alloc(M * fem.dim)
    fem.solve_batch(ctx.upload(a[perm]), M, Up)
    assert np.array_equal(Up.download(shape=(M, fem.dim)), ref[perm])
    g = ro.Geometry(blocks, N)
    rows = [0, 7, M - 1]
    Uo = ro.generate_solutions(g, a[rows].reshape(len(rows), *blocks), "lsqsparse")
    observed(f"single-tile path {blocks} N={N}: rows vs the SuperLU oracle (rel H10)", (ro.H10norm(g, ref[rows] - Uo) / ro.H10norm(g, Uo)).max(), SNAP_TOL)


@pytest.mark.parametrize("blocks,N,M", [((2, 2), 128, 200), ((3, 3), 20, 70), ((1, 2), 6, 5), ((2, 2), 16, 130), ((1, 1), 8, 3)])
def test_two_stage_sweep_is_bit_identical(api, blocks, N, M):
    """rom_solve_reduced_async + rom_expand_batch_async (the factored form that travels between GPUs) must
    reproduce rom_solve_batch bit for bit, also when the interface vectors are expanded in a different batch
    composition than they were solved in (rows of several 'ranks' concatenated, different row offsets)."""
    from romhighcontrast_amd import _ffi
    ctx = _ffi.get_context()
    fem = _ffi.Fem(ctx, blocks[0], blocks[1], N)
    a = 10.0 ** np.random.default_rng(N + M).uniform(0, 3, size=(M, blocks[0] * blocks[1]))
    ab = ctx.upload(a)
    U = ctx.alloc(M * fem.dim)
    fem.solve_batch(ab, M, U)
    ref = U.download(shape=(M, fem.dim))
    stride = fem.reduced_stride
    # "two ranks": the second half is solved first and lands behind the first in the gathered buffer
    h = M // 2
    Y = ctx.alloc(max((M + 3) * stride, 1))
    Y.fill(float("nan"))  # whatever was in the buffer must not matter
    fem.solve_reduced(ctx.upload(a[h:]), M - h, Y, y_row0=3 + h)
    fem.solve_reduced(ctx.upload(a[:h]), h, Y, y_row0=3)
    U2 = ctx.alloc((M + 2) * fem.dim)
    fem.expand(ab, M, Y, U2, y_row0=3, row0=2)
    ctx.solve_status()
    assert np.array_equal(U2.download(shape=(M + 2, fem.dim))[2:], ref)
    # the COMPACT form is what travels (rom_fem_pack_reduced_async / _unpack_): only the entries the expansion reads;
    # unpacked into a fresh buffer (nodal part zero) it must expand to the same bits
    kc = fem.compact_stride
    cols = fem.reduced_inputs
    assert kc == cols.size and kc <= stride
    Yc = ctx.alloc(max((M + 1) * kc, 1))
    Yc.fill(float("nan"))
    fem.pack_reduced(Y, M, Yc, y_row0=3, c_row0=1)
    assert np.array_equal(Yc.download(shape=(M + 1, kc))[1:], Y.download(shape=(M + 3, stride))[3:][:, cols])
    Y3 = ctx.alloc(max((M + 2) * stride, 1))
    Y3.fill(float("nan"))
    fem.unpack_reduced(Yc, M, Y3, c_row0=1, y_row0=2)
    U3 = ctx.alloc(M * fem.dim)
    fem.expand(ab, M, Y3, U3, y_row0=2)
    ctx.solve_status()
    assert np.array_equal(U3.download(shape=(M, fem.dim)), ref)


@pytest.mark.parametrize("name", ["b22", "b33", "b44"])
def test_repeated_sweeps_are_bit_identical(name):
    """Same parameters, fresh FE spaces, other work in between: the snapshots must not move by a bit (no atomics,
    no dependence on what a buffer, a workspace or the LDS held before)."""
    from romhighcontrast_amd import _ffi
    ctx = _ffi.get_context()
    z = load_golden("g4_contrast.npz")
    blocks, N = tuple(int(x) for x in z[f"{name}_blocks"]), int(z[f"{name}_N"])
    a = np.asarray(z[f"{name}_a"], dtype=np.float64).reshape(len(z[f"{name}_a"]), -1)
    M = len(a)
    ref = None
    for rep in range(8):
        if rep % 4 == 0:
            fem = _ffi.Fem(ctx, blocks[0], blocks[1], N)
            other = _ffi.Fem(ctx, 2, 2, 16 + rep)  # a different geometry churns allocator, caches and LDS
            Uo = ctx.alloc(8 * other.dim)
            other.solve_batch(ctx.upload(np.full((8, 4), 2.0)), 8, Uo)
        U = ctx.alloc(M * fem.dim)
        U.fill(float("nan"))
        fem.solve_batch(ctx.upload(a), M, U)
        out = U.download(shape=(M, fem.dim))
        assert not np.isnan(out).any()
        if ref is None:
            ref = out
        else:
            assert np.array_equal(out, ref), rep


@pytest.mark.parametrize("kind,m,n,k", [("nt", 24, 1024, 65025), ("nt", 50, 300, 4099), ("nt", 1, 257, 2048),
                                         ("nt", 64, 1000, 5007), ("nt", 17, 256, 2063), ("nt", 33, 513, 70001),
                                         ("nn", 24, 65025, 1024), ("nn", 50, 4097, 300), ("nn", 3, 1029, 129),
                                         ("nn", 64, 2000, 1013), ("nn", 16, 1024, 128), ("nn", 49, 262144, 50),
                                         ("nt", 1000, 50, 4099), ("nt", 300, 1, 2048), ("nt", 257, 64, 5007),
                                         ("nn", 1000, 4097, 50), ("nn", 130, 1029, 16), ("nn", 65, 2000, 256)])
def test_thin_gemms_vs_numpy(kind, m, n, k):
    """rom_gemm_nt / rom_gemm_nn with a thin operand against a snapshot-shaped block (the LDS-DMA kernels k_gemm_nt_thin /
    k_gemm_nn_thin): odd leading dimensions, rows / columns / K that are no multiples of the tile, alpha and beta, every
    count of 16-row blocks, the thin operand on either side (transposed product) and the tall-A / short-K lift -- against
    float64 NumPy with the bound of a length-K dot product."""
    from romhighcontrast_amd import _ffi
    ctx = _ffi.get_context()
    rng = np.random.default_rng(m * 7 + n + k)
    A = rng.standard_normal((m, k))
    C0 = rng.standard_normal((m, n))
    alpha, beta = -0.75, 0.5
    if kind == "nt":
        B = rng.standard_normal((n, k))
        ref = alpha * (A @ B.T) + beta * C0
    else:
        B = rng.standard_normal((k, n))
        ref = alpha * (A @ B) + beta * C0
    Ad, Bd, Cd = ctx.upload(A), ctx.upload(B), ctx.upload(C0)
    if kind == "nt":
        ctx.gemm_nt(m, n, k, Ad, 0, k, Bd, 0, k, Cd, 0, n, alpha=alpha, beta=beta)
    else:
        ctx.gemm_nn(m, n, k, Ad, 0, k, Bd, 0, n, Cd, 0, n, alpha=alpha, beta=beta)
    got = Cd.download(shape=(m, n))
    scale = np.abs(alpha) * (np.abs(A) @ (np.abs(B.T) if kind == "nt" else np.abs(B))) + np.abs(beta * C0)
    observed(f"thin gemm_{kind} {m}x{n}x{k}: |C - ref| / (|alpha| |A||B| + |beta C|)", np.abs(got - ref) / scale, 4e-16 * np.sqrt(k) + 1e-15)


@pytest.mark.parametrize("M,D", [(512, 4096), (700, 5001), (1025, 4111), (640, 65025)])
def test_gram_128_tiles_vs_numpy(api, M, D):
    """rom_gram on its 128 x 128 LDS-DMA path (M >= 512, D >= 4096): ragged last row tile, rows of odd length (8-byte
    aligned only), a K that is not a multiple of the 16-wide chunk (tail through registers), several K splits."""
    from romhighcontrast_amd import _ffi
    ctx = _ffi.get_context()
    x = np.random.default_rng(M + D).standard_normal((M, D)) * np.logspace(0, -6, M)[:, None]
    X, G = ctx.upload(x), ctx.alloc(M * M)
    G.fill(float("nan"))
    ctx.gram(M, D, X, 0, D, G, 0, M)
    g = G.download(shape=(M, M))
    ref = x @ x.T
    scale = np.sqrt(np.outer(np.diag(ref), np.diag(ref)))
    assert np.isfinite(g).all()
    assert np.array_equal(g, g.T)
    assert (np.abs(g - ref) / scale).max() < 1e-13


def test_ab_build_variants_agree():
    """The product picks extension tilings, workgroup orders, systems per workgroup and the tile-assembly form by geometry;
    libromhc_ab.so (the same objects with rom_fem_setup.hip compiled -DROMHC_AB) can FORCE each of them, and
    tests/ab_variants.py asserts on eleven geometries that every forced form gives the rows of the default, bit for bit
    (in a subprocess: a process loads one library; the product build reads none of those switches)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_ab = os.path.join(root, "romhighcontrast_amd", "csrc", "libromhc_ab.so")
    assert os.path.exists(lib_ab), "libromhc_ab.so missing: `make -C romhighcontrast_amd/csrc` builds it beside libromhc.so"
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "ab_variants.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=900)
    out = r.stdout.decode()
    assert r.returncode == 0 and out.rstrip().endswith("OK"), out[-4000:]


@pytest.mark.parametrize("env", ["ROMHC_NO_COMPRESS", "ROMHC_NO_PREELIM", "ROMHC_NO_FUSED", "ROMHC_NO_LOWRANK_EXT", "ROMHC_NO_EXT128"])
def test_algorithm_switches_agree(api, env, monkeypatch):
    """Every exact reduction of the solver can be switched off (A/B checks): the snapshots must not move beyond
    rounding, and each variant must itself meet the parity bound against the oracle."""
    from romhighcontrast_amd import _ffi
    ctx = _ffi.get_context()
    for blocks, N, M in (((2, 2), 128, 130), ((3, 3), 24, 40), ((2, 3), 40, 20)):
        a = 10.0 ** np.random.default_rng(N).uniform(0, 3, size=(M, blocks[0] * blocks[1]))
        ab = ctx.upload(a)
        monkeypatch.delenv(env, raising=False)
        fem = _ffi.Fem(ctx, blocks[0], blocks[1], N)
        U = ctx.alloc(M * fem.dim)
        fem.solve_batch(ab, M, U)
        ref = U.download(shape=(M, fem.dim))
        monkeypatch.setenv(env, "1")
        fem2 = _ffi.Fem(ctx, blocks[0], blocks[1], N)
        U2 = ctx.alloc(M * fem.dim)
        fem2.solve_batch(ab, M, U2)
        alt = U2.download(shape=(M, fem.dim))
        monkeypatch.delenv(env, raising=False)
        g = ro.Geometry(blocks, N)
        assert relh10(g, alt, ref).max() < 1e-11, (env, blocks, N)
        if N <= 40:
            assert relh10(g, alt[:4], ro.generate_solutions(g, a[:4].reshape((4,) + blocks))).max() < SNAP_TOL


def test_edge_compression_tolerance(api, monkeypatch):
    """The edge compression stops at 1e-14 of the strongest direction where that removes work (3x3 / N=171: 301 reduced
    unknowns = 5 tile columns = 15 tiles instead of 347 / 6 / 21 with the basis run down to 1e-17) and keeps the weaker
    directions where they are free (2x2 / N=128: one tile either way).  The snapshots of the two settings agree to
    1e-13 (measured 1.2e-14, profiles/r02_compress_tolerance.txt), far inside the parity bound."""
    from romhighcontrast_amd import _ffi
    ctx = _ffi.get_context()
    monkeypatch.delenv("ROMHC_COMPRESS_TOL", raising=False)
    assert _ffi.Fem(ctx, 2, 2, 128).n_tiles == 1
    blocks, N, M = (3, 3), 171, 48
    a = 10.0 ** np.random.default_rng(3).uniform(-4, 4, size=(M, 9))
    ab = ctx.upload(a)
    out = {}
    for name, tol, tiles in (("default", None, 15), ("1e-17", "1e-17", 21)):
        if tol:
            monkeypatch.setenv("ROMHC_COMPRESS_TOL", tol)
        fem = _ffi.Fem(ctx, blocks[0], blocks[1], N)
        assert fem.n_tiles == tiles, (name, fem.n_tiles)
        U = ctx.alloc(M * fem.dim)
        fem.solve_batch(ab, M, U)
        out[name] = U.download(shape=(M, fem.dim))
    monkeypatch.delenv("ROMHC_COMPRESS_TOL", raising=False)
    d = np.linalg.norm(out["default"] - out["1e-17"], axis=1) / np.linalg.norm(out["1e-17"], axis=1)
    assert d.max() < 1e-13, d.max()


def test_sub_batch_streams_agree(api, monkeypatch):
    """A sweep over a geometry whose reduced solve is the tile Cholesky can run as concurrent sub-batches on separate HIP
    streams (ROMHC_STREAMS = 2 .. 4, read when the context is created; disjoint workspaces and rows): same rows as the default
    -- one batch on one stream since round 5 -- with two and with four sub-batches."""
    from romhighcontrast_amd import _ffi
    blocks, N, M = (3, 3), 24, 700
    a = 10.0 ** np.random.default_rng(5).uniform(0, 3, size=(M, 9))
    out = {}
    for name, streams in (("default", None), ("one", "2"), ("four", "4")):
        if streams is None:
            monkeypatch.delenv("ROMHC_STREAMS", raising=False)
            ctx = _ffi.get_context()
        else:
            monkeypatch.setenv("ROMHC_STREAMS", streams)
            ctx = _ffi.Context(0)  # (a second context on the device: its own streams and workspace)
        fem = _ffi.Fem(ctx, blocks[0], blocks[1], N)
        U = ctx.alloc(M * fem.dim)
        U.fill(float("nan"))
        fem.solve_batch(ctx.upload(a), M, U)
        out[name] = U.download(shape=(M, fem.dim))
    monkeypatch.delenv("ROMHC_STREAMS", raising=False)
    assert np.array_equal(out["one"], out["default"])
    assert np.array_equal(out["four"], out["default"])
    g = ro.Geometry(blocks, N)
    assert relh10(g, out["default"][-3:], ro.generate_solutions(g, a[-3:].reshape((3,) + blocks))).max() < SNAP_TOL


def test_factored_snapshot_block(api):
    """U = Y B^T: rows, Gram matrix and POD of a sweep formed from the interface vectors alone must agree with
    the same quantities formed from the materialised snapshot rows."""
    from romhighcontrast_amd import factored
    SM, RB = api
    sm = SM.SolutionsManagerFEM((2, 2), 32)
    fem, ctx = sm._fem, sm._ctx
    assert fem.expansion_is_linear
    M = 150
    a = 10.0 ** np.random.default_rng(11).uniform(0, 2, size=(M, 2, 2))
    U = sm.generate_solutions(a)
    Y = ctx.alloc(M * fem.reduced_stride)
    fem.solve_reduced(ctx.upload(a.reshape(M, -1)), M, Y)
    ctx.solve_status()
    fs = factored.FactoredSnapshots(sm, Y, M)
    assert np.array_equal(fs.rows().numpy(), U)                       # same kernels, same bits
    assert np.array_equal(fs.rows(40, 47).numpy(), U[40:47])
    G = fs.gram().download(M * M, shape=(M, M))
    Gref = U @ U.T
    assert np.abs(G - Gref).max() <= 1e-11 * np.abs(Gref).max()
    n = 12
    comps, sig = factored.pod_modes_factored(fs, n)
    comps_ref, sig_ref = RB.pod_modes(ctx, SM.DeviceArray(ctx.upload(U), M, sm.vspace_dim), n)
    np.testing.assert_allclose(sig, sig_ref, rtol=1e-7, atol=1e-12 * sig_ref[0])
    big = sig_ref > 1e-6 * sig_ref[0]
    assert np.abs(comps[big] - comps_ref[big]).max() < 1e-6
    assert np.abs(comps @ comps.T - np.eye(n)).max() < 1e-9
    # the PCA builder takes the factored block as its training set (same basis as from rows)
    rb_f = RB.ReducedBasisPCA().build(6, sm, fs, a, 1)
    rb_r = RB.ReducedBasisPCA().build(6, sm, U, a, 1)
    assert np.abs(rb_f.basis - rb_r.basis).max() < 1e-6 and np.array_equal(np.asarray(rb_f.a), np.asarray(rb_r.a))
    sub = fs.take([3, 77, 5])
    assert np.array_equal(sub.rows().numpy(), U[[3, 77, 5]])
    # H^1_0 norms and both greedy builders on the factored block: same picks, same error curves as on rows
    h1 = sm.H10norm(U)
    np.testing.assert_allclose(factored.h10norm_factored(fs), h1, rtol=1e-11)
    for mode in (RB.GREEDY_FOR_H10, RB.GREEDY_FOR_GALERKIN):
        g_r = RB.ReducedBasisGreedy(mode).build(10, sm, U, a, h1)
        g_f = RB.ReducedBasisGreedy(mode).build(10, sm, fs, a, h1)
        assert g_f.picks == g_r.picks, (mode, g_f.picks, g_r.picks)
        np.testing.assert_allclose(g_f.max_errors, g_r.max_errors, rtol=1e-8, atol=1e-11)
        assert np.array_equal(g_f.basis, g_r.basis)
    # the documented default normalisation (solutions2train_h1norm = 1) and an arbitrary one: the first pick is then
    # argmax ||u_i|| / h1norm_i, not index 0 (src/lib/ReducedBasis.py:129)
    for hn in (1, np.linspace(0.5, 2.0, M)):
        for mode in (RB.GREEDY_FOR_H10, RB.GREEDY_FOR_GALERKIN):
            g_r = RB.ReducedBasisGreedy(mode).build(6, sm, U, a, hn)
            g_f = RB.ReducedBasisGreedy(mode).build(6, sm, fs, a, hn)
            assert g_r.picks[0] == int(np.argmax(h1 / hn)) and g_f.picks == g_r.picks, (mode, g_f.picks, g_r.picks)
            np.testing.assert_allclose(g_f.max_errors, g_r.max_errors, rtol=1e-8, atol=1e-11)
    # geometries with a node-by-node edge refuse the factored form
    sm2 = SM.SolutionsManagerFEM((1, 2), 6)
    if not sm2._fem.expansion_is_linear:
        with pytest.raises(Exception):
            factored.ExpansionMap(sm2)


def test_rccl_single_rank_allgather(api):
    """The RCCL plumbing with a 1-rank communicator (a box has one GPU): id, init, all-gather, reduce."""
    from romhighcontrast_amd import _ffi, sweep
    SM, _ = api
    ctx = _ffi.get_context()
    uid = ctx.comm_unique_id()
    assert len(uid) == 128
    ctx.comm_init(uid, 0, 1)
    try:
        x = np.arange(1000, dtype=np.float64)
        src, dst = ctx.upload(x), ctx.alloc(1000)
        ctx.allgather(src, 0, dst, 0, 1000)
        assert np.array_equal(dst.download(), x)
        assert ctx.allreduce_host([3.5, -1.0], "max").tolist() == [3.5, -1.0]
        # overlapped form on the communication stream, both double-buffer slots
        dst2 = ctx.alloc(1000)
        for slot in (0, 1, 0):
            ctx.comm_wait_slot(slot)
            src.upload(x + slot)
            ctx.allgather_async(src, 0, dst2, 0, 1000, slot=slot)
            ctx.comm_wait(True)
            assert np.array_equal(dst2.download(), x + slot)
        sm = SM.SolutionsManagerFEM((2, 2), 8)
        a = 10.0 ** np.random.default_rng(3).uniform(0, 2, size=(5, 2, 2))
        U = sweep.RcclSweep(sm, 0, 1).generate_solutions_device(a)
        assert np.array_equal(U.numpy(), sm.generate_solutions(a))
        sm32 = SM.SolutionsManagerFEM((2, 2), 32)             # (compressed edges: linear expansion)
        fs = sweep.RcclSweep(sm32, 0, 1).generate_factored(a)  # the gathered block left in factored form
        assert fs.M == 5 and np.array_equal(fs.rows().numpy(), sm32.generate_solutions(a))
        # the step loop of bench.py on the real backend: compact vectors, one collective per group of 2 steps, 5 steps (the
        # last group incomplete); every gathered shard must expand to the rows of a plain sweep, bit for bit
        fem, M = sm32._fem, 5
        a_dev = ctx.upload(np.ascontiguousarray(a).reshape(M, -1))
        ref = sm32.generate_solutions(a)
        be = sweep.GpuStepBackend(ctx, fem, a_dev, M, 1, every=2)
        assert be.cstride < be.stride
        for step in range(5):
            k = sweep.run_step(be, step, 2)
        sweep.drain(be, 5, 2)
        assert be.parts == [1, 2] and k == 0
        assert np.array_equal(be.U_loc.download(shape=(M, fem.dim)), ref)
        for slot, part in ((0, 0), (1, 0), (1, 1)):
            Yfull = be.gathered_vectors(slot, 0, part, 0, M)
            rows = ctx.alloc(M * fem.dim)
            fem.expand(a_dev, M, Yfull, rows)
            ctx.solve_status()
            assert np.array_equal(rows.download(shape=(M, fem.dim)), ref), (slot, part)
        # the same on a MULTI-TILE geometry (the general tile-Cholesky path: 3 x 3 blocks, reduced matrix of several 64 x 64
        # tiles), M = 70 rows (not a multiple of anything): factored gather + the grouped step loop, groups of 3 over 7 steps
        sm3 = SM.SolutionsManagerFEM((3, 3), 64)
        fem3, M3 = sm3._fem, 70
        assert fem3.n_tiles > 1 and fem3.expansion_is_linear
        a3 = 10.0 ** np.random.default_rng(11).uniform(0, 4, size=(M3, 3, 3))
        ref3 = sm3.generate_solutions_device(a3, keep_interface_vectors=False).numpy()
        fs3 = sweep.RcclSweep(sm3, 0, 1).generate_factored(a3)
        assert fs3.M == M3 and np.array_equal(fs3.rows().numpy(), ref3)
        a3_dev = ctx.upload(np.ascontiguousarray(a3).reshape(M3, -1))
        be3 = sweep.GpuStepBackend(ctx, fem3, a3_dev, M3, 1, every=3)
        assert be3.cstride < be3.stride
        for step in range(7):
            k3 = sweep.run_step(be3, step, 3)
        sweep.drain(be3, 7, 3)
        assert np.array_equal(be3.U_loc.download(shape=(M3, fem3.dim)), ref3)
        for slot in (0, 1):
            for part in range(be3.parts[slot]):
                Yfull = be3.gathered_vectors(slot, 0, part, 0, M3)
                rows = ctx.alloc(M3 * fem3.dim)
                fem3.expand(a3_dev, M3, Yfull, rows)
                ctx.solve_status()
                assert np.array_equal(rows.download(shape=(M3, fem3.dim)), ref3), (slot, part)
    finally:
        ctx.comm_destroy()


def test_pickle_roundtrip_of_basis_and_manager(api):
    import pickle
    SM, RB = api
    sm = SM.SolutionsManagerFEM((2, 2), 6)
    a = 10.0 ** np.random.default_rng(5).uniform(0, 2, size=(6, 2, 2))
    U = sm.generate_solutions(a)
    rb = RB.ReducedBasisGreedy(RB.GREEDY_FOR_H10).build(3, sm, U, a, sm.H10norm(U))
    rb2 = pickle.loads(pickle.dumps(rb))
    sm2 = pickle.loads(pickle.dumps(sm))
    assert np.array_equal(rb2.basis, rb.basis)
    assert np.array_equal(sm2.generate_solutions(a[:2]), U[:2])


def test_g8_experiment_statistics(api):
    """The statistics loop of experiment() (HighContrast.py:144-214) on the reference's own sampler output:
    snapshots, the four default builders, and for n = 1..4 the five error records, against fixture g8."""
    SM, RB = api
    from src.experiments.HighContrast import experiment_statistics, get_a2test_and_train
    z = load_golden("g8_experiment.npz")
    sm, a, ahc = get_a2test_and_train((2, 2), [[(0, 0), (1, 1)], [(0, 1)]], 6, 2, 30, 7, method="lsq")
    assert np.array_equal(a, z["a"])
    builders = [RB.ReducedBasisRandom(), RB.ReducedBasisRandom(False), RB.ReducedBasisGreedy(greedy_for=RB.GREEDY_FOR_H10),
                RB.ReducedBasisGreedy(greedy_for=RB.GREEDY_FOR_GALERKIN)]
    import tempfile
    from src.experiments.HighContrast import get_data
    with tempfile.TemporaryDirectory() as tmp:
        data0, data_path = get_data(tmp)
        assert data0 == {}
        data = experiment_statistics(sm, a, builders, vn_max_dim=4, num_measurements=12, data_path=data_path)
        loaded, _ = get_data(tmp)  # joblib round trip of the reference's cache layout
        assert np.array_equal(loaded["solutions"], data["solutions"])
        assert np.array_equal(loaded[builders[2].name]["basis"].basis, data[builders[2].name]["basis"].basis)
    g = ro.Geometry((2, 2), 6)
    # rows with INFINIT_A blocks: kappa ~ 1e11 (reference self-consistency ~1e-6); the rest to 1e-11
    err = relh10(g, data["solutions"], z["solutions"])
    hard = (a == RB.INFINIT_A).any(axis=(1, 2))
    # Every row to the snapshot bound, the INFINIT_A ones included: at (2,2) every block touches the Dirichlet boundary, the
    # 1e10 blocks simply clamp their edges, and the extended-precision referee (tests/golden/referee_g8_inf.npz) puts the
    # reference's own rows within 1e-15 of the truth -- so do ours (round 3 kept 5e-5 here for want of that fixture).
    observed("g8: snapshots vs the reference's, ordinary parameters (rel H10)", err[~hard], SNAP_TOL)
    observed("g8: snapshots vs the reference's, INFINIT_A parameters (rel H10)", err[hard], SNAP_TOL)
    r8 = load_golden("referee_g8_inf.npz")
    assert np.array_equal(r8["rows"], np.flatnonzero(hard))
    observed("g8: INFINIT_A rows vs the refined truth (rel H10; the reference's own: <= 9.4e-16)",
             relh10(g, np.asarray(data["solutions"])[hard], r8["truth"]), 1e-14)
    for b in builders:
        key = b.name.replace(" ", "_").replace("$", "").replace("\\", "").replace("^", "").replace("{", "").replace("}", "")
        assert str(z["name_" + key]) == b.name
        np.testing.assert_allclose(np.array(data[b.name]["basis"].basis), z["basis_" + key],
                                   atol=1e-5 * np.abs(z["basis_" + key]).max())
        for n in range(1, 5):
            e = data[b.name]["errors"][n]
            # forward modelling / projection errors: BASELINE bound where the reference is itself accurate
            for f in ("forward_modeling", "projection"):
                ref = z[f"err_{key}_{n}_{f}"]
                observed(f"g8 {key} n={n} {f}: error records vs reference, ordinary parameters (BASELINE: 1e-10)",
                         np.abs(getattr(e, f) - ref)[~hard], 1e-10)
                # INFINIT_A test parameters: the projection is taken in the unit-coefficient inner product (well conditioned:
                # BASELINE bound); the Galerkin ROM solves (C A(a) C^T) c = C B with a = 1e10 blocks, whose condition number
                # reaches 1e10 once the basis holds a snapshot that lives in such a block -- two exact implementations (the
                # reference's own pair included) then differ by cond x eps in the coefficients and in the error record
                tol_f = 1e-10
                if f == "forward_modeling":
                    Qb = np.linalg.qr(np.asarray(data[b.name]["basis"].basis)[:n].T)[0].T
                    conds = [np.linalg.cond(Qb @ ro.stencil_apply(g, am, Qb).T) for am in a[hard]]
                    tol_f = max(1e-10, 1e-14 * max(conds))
                observed(f"g8 {key} n={n} {f}: error records vs reference, INFINIT_A parameters (bound {tol_f:.1e}"
                         + (" = 1e-14 cond(C A(a) C^T)" if tol_f > 1e-10 else ": BASELINE") + ")",
                         np.abs(getattr(e, f) - ref)[hard], tol_f)
                if f == "forward_modeling":
                    # ... and who is right: the 80-bit Galerkin truth (tests/referee.py) on each side's OWN inputs -- our basis
                    # rows + snapshots, the reference's basis rows + snapshots (fixture) -- against each side's record.  Ours
                    # must be within the BASELINE bar of its truth, or no further from it than 4 x the reference is from its own
                    import referee
                    t_us = referee.galerkin_truth_nested(g, a, np.asarray(data[b.name]["basis"].basis)[:n], np.asarray(data["solutions"]), [n])[n]
                    t_ref = referee.galerkin_truth_nested(g, a, z["basis_" + key][:n], z["solutions"], [n])[n]
                    d_us, d_ref = np.abs(getattr(e, f) - t_us)[hard], np.abs(ref - t_ref)[hard]
                    observed(f"g8 {key} n={n} forward_modeling, INFINIT_A parameters: the REFERENCE's records vs the 80-bit truth on its inputs (for the record)", d_ref, 1e-3)
                    observed(f"g8 {key} n={n} forward_modeling, INFINIT_A parameters: our records vs the 80-bit truth on our inputs, in units of max(1e-10, 4 x the reference's distance)",
                             d_us / np.maximum(1e-10, 4 * d_ref.max()), 1.0)
            # state estimation (src/lib/ReducedBasis.py:65-70) and the two parameter estimators (:72-86,
            # src/lib/Estimators.py:24-37): a least-squares fit through the (points x n) matrix E of basis values.
            # Bases that hold INFINIT_A snapshots make E nearly rank deficient (cond(E) up to 5e12 in this fixture),
            # so the fitted coefficients -- and everything computed from them -- move by cond(E) x the 1e-12
            # differences between any two exact solvers (the reference's own lsq / lsqsparse included).  The bound
            # is therefore 1e-9 where E is well conditioned and cond(E) x 1e-13 otherwise, relative to the size of
            # the reference record (the linear estimator multiplies the coefficients by a = 1e10).
            Eb = sm.evaluate_solutions(data["measurement_points"], np.asarray(data[b.name]["basis"].basis)[:n])
            tol = max(1e-9, 1e-13 * np.linalg.cond(Eb))
            for f in ("state_estimation", "parameter_estimation_inverse", "parameter_estimation_linear"):
                ref, got = z[f"err_{key}_{n}_{f}"], np.asarray(getattr(e, f))
                assert got.shape == ref.shape, (b.name, n, f)
                scale = max(1.0, np.abs(ref).max())
                gap = np.abs(got - ref).reshape(len(ref), -1).max(axis=1) / scale   # per test parameter
                observed(f"g8 {key} n={n} {f}: records vs reference, all parameters (bound = max(1e-9, 1e-13 cond(E)), cond(E) = {np.linalg.cond(Eb):.1e})",
                         gap, tol)


def test_plain_c_caller_end_to_end(tmp_path):
    """tests/c_abi/c_abi_smoke.c: a C99 program drives the C-ABI directly -- FE space, sweep, stencil residual of every
    snapshot, H10 norms, rom_greedy, rom_project_h10, rom_pod, and (round 4) the same basis stage on the factored block:
    rom_fem_energy_map, rom_h10norm_factored, rom_greedy_factored, rom_pod_factored against the row calls -- and checks the
    identities listed in its header."""
    import subprocess
    from test_host_logic import _build_c_caller
    exe, env = _build_c_caller(tmp_path)
    r = subprocess.run([exe], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = r.stdout.decode()
    assert r.returncode == 0 and out.rstrip().endswith("OK"), out

