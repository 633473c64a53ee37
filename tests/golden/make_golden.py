#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by importing the REFERENCE unmodified.

Runs ONLY in the build container (the reference tree does not travel to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference imports ``pathos.multiprocessing`` (not installed, no network); it is never
exercised with ``num_cores=1`` (src/lib/SolutionsManagers.py:51), so a two-line scratch stub
outside both trees stands in for the missing *third-party* module.
Nothing from the reference is copied: the fixtures are inputs + the reference's outputs.
"""
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    stub = os.path.join(tempfile.gettempdir(), "romhc_refstub")
    os.makedirs(os.path.join(stub, "pathos"), exist_ok=True)
    open(os.path.join(stub, "pathos", "__init__.py"), "w").close()
    with open(os.path.join(stub, "pathos", "multiprocessing.py"), "w") as f:
        f.write("from multiprocessing import Pool, cpu_count\n")
    sys.path[:0] = [stub, REF, os.path.join(REF, "src")]
    sys.dont_write_bytecode = True
    import src.lib.SolutionsManagers as SM
    import src.lib.ReducedBasis as RB
    return SM, RB


def stencil_from_dense(A, nr, nc):
    """Read the 5-point stencil arrays back out of a dense reference matrix."""
    idx = np.arange(nr * nc).reshape(nr, nc)
    diag = A[idx, idx]
    east = A[idx[:, :-1], idx[:, 1:]]
    north = A[idx[:-1, :], idx[1:, :]]
    nnz = int(np.count_nonzero(A))
    return diag, east, north, nnz


def main():
    SM, RB = _import_reference()
    rng = np.random.default_rng(20240807)

    # ---- G1: (2,2)/N=10 basics -------------------------------------------------------
    sm = SM.SolutionsManagerFEM((2, 2), 10)
    a = np.array([[[1, 1], [1, 1]], [[1, 1], [1, 100]], [[1, 2], [3, 4]]], dtype=float)
    U = sm.generate_solutions(a)
    pts = np.array([[0.25, -0.5], [-0.3, 0.7], [0.0, 0.0], [0.93, 0.11], [-0.999, -0.999]])
    A_gen = np.einsum("pqij,pq->ij", sm.A_preassembled, a[2])
    d, e, n, nnz = stencil_from_dense(A_gen, sm.nr_inner_vertices, sm.nc_inner_vertices)
    A1 = np.einsum("pqij,pq->ij", sm.A_preassembled, a[0])
    np.savez_compressed(
        os.path.join(OUT, "g1_basic.npz"), blocks=(2, 2), N=10, a=a, U=U, B_total=sm.B_total,
        H10=sm.H10norm(U), l2=sm.l2norm(U), points=pts, evals=sm.evaluate_solutions(pts, U),
        diag=d, east=e, north=n, nnz=nnz, nnz_unit=int(np.count_nonzero(A1)),
        points_c=sm.points_c, points_r=sm.points_r)

    # ---- G2: config C1 = (2,2)/N=16, checkerboard 4x4 sweep, both solver methods -----
    vals = np.array([1.0, 10.0, 100.0, 1000.0])
    grid = np.array([[v0, v1] for v0 in vals for v1 in vals])
    groups = [[(0, 0), (1, 1)], [(0, 1), (1, 0)]]
    a_c1 = np.ones((16, 2, 2))
    for gi, members in enumerate(groups):
        for (p, q) in members:
            a_c1[:, p, q] = grid[:, gi]
    sm = SM.SolutionsManagerFEM((2, 2), 16, method="lsq")
    U_lsq = sm.generate_solutions(a_c1)
    sm.method = "lsqsparse"
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        U_sp = sm.generate_solutions(a_c1)
    np.savez_compressed(os.path.join(OUT, "g2_c1.npz"), blocks=(2, 2), N=16, a=a_c1, U_lsq=U_lsq,
                        U_lsqsparse=U_sp, H10=sm.H10norm(U_lsq), grid=grid)

    # ---- G3: rectangular geometries (orientation) -------------------------------------
    out = {}
    for name, blocks, N in (("r23", (2, 3), 5), ("r32", (3, 2), 4)):
        sm = SM.SolutionsManagerFEM(blocks, N)
        a = 10.0 ** rng.uniform(0, 2, size=(4,) + blocks)
        a[0] = np.arange(1, blocks[0] * blocks[1] + 1).reshape(blocks)
        U = sm.generate_solutions(a)
        A_gen = np.einsum("pqij,pq->ij", sm.A_preassembled, a[0])
        d, e, n, nnz = stencil_from_dense(A_gen, sm.nr_inner_vertices, sm.nc_inner_vertices)
        out.update({f"{name}_blocks": blocks, f"{name}_N": N, f"{name}_a": a, f"{name}_U": U,
                    f"{name}_H10": sm.H10norm(U), f"{name}_diag": d, f"{name}_east": e,
                    f"{name}_north": n, f"{name}_B": sm.B_total})
    np.savez_compressed(os.path.join(OUT, "g3_rect.npz"), **out)

    # ---- G4: high contrast, incl. INFINIT_A blocks ------------------------------------
    out = {}
    for name, blocks, N in (("b22", (2, 2), 16), ("b33", (3, 3), 11), ("b44", (4, 4), 8)):
        sm = SM.SolutionsManagerFEM(blocks, N)
        a = 10.0 ** rng.uniform(0, 8, size=(8,) + blocks)
        a[5] = 1.0
        a[5].flat[0] = RB.INFINIT_A
        a[6] = 1.0
        a[6].flat[-1] = RB.INFINIT_A
        a[6].flat[1] = RB.INFINIT_A
        a[7] = 1.0
        a[7][blocks[0] // 2, blocks[1] // 2] = RB.INFINIT_A  # floating block for 3x3 / 4x4
        U = sm.generate_solutions(a)
        sm.method = "lsqsparse"
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            U_sp = sm.generate_solutions(a)
        out.update({f"{name}_blocks": blocks, f"{name}_N": N, f"{name}_a": a, f"{name}_U": U,
                    f"{name}_U_lsqsparse": U_sp, f"{name}_H10": sm.H10norm(U)})
    np.savez_compressed(os.path.join(OUT, "g4_contrast.npz"), **out)

    # ---- G5: projectors ---------------------------------------------------------------
    sm = SM.SolutionsManagerFEM((2, 2), 8)
    M = 12
    a = 10.0 ** rng.uniform(0, 3, size=(M, 2, 2))
    U = sm.generate_solutions(a)
    out = dict(blocks=(2, 2), N=8, a=a, U=U)
    for n in (0, 1, 5, 10):
        if n == 0:
            C = np.empty((0, 0))
        else:
            C = RB.orthonormalize_base(rng.standard_normal((n, sm.vspace_dim)))
        out[f"C{n}"] = C
        out[f"proj{n}"] = sm.project_solutions(U, C)
        out[f"fm{n}"] = sm.generate_fm_solutions(a, C)
    # a basis made of actual snapshots (the case the greedy produces)
    Csnap = RB.orthonormalize_base(U[:4])
    out["Csnap"] = Csnap
    out["proj_snap"] = sm.project_solutions(U, Csnap)
    out["fm_snap"] = sm.generate_fm_solutions(a, Csnap)
    np.savez_compressed(os.path.join(OUT, "g5_projectors.npz"), **out)

    # ---- G6: greedy (both modes) ------------------------------------------------------
    sm = SM.SolutionsManagerFEM((2, 2), 8)
    M, n = 40, 6
    a = 10.0 ** rng.uniform(0, 4, size=(M, 2, 2))
    U = sm.generate_solutions(a)
    h1 = sm.H10norm(U)
    out = dict(blocks=(2, 2), N=8, a=a, U=U, h1=h1, n=n)
    for tag, mode in (("h10", RB.GREEDY_FOR_H10), ("gal", RB.GREEDY_FOR_GALERKIN)):
        rb = RB.ReducedBasisGreedy(greedy_for=mode).build(n=n, sm=sm, solutions2train=U, a2train=a,
                                                         solutions2train_h1norm=h1)
        picks = [int(np.flatnonzero((a == ai).all(axis=(1, 2)))[0]) for ai in rb.a]
        # error curves: relative H10 projection / forward-modelling error for the first m vectors
        errs_proj, errs_fm = [], []
        for m in range(1, n + 1):
            sub = rb[:m]
            sub.orthonormalize()
            errs_proj.append(sm.H10norm(sub.projection(sm, U) - U) / h1)
            errs_fm.append(sm.H10norm(sub.forward_modeling(sm, a) - U) / h1)
        rb.orthonormalize()
        out.update({f"{tag}_basis": rb.basis, f"{tag}_picks": np.array(picks),
                    f"{tag}_errs_proj": np.array(errs_proj), f"{tag}_errs_fm": np.array(errs_fm)})
    np.savez_compressed(os.path.join(OUT, "g6_greedy.npz"), **out)

    # ---- G7: PCA / Random -------------------------------------------------------------
    from sklearn.decomposition import PCA
    sm = SM.SolutionsManagerFEM((2, 2), 8)
    M, n = 30, 5
    a = 10.0 ** rng.uniform(0, 3, size=(M, 2, 2))
    a[3, 0, 1] = RB.INFINIT_A
    a[11, 1, 1] = RB.INFINIT_A
    U = sm.generate_solutions(a)
    pca = PCA(n_components=n, svd_solver="full").fit(U[[i for i in range(M) if i not in (3, 11)]])
    out = dict(blocks=(2, 2), N=8, a=a, U=U, n=n, sigma=pca.singular_values_, comps=pca.components_)
    for flag in (True, False):
        rb = RB.ReducedBasisPCA(add_inf_solutions=flag).build(n=n, sm=sm, solutions2train=U, a2train=a)
        out[f"pca_basis_{int(flag)}"] = rb.basis
        out[f"pca_a_{int(flag)}"] = np.array(rb.a)
        rr = RB.ReducedBasisRandom(add_inf_solutions=flag).build(n=n, sm=sm, solutions2train=U, a2train=a)
        out[f"rnd_basis_{int(flag)}"] = rr.basis
        out[f"rnd_a_{int(flag)}"] = np.array(rr.a)
    np.savez_compressed(os.path.join(OUT, "g7_pca_random.npz"), **out)

    # ---- G9: largest dense-feasible cross-check, (2,2)/N=32 -> norms / checksums only --
    sm = SM.SolutionsManagerFEM((2, 2), 32, method="lsq")
    a = 10.0 ** rng.uniform(0, 2, size=(3, 2, 2))
    U = sm.generate_solutions(a)
    probe = rng.integers(0, sm.vspace_dim, size=64)
    np.savez_compressed(os.path.join(OUT, "g9_n32.npz"), blocks=(2, 2), N=32, a=a, H10=sm.H10norm(U),
                        l2=sm.l2norm(U), sums=U.sum(axis=1), probe=probe, U_probe=U[:, probe])
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()


def make_g8_experiment():
    """G8: the reference's experiment() end to end from a scratch copy of the tree (src/config.py mkdirs next
    to src/, which the read-only mount forbids): sampler outputs + the per-dimension error records."""
    import shutil
    import subprocess
    scratch = os.path.join(tempfile.gettempdir(), "romhc_refcopy")
    if os.path.exists(scratch):
        shutil.rmtree(scratch)
    shutil.copytree(REF, scratch)
    code = r'''
import sys, os, numpy as np
sys.path[:0] = [os.path.join(os.environ["STUB"]), os.environ["SCR"], os.path.join(os.environ["SCR"], "src")]
import matplotlib
matplotlib.use("Agg")
from src.experiments import HighContrast as HC
from lib.ReducedBasis import ReducedBasisGreedy, ReducedBasisRandom, GREEDY_FOR_H10, GREEDY_FOR_GALERKIN
builders = [ReducedBasisRandom(), ReducedBasisRandom(False), ReducedBasisGreedy(greedy_for=GREEDY_FOR_H10),
            ReducedBasisGreedy(greedy_for=GREEDY_FOR_GALERKIN)]
kw = dict(mesh_discretization_per_dim=6, diff_coef_refinement=2, vn_max_dim=4, num_measurements=12,
          blocks_geometry=(2, 2), high_contrast_blocks=[[(0, 0), (1, 1)], [(0, 1)]], max_num_samples_offline=30,
          seed=7, method="lsq")
sm, data, a, ahc = HC.experiment("g8", reduced_basis_builders=builders, recalculate=True, verbose=False, **kw)
out = dict(a=a, a_high_contrast=ahc, solutions=data["solutions"], h1=data["solutions_H1norm"])
for b in builders:
    key = b.name.replace(" ", "_").replace("$", "").replace("\\", "").replace("^", "").replace("{", "").replace("}", "")
    out["name_" + key] = np.array(b.name)
    out["basis_" + key] = np.array(data[b.name]["basis"].basis)
    for n in range(1, 5):
        e = data[b.name]["errors"][n]
        for f in e._fields:
            out[f"err_{key}_{n}_{f}"] = np.array(getattr(e, f))
# sampler alone with two more seeds / shapes
for tag, args in (("s1", ((3, 3), [[(1, 1)]], 4, 3, 25, 42)), ("s2", ((2, 2), [[(0, 0)], [(1, 1)], [(0, 1), (1, 0)]], 4, 1, 20, 3))):
    sm2, a2, ahc2 = HC.get_a2test_and_train(*args)
    out[tag + "_a"], out[tag + "_ahc"] = a2, ahc2
np.savez_compressed(os.environ["OUTF"], **out)
'''
    stub = os.path.join(tempfile.gettempdir(), "romhc_refstub")
    env = dict(os.environ, STUB=stub, SCR=scratch, OUTF=os.path.join(OUT, "g8_experiment.npz"),
               PYTHONDONTWRITEBYTECODE="1")
    subprocess.check_call([sys.executable, "-c", code], env=env, cwd=scratch)
    shutil.rmtree(scratch)
    print("g8 written")


if __name__ == "__main__" and os.environ.get("ROMHC_G8", "1") == "1":
    make_g8_experiment()
