"""Extended-precision referee for the "floating block" rows.

A block that touches no Dirichlet side and dominates its neighbours by a factor c has its plateau level fixed by fluxes
that are 1/c of the matrix entries: two backward-stable fp64 direct solvers agree on such a row to ~c * eps * N only
(the reference's own lsq / lsqsparse pair differs by 8.5e-6 at (3,3)/N=11, fixture g4).  To decide who is right this
script builds a TRUTH that does not depend on any fp64 solver's rounding:

    x_0 = SuperLU solve (scipy.sparse.linalg.splu of the oracle's CSC matrix: the reference's call at
          src/lib/SolutionsManagers.py:31), then  x_{k+1} = x_k + LU^-1 (b - A x_k)

with the residual b - A x_k evaluated in 80-bit long double in EDGE form
    (A x)_i = sum_j w_ij (x_i - x_j) + (boundary weights) x_i
(differences of neighbouring values first: inside the floating block they are ~1/c of the values, so the products
w_ij (x_i - x_j) carry no cancellation).  The correction contracts by ~kappa * eps per step; iteration stops when it
stalls.  The final correction norm bounds the distance to the exact solution of the fp64-assembled system.

Outputs (committed): tests/golden/referee_c4_row5.npz (C4 training row 5) and tests/golden/referee_g4_floating.npz (the
floating rows of fixture g4) holding the truth rows, err_superlu_vs_truth, the errors of the reference's own two outputs
against the truth (relative H^1_0) and the refinement history.

Run in the build container:  python tests/golden/make_referee.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rom_oracle as ro  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, 'tests'))
from referee import LD, h10_ld, referee  # noqa: E402


def main():
    out = os.path.dirname(os.path.abspath(__file__))
    import bench
    # C4 training row 5: centre block of (3,3)/N=171 at 1e8 among ones
    a = bench.workload_parameters("c4", (3, 3), 16)[5]
    print("C4 row 5", a.tolist(), flush=True)
    g, truth, x0, err, hist = referee((3, 3), 171, a)
    print("C4 row 5: err_superlu_vs_truth =", err)
    # (the SuperLU row itself is not stored: the oracle reproduces it; 2 MB of truth is enough)
    np.savez_compressed(os.path.join(out, "referee_c4_row5.npz"), a=a, truth=truth,
                        err_superlu_vs_truth=err, history=np.array(hist))
    # fixture g4: the floating-block rows of the reference's own outputs ((3,3)/N=11 and (4,4)/N=8, one interior block at
    # INFINIT_A = 1e10): the rows where the reference's two solvers disagree
    z = np.load(os.path.join(out, "g4_contrast.npz"))
    rec = {}
    for name in ("b33", "b44"):
        blocks, N = tuple(int(v) for v in z[f"{name}_blocks"]), int(z[f"{name}_N"])
        g0 = ro.Geometry(blocks, N)
        gap = ro.H10norm(g0, z[f"{name}_U_lsqsparse"] - z[f"{name}_U"]) / ro.H10norm(g0, z[f"{name}_U"])
        row = int(np.argmax(gap))
        a = z[f"{name}_a"][row]
        print(name, "row", row, a.tolist(), "reference self-gap", gap[row], flush=True)
        g, truth, x0, err, hist = referee(blocks, N, a)
        tl = truth.astype(LD)
        e_lsq = float(h10_ld(g, z[f"{name}_U"][row].astype(LD) - tl) / h10_ld(g, tl))
        e_sp = float(h10_ld(g, z[f"{name}_U_lsqsparse"][row].astype(LD) - tl) / h10_ld(g, tl))
        print(f"{name} row {row}: reference lsq vs truth {e_lsq:.3e}, reference lsqsparse vs truth {e_sp:.3e}, oracle SuperLU {err:.3e}")
        rec.update({f"{name}_row": row, f"{name}_a": a, f"{name}_truth": truth, f"{name}_err_superlu_vs_truth": err,
                    f"{name}_err_ref_lsq_vs_truth": e_lsq, f"{name}_err_ref_lsqsparse_vs_truth": e_sp,
                    f"{name}_history": np.array(hist)})
    np.savez_compressed(os.path.join(out, "referee_g4_floating.npz"), **rec)
    # fixture g8: the rows of the experiment's training set that have blocks at INFINIT_A ((2,2)/N=6: no floating block, but
    # kappa ~ 1e11): the truth of every such row and the distance of the reference's own row from it
    z = np.load(os.path.join(out, "g8_experiment.npz"), allow_pickle=True)
    a8 = z["a"]
    rows = np.flatnonzero((a8 == 1e10).reshape(len(a8), -1).any(axis=1))
    truths, e_ref, e_slu = [], [], []
    for r in rows:
        g, truth, x0, err, hist = referee((2, 2), 6, a8[r], verbose=False)
        tl = truth.astype(LD)
        truths.append(truth)
        e_slu.append(err)
        e_ref.append(float(h10_ld(g, z["solutions"][r].astype(LD) - tl) / h10_ld(g, tl)))
    print("g8 INFINIT_A rows", rows.tolist(), "reference vs truth", np.array(e_ref), "SuperLU vs truth", np.array(e_slu))
    np.savez_compressed(os.path.join(out, "referee_g8_inf.npz"), rows=rows, truth=np.array(truths),
                        err_ref_vs_truth=np.array(e_ref), err_superlu_vs_truth=np.array(e_slu))


if __name__ == "__main__":
    main()
