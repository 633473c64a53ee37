"""Random small FE spaces: the projector / ROM / orthonormalisation / evaluation calls of the host API against the oracle
(dev probe; GPU).  env: CASES (40), SEED (0), CMAX (5)."""
import os, sys, logging
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romhighcontrast_amd.lib import SolutionsManagers as SM, ReducedBasis as RB
from oracle import rom_oracle as ro
logging.disable(logging.WARNING)
rng = np.random.default_rng(int(os.environ.get("SEED", "0")))
worst = {}
def rec(name, v, info=""):
    v = float(v)
    if v > worst.get(name, (-1.0, ""))[0]:
        worst[name] = (v, info)
def relh10(g, A, B):
    return ro.H10norm(g, A - B) / np.maximum(ro.H10norm(g, B), 1e-300)
for case in range(int(os.environ.get("CASES", "40"))):
    blocks = (int(rng.integers(1, 4)), int(rng.integers(1, 4)))
    N = int(rng.integers(3, 26))
    M = int(rng.integers(4, 60))
    cexp = float(rng.uniform(0.3, float(os.environ.get("CMAX", "5"))))
    a = 10.0 ** rng.uniform(0, cexp, size=(M,) + blocks)
    tag = f"case {case}: blocks {blocks} N {N} M {M} contrast 1e{cexp:.1f}"
    sm = SM.SolutionsManagerFEM(blocks, N)
    g = ro.Geometry(blocks, N)
    U = sm.generate_solutions(a)
    Uo = ro.generate_solutions(g, a)
    rec("snapshots vs oracle (rel H10)", relh10(g, U, Uo).max(), tag)
    nb = int(rng.integers(1, min(M, 12) + 1))
    idx = rng.choice(M, nb, replace=False)
    kind = int(rng.integers(0, 3))
    if kind == 0:
        C = Uo[idx]                      # snapshots as basis (raw, badly scaled)
    elif kind == 1:
        C = ro.orthonormalize_base(Uo[idx])
    else:
        C = rng.standard_normal((nb, g.dim if hasattr(g, "dim") else U.shape[1]))
    # orthonormalisation
    Qg = RB.orthonormalize_base(C)
    Qo = ro.orthonormalize_base(C)
    sgn = np.sign(np.sum(Qg * Qo, axis=1))
    # rows behind the numerical rank of C are rounding noise in LAPACK's QR as well (snapshots of a two-block space: rank 3):
    # compared are the rows up to the first diagonal entry of R below 1e-10 of the largest; all rows must be orthonormal
    rd = np.abs(np.diag(np.linalg.qr(C.T)[1]))
    lead = int(np.argmax(rd < 1e-10 * rd.max())) if (rd < 1e-10 * rd.max()).any() else len(rd)
    rec("orthonormalize_base vs oracle (rows up to sign and up to the numerical rank, abs)", np.abs(Qg * sgn[:, None] - Qo)[:lead].max() if lead else 0.0,
        tag + f" nb {nb} kind {kind} rank {lead}")
    rec("orthonormalize_base: orthonormality of all rows", np.abs(Qg @ Qg.T - np.eye(len(Qg))).max(), tag + f" nb {nb} kind {kind}")
    # projection and ROM
    def both(name, fg, fo):
        rg = ro_ = None
        try:
            rg = fg()
        except np.linalg.LinAlgError as e:
            rg = e
        try:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ro_ = fo()
        except np.linalg.LinAlgError as e:
            ro_ = e
        ge, oe = isinstance(rg, Exception), isinstance(ro_, Exception)
        if ge or oe:
            rec(f"{name}: LinAlgError on one side only (1 = ours only, 2 = oracle only)", (1.0 if ge else 2.0) if ge != oe else 0.0,
                tag + f" nb {nb} kind {kind} cond(C A1 C^T) {np.linalg.cond(C @ ro.stencil_apply(g, np.ones(blocks), C).T):.1e}")
            return
        rec(f"{name} vs oracle (abs / max|U|), basis kind {kind}", np.abs(rg - ro_).max() / np.abs(Uo).max(), tag + f" nb {nb}")
    both("project_solutions", lambda: sm.project_solutions(U, C), lambda: ro.project_solutions(g, Uo, C))
    both("generate_fm_solutions", lambda: sm.generate_fm_solutions(a, C), lambda: ro.generate_fm_solutions(g, a, C))
    # evaluation
    lo, hi = max(sm.points_c[0], sm.points_r[0]), min(sm.points_c[-1], sm.points_r[-1])
    pts = rng.uniform(lo + 1e-9, hi - 1e-9, size=(7, 2))   # (the reference indexes the vertex arrays without bounds: points between the first and last vertex)
    Eg = sm.evaluate_solutions(pts, U)
    Eo = ro.evaluate_solutions(g, pts, Uo)
    rec("evaluate_solutions vs oracle (abs / max|U|)", np.abs(Eg - Eo).max() / np.abs(Uo).max(), tag)
    del sm
for k, (v, info) in sorted(worst.items()):
    print(f"{k:70s} {v:.3e}   {info[:160]}")
