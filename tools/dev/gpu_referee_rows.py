"""Dev probe (GPU box): the floating-block rows of C4 (row 5) and of fixture g4/b33 against the extended-precision
truths of tests/golden/make_referee.py, for the default algorithm and its A/B switches.  Prints relative H10 errors."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rom_oracle as ro  # noqa: E402
from romhighcontrast_amd.lib import SolutionsManagers as SM  # noqa: E402

out = {}
for name, blocks, N in (("c4_row5", (3, 3), 171), ("b33_row7", (3, 3), 11)):
    z = np.load(os.path.join(ROOT, "tests", "golden", f"referee_{name}.npz"))
    g = ro.Geometry(blocks, N)
    a = z["a"][None]
    for tag, env in (("default", {}), ("nocompress", {"ROMHC_NO_COMPRESS": "1"}), ("tol1e-17", {"ROMHC_COMPRESS_TOL": "1e-17"}),
                     ("nopreelim", {"ROMHC_NO_PREELIM": "1"})):
        for k in ("ROMHC_NO_COMPRESS", "ROMHC_COMPRESS_TOL", "ROMHC_NO_PREELIM"):
            os.environ.pop(k, None)
        os.environ.update(env)
        sm = SM.SolutionsManagerFEM(blocks, N)
        U = sm.generate_solutions(a)
        e_truth = float(ro.H10norm(g, U - z["truth"])[0] / ro.H10norm(g, z["truth"][None])[0])
        e_slu = float(ro.H10norm(g, U - z["superlu"])[0] / ro.H10norm(g, z["truth"][None])[0])
        print(f"{name} {tag}: gpu vs truth {e_truth:.3e}   gpu vs superlu {e_slu:.3e}   (superlu vs truth {float(z['err_superlu_vs_truth']):.3e})", flush=True)
        out[f"{name}_{tag}"] = U[0]
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "referee_gpu_rows.npz"), **out)
