import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the shared library is a build artefact (git-ignored): build it when a fresh checkout has none
    lib = os.path.join(ROOT, "romhighcontrast_amd", "csrc", "libromhc.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_h10(oracle_geom, U, Uref):
    """max_i ||U_i - Uref_i||_{H10} / ||Uref_i||_{H10} using the oracle norm."""
    from oracle import rom_oracle as ro
    return float(np.max(ro.H10norm(oracle_geom, np.asarray(U) - np.asarray(Uref)) / ro.H10norm(oracle_geom, Uref)))


# ---- observed parity values -------------------------------------------------------------------------------------------
# Every parity assertion that goes through observed() leaves its measured value next to its bound in the terminal
# summary -- also when it passes -- so a green log shows HOW far inside the bar the run was.
_OBSERVED = []


def observed(name, value, bound, *, detail=""):
    """assert value <= bound, recording both.  `value` may be an array (its maximum is recorded)."""
    v = float(np.max(value)) if np.size(value) else 0.0
    _OBSERVED.append((name, v, float(bound)))
    assert v <= bound, f"{name}: observed {v:.3e} > bound {bound:.3e} {detail}"
    return v


def pytest_terminal_summary(terminalreporter):
    if not _OBSERVED:
        return
    terminalreporter.write_sep("-", "observed parity values (max over the assertion's entries) vs bound")
    for name, v, b in _OBSERVED:
        terminalreporter.write_line(f"  {name:<78s} {v:10.3e}  <= {b:8.1e}")
