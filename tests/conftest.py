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
