"""The sampling / error-definition part of the reference's src/experiments/HighContrast.py under its import
path (plots and the results-folder driver are out of scope; see romhighcontrast_amd/experiments.py)."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)

from romhighcontrast_amd.experiments import (MachinePrecision, TypeOfProblems, calculate_relative_error,  # noqa: E402,F401
                                             calculate_time, experiment_statistics, get_a2test_and_train, get_data,
                                             get_full_a, save_data)
from romhighcontrast_amd.lib.ReducedBasis import (GREEDY_FOR_GALERKIN, GREEDY_FOR_H10, INFINIT_A,  # noqa: E402,F401
                                                  ReducedBasisGreedy, ReducedBasisRandom)
from romhighcontrast_amd.lib.SolutionsManagers import SolutionsManager, SolutionsManagerFEM  # noqa: E402,F401
