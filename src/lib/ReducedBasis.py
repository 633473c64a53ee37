"""Re-export of romhighcontrast_amd.lib.ReducedBasis under the reference's import path."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)

from romhighcontrast_amd.lib.ReducedBasis import *  # noqa: F401,F403,E402
from romhighcontrast_amd.lib import ReducedBasis as _impl  # noqa: E402

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
