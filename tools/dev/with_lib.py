"""Runs a script of this repository against a variant build of libromhc (dev tool):
python tools/dev/with_lib.py path/to/libromhc_variant.so script.py [args ...]"""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romhighcontrast_amd import _ffi
_ffi.load_library(os.path.abspath(sys.argv[1]))
script = sys.argv[2]
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")
