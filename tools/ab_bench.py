"""A/B of library builds on the GPU box: python tools/ab_bench.py <lib.so> [bench.py args...] runs bench.py against that build
(instantsplat_amd/lib/variants/*.so travel with the snapshot; they are git-ignored).  Measurement helper, not product code."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instantsplat_amd import _lib
lib = os.path.abspath(sys.argv[1])
import ctypes
_probe = ctypes.CDLL(lib)
for name in list(_lib._SIGNATURES):   # an older build may lack newer optional entry points
    if not hasattr(_probe, name):
        del _lib._SIGNATURES[name]
_lib._use_library_for_testing(lib)
if os.environ.get("GS_MIN_UNITS"):   # A/B of the backward's unit length (mi355gs_tune_min_units)
    _lib.lib().mi355gs_tune_min_units(int(os.environ["GS_MIN_UNITS"]))
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
