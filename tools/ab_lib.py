"""Development aid: run a script against another build of the library (A/B of two kernel versions on one GPU box).
    python tools/ab_lib.py llamagen_amd/liblgen_hip_prev.so bench.py --no-cpu-baseline ...
"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
