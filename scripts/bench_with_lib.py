#!/usr/bin/env python
"""A/B helper: run bench.py against ANOTHER build of libpmn_hip.so (the product has no such switch; a process loads one library).
    python scripts/bench_with_lib.py build/pw/libpmn_hip_g2.so --eager --steps 24 --no-cpu-baseline ..."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from patchmatchnet_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
