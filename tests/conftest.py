import os
import sys

import pytest

# The GPU suite calls eval.main() / bench pieces IN PROCESS: the one-hardware-queue setting those programs make for themselves
# (eval.py / bench.py, DESIGN_LESSONS.md lesson 45: a forward replayed as a HIP graph beside other GPU work of the process is only
# bit-identical to the eager forward on ONE hardware queue) has to be in the environment before torch initialises HIP.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are selected with -m gpu; when selected on a box without a GPU they FAIL loudly inside the
    test (the product refuses to run without its HIP library / device), they are never silently skipped."""
    return
