import os
import sys

import pytest

# No GPU_MAX_HW_QUEUES override here (round 5 set it to 1 for the whole suite, which hid the very failure it was about): replayed
# forwards must be bit-identical to the eager forward on the runtime's default hardware queues, and the suite runs on those.

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
