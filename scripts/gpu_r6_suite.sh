#!/bin/bash
# gpurun payload: the whole GPU suite, the overlap test against the UNFIXED probe build (must fail), smoke, bench
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r06_pytest_gpu.log
echo "== pytest -m gpu" | tee $L
timeout 3000 python -m pytest tests/ -q -m gpu -x --durations=10 2>&1 | tail -30 | tee -a $L
echo "== tests/test_overlap_gpu.py against build/wc/libpmn_hip_nosettle.so (the fix compiled out: MUST fail)" | tee -a $L
timeout 600 python - <<'PY' 2>&1 | tail -12 | tee -a $L
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from patchmatchnet_amd import _lib
_lib.LIB_PATH = os.path.abspath("build/wc/libpmn_hip_nosettle.so")
import pytest
rc = pytest.main(["tests/test_overlap_gpu.py", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"])
print("overlap test on the unfixed build: exit code", int(rc), "(expected 1)")
PY
echo "== smoke" | tee -a $L
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $L
