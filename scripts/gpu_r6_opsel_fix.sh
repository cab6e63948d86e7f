#!/bin/bash
# After the op_sel fix: the instruction matrix, the library-level reproducer, the overlap test (both configurations) on the product
# build and on the build with the pins compiled out, and the kernel parity tests -> gpurun_out/r06_opsel_fix.log
mkdir -p gpurun_out build
export TMPDIR=/tmp
L=gpurun_out/r06_opsel_fix.log
: > $L
bash scripts/gpu_r6_pkmatrix.sh > /dev/null 2>&1
grep "forms with\|v_pk_mov" gpurun_out/r06_pk_opsel_matrix.log | cut -c1-200 >> $L
echo "== library reproducer" >> $L
for lib in patchmatchnet_amd/csrc/libpmn_hip.so build/wc/libpmn_hip_nosettle.so; do
  timeout 120 build/library_overlap_repro $lib 24 400 2>&1 | grep -av "amdgpu.ids" | sed 's/ (ABI.*when the victims finished)//' >> $L
done
echo "== tests/test_overlap_gpu.py, product build" >> $L
timeout 900 python -m pytest tests/test_overlap_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | grep -av "Warning\|warnings.warn\|^$" | tail -4 >> $L
echo "== tests/test_overlap_gpu.py, build/wc/libpmn_hip_nosettle.so (MUST fail, both configurations)" >> $L
timeout 900 python - >> $L 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from patchmatchnet_amd import _lib
_lib.LIB_PATH = os.path.abspath("build/wc/libpmn_hip_nosettle.so")
import pytest
rc = pytest.main(["tests/test_overlap_gpu.py", "-q", "-m", "gpu", "-p", "no:cacheprovider", "--tb=line"])
print("exit code", int(rc))
PY
echo "== kernel parity + plans" >> $L
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_plan_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -av "Warning\|warnings.warn\|^$" | tail -4 >> $L
grep -av "amdgpu.ids\|^  warnings\|UserWarning" $L | cut -c1-260
