#!/bin/bash
# Probe builds for DESIGN_LESSONS.md lesson 46 (output build/wc/libpmn_hip_<variant>.so; only gather_corr.hip differs):
#   nosettle  -DPMN_NO_SETTLE: lesson 46's fix compiled out -- tests/test_overlap_gpu.py must FAIL on it (scripts/gpu_r6_suite.sh)
# (The builds that located the defect -- -DPMN_DBG_NEIGHBOR / _LANE / _BLEND, -DPMN_SETTLE_X/_TAIL/_W, -O1, -amdgpu-waitcnt-forcezero --
#  need scripts/experiments/source_switches/attribution_and_probe_switches.patch applied first; their results are in
#  profiles/r06_overlap/r06_variants.log and r06_fixcheck.log.)
set -e
cd "$(dirname "$0")/.."
CS=patchmatchnet_amd/csrc
make -s -C $CS -j8
mkdir -p build/wc
FLAGS="-std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -I$CS"
OTHERS=$(ls $CS/*.o | grep -v '\.x\.o' | grep -v gather_corr.o)
build() { # name flags...
  local name=$1; shift
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $CS/gather_corr.hip -o build/wc/gather_corr_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/wc/libpmn_hip_$name.so build/wc/gather_corr_$name.o $OTHERS
  echo "built build/wc/libpmn_hip_$name.so"
}
build nosettle -O3 -DPMN_NO_SETTLE
