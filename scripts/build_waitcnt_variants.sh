#!/bin/bash
# Probe builds for DESIGN_LESSONS.md lesson 46 (output build/wc/libpmn_hip_<variant>.so; only gather_corr.hip differs):
#   fz     -mllvm -amdgpu-waitcnt-forcezero : an s_waitcnt 0 after every instruction -- if the overlap corruption disappears, some wait
#          the compiler left out (or a hazard its tables lack) is what the co-running kernel exposes
#   O1     the same source at -O1 (different schedule, same semantics)
set -e
cd "$(dirname "$0")/.."
CS=patchmatchnet_amd/csrc
make -s -C $CS -j8
mkdir -p build/wc
FLAGS="-std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -I$CS"
OTHERS=$(ls $CS/*.o | grep -v '\.x\.o' | grep -v gather_corr.o)
build() { # name opt flags...
  local name=$1; shift
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $CS/gather_corr.hip -o build/wc/gather_corr_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/wc/libpmn_hip_$name.so build/wc/gather_corr_$name.o $OTHERS
  echo "built build/wc/libpmn_hip_$name.so"
}
build rows -O3 -DPMN_SETTLE_W=1 &
wait
