#!/bin/bash
# What part of lesson 46's fix is the fix?  pmn_settle is `v_mov_b32 v, v` (inline asm).  Probe builds of gather_corr.hip from a scratch
# copy of csrc/ with the asm replaced (build/wc/libpmn_hip_settle_<name>.so), for scripts/repro/library_overlap_repro.cpp:
#   empty   asm volatile("" : "+v"(v))          the value is pinned to a register at that point (hipcc places its s_waitcnt there), no instruction
#   nop     asm volatile("s_nop 4" : "+v"(v))   the same plus five idle cycles
#   copy    asm volatile("v_mov_b32 %0, %1" : "=v"(o) : "v"(v))   a VALU copy into a register of hipcc's choice
# Results: profiles/r06_overlap/r06_settle_probes.log
set -e
cd "$(dirname "$0")/.."
CS=patchmatchnet_amd/csrc
make -s -C $CS -j8
mkdir -p build/wc
FLAGS="-std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -O3"
OTHERS=$(ls $CS/*.o | grep -v '\.x\.o' | grep -v gather_corr.o)
probe() { # name, replacement for the asm statement
  local name=$1 repl=$2 d=build/wc/csrc_$1
  rm -rf $d; mkdir -p $d
  cp $CS/*.hpp $CS/gather_corr.hip $d/
  sed -i "s|\"../../include/pmn_hip.h\"|\"$PWD/include/pmn_hip.h\"|" $d/*.hpp $d/*.hip
  python3 - "$d/gather_common.hpp" "$repl" <<'PY'
import sys
p, repl = sys.argv[1], sys.argv[2]
s = open(p).read()
old = 'asm volatile("v_mov_b32 %0, %0" : "+v"(v));'
assert s.count(old) == 1
open(p, "w").write(s.replace(old, repl))
PY
  /opt/rocm/bin/hipcc $FLAGS -I$d -c $d/gather_corr.hip -o build/wc/gather_corr_settle_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/wc/libpmn_hip_settle_$name.so build/wc/gather_corr_settle_$name.o $OTHERS
  echo "built build/wc/libpmn_hip_settle_$name.so"
}
probe empty 'asm volatile("" : "+v"(v));'
probe nop 'asm volatile("s_nop 4" : "+v"(v));'
probe copy '{ float o; asm volatile("v_mov_b32 %0, %1" : "=v"(o) : "v"(v)); v = o; }'
