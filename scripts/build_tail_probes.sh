#!/bin/bash
# Lesson 46, narrowing: the -DPMN_NO_SETTLE build (everything unguarded) plus an EMPTY asm pin (`asm volatile("" : "+v"(v))`, no
# instruction) on a subset of the PixelwiseNet MLP's tail constants -- which of t1[0..3] (ta), t1[4..7] (tb), w2[0..3] (wa), w2[4..7] (wb),
# b2 must hipcc be kept from treating as part of its ds_read_b128 vector for the launch to come out right beside MFMA kernels?
# build/wc/libpmn_hip_tail_<set>.so for scripts/repro/library_overlap_repro.cpp.  Results: profiles/r06_overlap/r06_settle_probes.log
set -e
cd "$(dirname "$0")/.."
CS=patchmatchnet_amd/csrc
make -s -C $CS -j8
mkdir -p build/wc
FLAGS="-std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -O3 -DPMN_NO_SETTLE"
OTHERS=$(ls $CS/*.o | grep -v '\.x\.o' | grep -v gather_corr.o)
probe() { # name = the pinned subset, e.g. ta_wb
  local name=$1 d=build/wc/csrc_tail_$1
  rm -rf $d; mkdir -p $d
  cp $CS/*.hpp $CS/gather_corr.hip $d/
  sed -i "s|\"../../include/pmn_hip.h\"|\"$PWD/include/pmn_hip.h\"|" $d/*.hpp $d/*.hip
  python3 - "$d/gather_common.hpp" "$name" <<'PY'
import re, sys
p, name = sys.argv[1], sys.argv[2]
on = set(name.split("_"))
s = open(p).read()
s = s.replace("__device__ __forceinline__ float4 pmn_settle4(float4 v) {",
              '__device__ __forceinline__ float pmn_pin(float v) {\n    asm volatile("" : "+v"(v));\n    return v;\n}\n'
              "__device__ __forceinline__ float4 pmn_settle4(float4 v) {", 1)
i = s.index("    const float t1[8] = {pmn_settle(ta.x)")
j = s.index("    const float b2 = pmn_settle(W[336]);") + len("    const float b2 = pmn_settle(W[336]);")
f = lambda k: "pmn_pin" if k in on else ""
new = ("    const float t1[8] = {%s(ta.x), %s(ta.y), %s(ta.z), %s(ta.w), %s(tb.x), %s(tb.y), %s(tb.z), %s(tb.w)};\n" % ((f("ta"),) * 4 + (f("tb"),) * 4) +
       "    const float w2[8] = {%s(wa.x), %s(wa.y), %s(wa.z), %s(wa.w), %s(wb.x), %s(wb.y), %s(wb.z), %s(wb.w)};\n" % ((f("wa"),) * 4 + (f("wb"),) * 4) +
       "    const float b2 = %s(W[336]);" % f("b2"))
open(p, "w").write(s[:i] + new + s[j:])
PY
  /opt/rocm/bin/hipcc $FLAGS -I$d -c $d/gather_corr.hip -o build/wc/gather_corr_tail_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/wc/libpmn_hip_tail_$name.so build/wc/gather_corr_tail_$name.o $OTHERS
  echo "built build/wc/libpmn_hip_tail_$name.so"
}
for v in "$@"; do probe $v; done
