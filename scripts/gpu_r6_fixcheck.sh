#!/bin/bash
# lesson 46: the four victim launches (and, for the product build, every call of the forward) beside the strongest disturbers
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r06_fixcheck.log
: > $L
for v in "$@"; do
  echo "== build $v" | tee -a $L
  lib=build/wc/libpmn_hip_$v.so; [ "$v" = product ] && lib=patchmatchnet_amd/csrc/libpmn_hip.so
  timeout 600 python scripts/overlap_pairs.py --lib $lib --victims ${VICTIMS:-20,22,28,36} --disturbers 9:,43:,micro:\ fp16 --reps 24 2>&1 | grep -av "amdgpu.ids" | grep -a "^victim" | cut -c1-250 | tee -a $L
done
