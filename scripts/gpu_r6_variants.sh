#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r06_variants.log
for v in "$@"; do
  echo "== variant $v" | tee -a $L
  PMN_PLANES=8 timeout 600 python scripts/overlap_pairs.py --lib build/wc/libpmn_hip_$v.so --victims 28,36 --disturbers 9:,micro:\ fp16 2>&1 | grep -av "amdgpu.ids" | grep -a "^victim" | cut -c1-330 | tee -a $L
done
