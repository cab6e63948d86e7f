#!/bin/bash
# same-box A/B of library variants with scripts/kernel_bench.py (all hot-path launch shapes)
export TMPDIR=/tmp
LIB=patchmatchnet_amd/csrc/libpmn_hip.so
cp $LIB /tmp/libpmn_orig.so
: > gpurun_out/ab_kb.log
for round in 1 2; do
  for v in ${VARIANTS:-A B}; do
    cp scripts/microbench/variants/libpmn_$v.so $LIB
    echo "== variant $v round $round" >> gpurun_out/ab_kb.log
    timeout 300 python scripts/kernel_bench.py --reps 20 2>&1 | grep -E "feature_weight|warp|aggregate|hypoth" | cut -c1-120 >> gpurun_out/ab_kb.log
  done
done
cp /tmp/libpmn_orig.so $LIB
cat gpurun_out/ab_kb.log
