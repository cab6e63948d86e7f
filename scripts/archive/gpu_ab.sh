#!/bin/bash
# A/B of kernel variants on ONE box: scripts/microbench/variants/libpmn_<X>.so are swapped in turn (two rounds, interleaved)
mkdir -p gpurun_out
export TMPDIR=/tmp
LIB=patchmatchnet_amd/csrc/libpmn_hip.so
cp $LIB /tmp/libpmn_orig.so
: > gpurun_out/ab.log
for round in 1 2; do
  for v in ${VARIANTS:-A B C}; do
    cp scripts/microbench/variants/libpmn_$v.so $LIB
    echo "== variant $v round $round" >> gpurun_out/ab.log
    timeout 300 python scripts/warp_tune.py --reps ${REPS:-10} --configs stream 2>&1 | grep -E "launch [0-9]|total" | cut -c1-150 >> gpurun_out/ab.log
  done
done
cp /tmp/libpmn_orig.so $LIB
grep -E "==|total" gpurun_out/ab.log
