#!/bin/bash
# A/B of conv_f16s.hip build variants on ONE box (scripts/microbench/variants/libpmn_f16s_v<N>.so, swapped in turn, two rounds).
mkdir -p gpurun_out
export TMPDIR=/tmp
LIB=patchmatchnet_amd/csrc/libpmn_hip.so
cp $LIB /tmp/libpmn_orig.so
: > gpurun_out/f16s_ab.log
for round in 1 2; do
  for v in ${VARIANTS:-0 1 2}; do
    cp scripts/microbench/variants/libpmn_f16s_v$v.so $LIB
    PMN_F16S_VARIANT=$v timeout 200 python scripts/f16s_layers.py 2>&1 | grep -E "variant|Error|error" >> gpurun_out/f16s_ab.log
  done
done
cp /tmp/libpmn_orig.so $LIB
cat gpurun_out/f16s_ab.log
