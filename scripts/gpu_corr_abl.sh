#!/bin/bash
# gpurun payload: timing of ablation builds of corr_mfma.hip (wrong results on purpose; development aid)
mkdir -p gpurun_out
export TMPDIR=/tmp
export PMN_EXPERIMENTAL=1  # the matrix-core formulation lives in the research build
LOG=gpurun_out/corr_abl.log
: > $LOG
LIB=patchmatchnet_amd/csrc/libpmn_hip_experimental.so
cp $LIB /tmp/libpmn_orig.so
for v in ${VARIANTS:-A}; do
  cp scripts/microbench/variants/libpmn_$v.so $LIB
  echo "== variant $v" | tee -a $LOG
  timeout 300 python scripts/corr_ab.py --impls mfma --reps ${REPS:-10} 2>&1 | grep -E '^\{' | python scripts/corr_ab_fmt.py | tee -a $LOG
done
cp /tmp/libpmn_orig.so $LIB
