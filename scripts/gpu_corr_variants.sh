#!/bin/bash
# gpurun payload: parity tests on the product library, then corr_ab.py (mfma only) on each variant library, interleaved twice
mkdir -p gpurun_out
export TMPDIR=/tmp
export PMN_EXPERIMENTAL=1  # the matrix-core formulation lives in the research build
LOG=gpurun_out/corr_variants.log
: > $LOG
echo "== pytest tests/test_corr_mfma.py (product library)" | tee -a $LOG
timeout 900 python -m pytest tests/test_corr_mfma.py -q -x 2>&1 | tail -15 | tee -a $LOG
LIB=patchmatchnet_amd/csrc/libpmn_hip_experimental.so
cp $LIB /tmp/libpmn_orig.so
for round in 1 2; do
  echo "== stream round $round" | tee -a $LOG
  timeout 300 python scripts/corr_ab.py --impls stream --reps ${REPS:-20} 2>&1 | grep -E '^\{' | python scripts/corr_ab_fmt.py | tee -a $LOG
  for v in ${VARIANTS:-A}; do
    cp scripts/microbench/variants/libpmn_$v.so $LIB
    echo "== variant $v round $round" | tee -a $LOG
    timeout 300 python scripts/corr_ab.py --impls mfma --reps ${REPS:-20} 2>&1 | grep -E '^\{' | python scripts/corr_ab_fmt.py | tee -a $LOG
  done
  cp /tmp/libpmn_orig.so $LIB
done
