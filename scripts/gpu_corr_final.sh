#!/bin/bash
# gpurun payload: the research build's matrix-core pmn_warp_correlate -- parity tests, then the same-box A/B against streaming
mkdir -p gpurun_out
export TMPDIR=/tmp
export PMN_EXPERIMENTAL=1
LOG=gpurun_out/corr_final.log
echo "== pytest tests/test_corr_mfma.py tests/test_gather_win.py (research build)" | tee $LOG
timeout 1200 python -m pytest tests/test_corr_mfma.py tests/test_gather_win.py -q 2>&1 | tail -8 | tee -a $LOG
echo "== A/B on a real forward (scripts/corr_ab.py)" | tee -a $LOG
timeout 600 python scripts/corr_ab.py --reps 30 --json gpurun_out/corr_ab.json 2>&1 | grep -E '^\{' | python scripts/corr_ab_fmt.py | tee -a $LOG
