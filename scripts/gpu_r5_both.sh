#!/bin/bash
# gpurun payload (round 5): the packed fusion path -- its tests, then eval.py --output_type both on 24 generated scans in fresh
# processes (async fusion workers; one worker; inline)
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== tests" | tee gpurun_out/r05_both_tests.log
timeout 1500 python -m pytest tests/test_fusion_gpu.py tests/test_eval_gpu.py -q -m gpu -k "fus or eval or packed or ranks" --durations=5 2>&1 | grep -v Warning | tail -60 | tee -a gpurun_out/r05_both_tests.log
OUTPUT_TYPE=both RUNS="1 2" bash scripts/eval_procs.sh
mv gpurun_out/eval_procs_both.log gpurun_out/eval_procs_both_async.log
OUTPUT_TYPE=both RUNS="1" EVAL_EXTRA="--fuse_workers 1" bash scripts/eval_procs.sh
mv gpurun_out/eval_procs_both.log gpurun_out/eval_procs_both_async1.log
OUTPUT_TYPE=both RUNS="1" EVAL_EXTRA="--fuse_workers 3 --fuse_threads 24" bash scripts/eval_procs.sh
mv gpurun_out/eval_procs_both.log gpurun_out/eval_procs_both_async3.log
