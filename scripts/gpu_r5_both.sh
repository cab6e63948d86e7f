#!/bin/bash
# gpurun payload (round 5): the packed fusion path -- its tests, then eval.py --output_type both on 24 generated scans in fresh
# processes (async fusion worker, then --fuse_async 0 for the A/B, then more PNG threads)
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== tests" | tee gpurun_out/r05_both_tests.log
timeout 1500 python -m pytest tests/test_fusion_gpu.py tests/test_eval_gpu.py tests/test_fullsize_parity.py -q -m gpu -k "fus or eval or rocm or packed or ranks" --durations=5 2>&1 | tail -15 | tee -a gpurun_out/r05_both_tests.log
OUTPUT_TYPE=both RUNS="1 2" bash scripts/eval_procs.sh
mv gpurun_out/eval_procs_both.log gpurun_out/eval_procs_both_async.log
OUTPUT_TYPE=both RUNS="1" EVAL_EXTRA="--fuse_threads 16" bash scripts/eval_procs.sh
mv gpurun_out/eval_procs_both.log gpurun_out/eval_procs_both_async16.log
OUTPUT_TYPE=both RUNS="1" EVAL_EXTRA="--fuse_async 0" bash scripts/eval_procs.sh
mv gpurun_out/eval_procs_both.log gpurun_out/eval_procs_both_inline.log
