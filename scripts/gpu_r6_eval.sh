#!/bin/bash
# gpurun payload (round 6): the two new scene parity tests, then eval.py end to end in fresh processes (depth, then both)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fullsize_parity.py -q -m gpu -k "cfg3_scene_end or cfg5_scene_end" 2>&1 | grep -av "Warning\|warnings.warn\|return \|^$\|pin_memory" | tail -15 | tee gpurun_out/r06_scene_tests.log
OUTPUT_TYPE=depth RUNS="1 2 3" bash scripts/eval_procs.sh; mv gpurun_out/eval_procs_depth.log gpurun_out/eval_procs_depth_f2.log
OUTPUT_TYPE=depth RUNS="1 2" EVAL_EXTRA="--in_flight 3" bash scripts/eval_procs.sh; mv gpurun_out/eval_procs_depth.log gpurun_out/eval_procs_depth_f3.log
OUTPUT_TYPE=both RUNS="1 2" bash scripts/eval_procs.sh
