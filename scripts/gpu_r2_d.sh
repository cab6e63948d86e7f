#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
PMC_ARGS="--configs lane12w2 --reps 2" bash scripts/gpu_pmc_win.sh > gpurun_out/pmc_lane.log 2>&1
cp gpurun_out/pmc_win/summary.txt gpurun_out/pmc_lane_summary.txt
grep -A32 "gather_lane_kernel" gpurun_out/pmc_lane_summary.txt | cut -c1-100
