#!/bin/bash
# round-2 GPU pass C: lane = item engine: bit-exactness vs streaming, replay timings / ablations
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gather_win.py -x -q -m gpu > gpurun_out/pytest_win.log 2>&1; echo "pytest_win exit $?" >> gpurun_out/pytest_win.log
timeout 900 python scripts/warp_tune.py --reps 8 --configs stream lane12 lane12w2 lane8 lane16w2 lane20w2 lane12w2@1 lane12w2@2 lane12w2@3 lane12w2@8 lane12w2@11 > gpurun_out/warp_lane.log 2>&1; echo "tune exit $?" >> gpurun_out/warp_lane.log
tail -25 gpurun_out/pytest_win.log; grep -E "launch [0-9]|total|Error|error" gpurun_out/warp_lane.log | cut -c1-150
