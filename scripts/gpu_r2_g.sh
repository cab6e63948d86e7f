#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gather_win.py tests/test_hip_parity.py -x -q -m gpu -k "not large_configs" > gpurun_out/pytest_quick.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_quick.log
timeout 600 python scripts/warp_tune.py --reps 10 --configs stream > gpurun_out/warp_stream.log 2>&1
tail -4 gpurun_out/pytest_quick.log; grep -E "launch [0-9]|total" gpurun_out/warp_stream.log | cut -c1-150
