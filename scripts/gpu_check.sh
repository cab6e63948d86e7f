#!/bin/bash
# One GPU-box pass: parity tests (no -x: every failure is wanted in one call), smoke, default bench.  Outputs land in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch;print(torch.__version__, torch.cuda.get_device_name(0))" > gpurun_out/env.log 2>&1
SECONDS=0
timeout ${PYTEST_TIMEOUT:-1200} python -m pytest tests -q -m gpu --durations=40 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
tail -60 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; tail -2 gpurun_out/bench.log | cut -c1-1500
