#!/bin/bash
# One GPU-box pass: parity tests (no -x: every failure is wanted in one call), smoke, default bench.  Outputs land in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch;print(torch.__version__, torch.cuda.get_device_name(0))" > gpurun_out/env.log 2>&1
timeout 1500 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
timeout 300 python bench.py --scene rolled --no-cpu-baseline > gpurun_out/bench_rolled.log 2>&1
tail -40 gpurun_out/pytest_gpu.log; tail -5 gpurun_out/smoke.log; tail -3 gpurun_out/bench.log; tail -2 gpurun_out/bench_rolled.log
