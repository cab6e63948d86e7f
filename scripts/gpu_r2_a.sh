#!/bin/bash
# round-2 GPU pass A: windowed-vs-streaming bit-exactness, full parity suite, replay tuner, short bench
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch;print(torch.__version__, torch.cuda.get_device_name(0))" > gpurun_out/env.log 2>&1
timeout 600 python -m pytest tests/test_gather_win.py -x -q -m gpu > gpurun_out/pytest_win.log 2>&1; echo "pytest_win exit $?" >> gpurun_out/pytest_win.log
timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gather_win.py > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 600 python scripts/warp_tune.py --reps 10 > gpurun_out/warp_tune.log 2>&1; echo "tune exit $?" >> gpurun_out/warp_tune.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
tail -15 gpurun_out/pytest_win.log; tail -5 gpurun_out/pytest_gpu.log; tail -50 gpurun_out/warp_tune.log; tail -2 gpurun_out/bench.log | cut -c1-1500
