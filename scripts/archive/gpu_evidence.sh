#!/bin/bash
# Round-end evidence pass on the GPU box: full parity suite, smoke, bench (with cpu_baseline), rocprofv3 kernel stats of the
# bench, kernel micro-benchmark.  Everything lands in gpurun_out/ (copy what is judged into profiles/).
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 900 python bench.py --steps 40 --warmup 5 > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
tail -1 gpurun_out/bench.log > /dev/null
grep '^{' gpurun_out/bench.log | tail -1 > gpurun_out/bench_final.json
bash scripts/gpu_profile.sh 20 > gpurun_out/profile_final.txt 2>&1
timeout 600 python scripts/kernel_bench.py --reps 10 > gpurun_out/kernel_bench.log 2>&1
tail -4 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; cat gpurun_out/bench_final.json | cut -c1-400; tail -14 gpurun_out/kernel_bench.log
