#!/bin/bash
# round-2 GPU pass E: whole GPU suite (incl. full-size parity, fusion, 2-rank eval) + bench
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
timeout 1700 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 40 --warmup 5 > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
tail -40 gpurun_out/pytest_gpu.log; cat gpurun_out/parity_report.jsonl; tail -2 gpurun_out/bench.log | cut -c1-1800
