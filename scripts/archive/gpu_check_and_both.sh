bash scripts/gpu_check.sh
timeout 400 python scripts/eval_bench.py 3 49 --both > gpurun_out/eval_bench_both.log 2>&1; echo "exit $?" >> gpurun_out/eval_bench_both.log
grep -E "RESULT|fusion stage|depth stage|exit" gpurun_out/eval_bench_both.log | tail -20
