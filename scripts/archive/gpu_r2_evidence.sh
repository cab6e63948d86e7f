#!/bin/bash
# Round-2 evidence pass on one box: PMC traffic of the gather kernels -> profiles/pmc_traffic.json (hash-stamped), bench line,
# rocprofv3 kernel stats of the bench command, SQ counters of every kernel of the forward, eval.py-style rate, TA microbenchmark.
mkdir -p gpurun_out/evidence
export TMPDIR=/tmp
E=gpurun_out/evidence
rm -rf gpurun_out/pmc_win
PMC_ARGS="--configs stream --reps 2" bash scripts/gpu_pmc_win.sh > $E/pmc_win.log 2>&1
python scripts/make_traffic_json.py > $E/pmc_traffic_print.txt 2>&1
cp profiles/pmc_traffic.json $E/pmc_traffic.json
cp gpurun_out/pmc_win/summary.txt $E/pmc_gather_summary.txt
python bench.py > $E/bench.json 2> $E/bench.err
tail -1 $E/bench.json
bash scripts/gpu_profile.sh 20 > $E/profile.txt 2>&1
cp gpurun_out/prof_summary/*kernel_stats.csv $E/bench_kernel_stats.csv
bash scripts/gpu_profile.sh 20 default > $E/profile_default.txt 2>&1
cp gpurun_out/prof_summary/*kernel_stats.csv $E/bench_default_kernel_stats.csv
bash scripts/gpu_pmc_bench.sh > $E/pmc_bench.txt 2>&1
cp gpurun_out/pmc_bench/summary.txt $E/pmc_all_kernels.txt
timeout 400 python scripts/pipeline_bench.py --samples 128 --writer_threads 3 --outdirs /dev/shm > $E/pipeline_bench.log 2>&1
grep PIPELINE $E/pipeline_bench.log | cut -c1-300
timeout 60 scripts/microbench/ta_mask > $E/ta_mask.txt 2>&1
rm -rf gpurun_out/pmc_win/p*/*/*.db gpurun_out/pmc_bench/p*/*/*.db 2>/dev/null
du -sh gpurun_out | tail -1
