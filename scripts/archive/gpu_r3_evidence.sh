#!/bin/bash
# Round 3's evidence pass (one GPU-box call): PMC traffic of the gather launches -> profiles/pmc_traffic.json (so that the bench line's
# roofline.traffic is measured on THIS tree), the default bench line, rocprofv3 kernel stats of the eager and the default bench,
# SQ counters of every kernel, eval.py end to end.  Everything judged is copied from gpurun_out/r03/ into profiles/.
export TMPDIR=/tmp
E=$GRAFT_REPO_ROOT/gpurun_out/r03
rm -rf $E; mkdir -p $E
bash scripts/gpu_pmc_traffic.sh > $E/pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json; cp gpurun_out/pmc_traffic.json $E/pmc_traffic.json
timeout 600 python bench.py > $E/bench.log 2>&1; grep '^{' $E/bench.log > $E/r03_bench.json
bash scripts/gpu_profile.sh 20 > $E/profile_eager.log 2>&1; cp gpurun_out/prof_summary/bench_kernel_stats.csv $E/r03_bench_kernel_stats.csv
bash scripts/gpu_profile.sh 20 default > $E/profile_default.log 2>&1; cp gpurun_out/prof_summary/bench_kernel_stats.csv $E/r03_bench_default_kernel_stats.csv
rm -rf gpurun_out/prof gpurun_out/prof_summary
bash scripts/gpu_pmc_bench.sh > $E/pmc_bench.log 2>&1; cp gpurun_out/pmc_bench/summary.txt $E/r03_pmc_all_kernels.txt; rm -rf gpurun_out/pmc_bench
timeout 600 python scripts/eval_bench.py 6 49 --all > $E/r03_eval_bench.log 2>&1
grep -E "RESULT|BEST" $E/r03_eval_bench.log; cut -c1-400 $E/r03_bench.json; grep -o '"roofline".*"per_shape"' $E/r03_bench.json | cut -c1-900; head -30 $E/profile_eager.log | cut -c1-200
du -sh gpurun_out
