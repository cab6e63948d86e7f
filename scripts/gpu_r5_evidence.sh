#!/bin/bash
# Round 5's evidence pass (one GPU-box call).  Everything judged is copied from gpurun_out/r05/ into profiles/.
#  1 PMC traffic of the pmn_warp_correlate launches for the three BASELINE configurations -> profiles/pmc_traffic.json (hash-stamped)
#  2 the default bench line incl. the REFERENCE timed on this box (host cores + PyTorch-ROCm, features-forced parity figure)
#  3 BASELINE configs[2] / configs[4] lines
#  4 rocprofv3 kernel stats of bench.py --eager (three configurations);  5 SQ / MFMA counters of every kernel;  6 the whole GPU suite
#  7 eval.py end to end in fresh processes: --output_type depth and both
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/r05
rm -rf $E; mkdir -p $E
bash scripts/gpu_pmc_traffic.sh > $E/pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json; cp gpurun_out/pmc_traffic.json $E/pmc_traffic.json
timeout 900 python bench.py > $E/bench.log 2>&1; grep '^{' $E/bench.log > $E/r05_bench.json
timeout 300 python bench.py --no-cpu-baseline --width 1920 --height 1056 --views 7 --samples 8 --steps 40 2>/dev/null | grep '^{' > $E/r05_bench_cfg3.json
timeout 300 python bench.py --no-cpu-baseline --width 3072 --height 2048 --views 10 --samples 3 --steps 40 2>/dev/null | grep '^{' > $E/r05_bench_cfg5.json
bash scripts/gpu_profile.sh 20 > $E/profile_eager.log 2>&1; cp gpurun_out/prof_summary/bench_kernel_stats.csv $E/r05_bench_kernel_stats.csv
rm -rf gpurun_out/prof gpurun_out/prof_summary
for cfg in "cfg3 1920 1056 7" "cfg5 3072 2048 10"; do
  set -- $cfg
  rm -rf $E/prof; mkdir -p $E/prof
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof -o bench -- \
      python $R/bench.py --width $2 --height $3 --views $4 --samples 2 --steps 10 --warmup 2 --no-cpu-baseline --eager --settle-seconds 0.2 --steady-seconds 0 --roofline-steps 8 > $E/prof_$1.log 2>&1)
  for f in $(find $E/prof -name "*kernel_stats.csv"); do cp $f $E/r05_bench_$1_kernel_stats.csv; done
  rm -rf $E/prof
done
bash scripts/gpu_pmc_bench.sh > $E/pmc_bench.log 2>&1; cp gpurun_out/pmc_bench/summary.txt $E/r05_pmc_all_kernels.txt; rm -rf gpurun_out/pmc_bench
rm -f gpurun_out/parity_report.jsonl
timeout 2400 python -m pytest tests/ -q -m gpu --durations=6 2>&1 | tail -16 > $E/r05_pytest_gpu.log
PMN_EXPERIMENTAL=1 timeout 900 python -m pytest tests/test_corr_mfma.py tests/test_gather_win.py tests/test_hip_parity.py -q -m gpu -k "corr or gather or windowed or winograd or mfma or research" 2>&1 | tail -3 >> $E/r05_pytest_gpu.log
cp gpurun_out/parity_report.jsonl $E/r05_parity_report.jsonl; cp gpurun_out/rocm_parity.json $E/r05_rocm_parity.json
OUTPUT_TYPE=depth RUNS="1 2 3" bash scripts/eval_procs.sh > /dev/null 2>&1; cp gpurun_out/eval_procs_depth.log $E/r05_eval_procs_depth.log
OUTPUT_TYPE=both RUNS="1 2 3" bash scripts/eval_procs.sh > /dev/null 2>&1; cp gpurun_out/eval_procs_both.log $E/r05_eval_procs_both.log
python - <<'PY'
import json,os
E=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r05/'
for n in ('r05_bench','r05_bench_cfg3','r05_bench_cfg5'):
    try:
        j=json.load(open(E+n+'.json')); r=j['roofline']
        print(n,'value',j['value'],'steady',(j.get('steady_state') or {}).get('value'),'eager',j['single_stream_eager']['value'],'frac',r['frac'],'kernel_ms',r['kernel_ms_per_step'],'traffic',r['traffic'],'alg',r['alg_bytes_per_step'],'other_mode',(j.get('value_other_input_mode') or {}).get('value'))
        print('  per_shape',{k:v['ms_avg'] for k,v in r['per_shape'].items()})
        if 'cpu_baseline' in j: print('  cpu_baseline',j['cpu_baseline']['value'],j['cpu_baseline']['kind'],j['cpu_baseline']['cores'],'port',j['cpu_baseline'].get('port',{}).get('value'))
        if 'reference_rocm' in j: print('  reference_rocm',{k:v for k,v in j['reference_rocm'].items() if k not in ('kind',)})
    except Exception as e: print(n,'unreadable',e)
PY
cat $E/r05_pytest_gpu.log | tail -12; cat $E/r05_eval_procs_depth.log $E/r05_eval_procs_both.log | grep -E "depth stage|both stages"; du -sh $R/gpurun_out
