#!/bin/bash
# Round 4's evidence pass (one GPU-box call).  Everything judged is copied from gpurun_out/r04/ into profiles/.
#  1 PMC traffic of the pmn_warp_correlate launches for the three BASELINE configurations -> profiles/pmc_traffic.json (hash-stamped:
#    bench.py's roofline.traffic only reports a file measured on THIS tree's kernel sources)
#  2 the default bench line incl. the REFERENCE timed on this box (host cores + PyTorch-ROCm); in-flight sweep
#  3 BASELINE configs[2] / configs[4] lines
#  4 rocprofv3 kernel stats of bench.py --eager;  5 SQ / MFMA counters of every kernel;  6 the whole GPU suite
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/r04
rm -rf $E; mkdir -p $E
bash scripts/gpu_pmc_traffic.sh > $E/pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json; cp gpurun_out/pmc_traffic.json $E/pmc_traffic.json
timeout 900 python bench.py > $E/bench.log 2>&1; grep '^{' $E/bench.log > $E/r04_bench.json
for s in 2 4; do timeout 300 python bench.py --no-cpu-baseline --in-flight $s --roofline-steps 4 2>/dev/null | grep '^{' > $E/r04_bench_inflight$s.json; done
timeout 300 python bench.py --no-cpu-baseline --width 1920 --height 1056 --views 7 --samples 8 --steps 40 2>/dev/null | grep '^{' > $E/r04_bench_cfg3.json
timeout 300 python bench.py --no-cpu-baseline --width 3072 --height 2048 --views 10 --samples 3 --steps 40 2>/dev/null | grep '^{' > $E/r04_bench_cfg5.json
bash scripts/gpu_profile.sh 20 > $E/profile_eager.log 2>&1; cp gpurun_out/prof_summary/bench_kernel_stats.csv $E/r04_bench_kernel_stats.csv
rm -rf gpurun_out/prof gpurun_out/prof_summary
for cfg in "cfg3 1920 1056 7" "cfg5 3072 2048 10"; do
  set -- $cfg
  rm -rf $E/prof; mkdir -p $E/prof
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof -o bench -- \
      python $R/bench.py --width $2 --height $3 --views $4 --samples 2 --steps 10 --warmup 2 --no-cpu-baseline --eager --settle-seconds 0.2 --steady-seconds 0 --roofline-steps 8 > $E/prof_$1.log 2>&1)
  for f in $(find $E/prof -name "*kernel_stats.csv"); do cp $f $E/r04_bench_$1_kernel_stats.csv; done
  rm -rf $E/prof
done
bash scripts/gpu_pmc_bench.sh > $E/pmc_bench.log 2>&1; cp gpurun_out/pmc_bench/summary.txt $E/r04_pmc_all_kernels.txt; rm -rf gpurun_out/pmc_bench
timeout 2400 python -m pytest tests/ -q -m gpu --durations=6 2>&1 | tail -16 > $E/r04_pytest_gpu.log
PMN_EXPERIMENTAL=1 timeout 900 python -m pytest tests/test_corr_mfma.py tests/test_gather_win.py tests/test_hip_parity.py -q -m gpu -k "corr or gather or windowed or winograd or mfma or research" 2>&1 | tail -3 >> $E/r04_pytest_gpu.log
python - <<'PY'
import json,os
E=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04/'
for n in ('r04_bench','r04_bench_inflight2','r04_bench_inflight4','r04_bench_cfg3','r04_bench_cfg5'):
    try:
        j=json.load(open(E+n+'.json')); r=j['roofline']
        print(n,'value',j['value'],'steady',(j.get('steady_state') or {}).get('value'),'eager',j['single_stream_eager']['value'],'frac',r['frac'],'kernel_ms',r['kernel_ms_per_step'],'traffic',r['traffic'],'alg',r['alg_bytes_per_step'])
        if 'cpu_baseline' in j: print('  cpu_baseline',j['cpu_baseline']['value'],j['cpu_baseline']['kind'],j['cpu_baseline']['cores'],'port',j['cpu_baseline'].get('port',{}).get('value'))
        if 'reference_rocm' in j: print('  reference_rocm',{k:v for k,v in j['reference_rocm'].items() if k not in ('kind',)})
    except Exception as e: print(n,'unreadable',e)
PY
cat $E/r04_pytest_gpu.log | tail -12; du -sh $R/gpurun_out
