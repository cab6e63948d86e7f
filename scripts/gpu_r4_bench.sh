#!/bin/bash
# Round 4 bench pass (one GPU-box call): the default bench line incl. the reference timed on this box (host cores + PyTorch-ROCm),
# then BASELINE configs[2] / configs[4] (one GPU's share) with rocprofv3 kernel stats of their eager runs.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/r04
rm -rf $E; mkdir -p $E
timeout 900 python bench.py > $E/bench.log 2>&1; grep '^{' $E/bench.log > $E/r04_bench.json
python - <<'PY'
import json,os
p=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04/r04_bench.json'
try:
    j=json.load(open(p))
    print('value',j['value'],'frac',j['roofline']['frac'])
    print('cpu_baseline',json.dumps(j.get('cpu_baseline'))[:700])
    print('reference_rocm',json.dumps(j.get('reference_rocm'))[:900])
except Exception as e:
    print('bench line unreadable',e); print(open(p.replace('r04_bench.json','bench.log')).read()[-2000:])
PY
for cfg in "cfg3 1920 1056 7 8" "cfg5 3072 2048 10 3"; do
  set -- $cfg
  timeout 900 python bench.py --width $2 --height $3 --views $4 --samples $5 --no-cpu-baseline --steps 40 > $E/bench_$1.log 2>&1
  grep '^{' $E/bench_$1.log > $E/r04_bench_$1.json; cut -c1-260 $E/r04_bench_$1.json; tail -2 $E/bench_$1.log | cut -c1-300
  rm -rf $E/prof; mkdir -p $E/prof
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof -o bench -- \
      python $R/bench.py --width $2 --height $3 --views $4 --samples 2 --steps 10 --warmup 2 --no-cpu-baseline --eager --settle-seconds 0.2 --steady-seconds 0 --roofline-steps 8 > $E/prof_$1.log 2>&1)
  for f in $(find $E/prof -name "*kernel_stats.csv"); do cp $f $E/r04_bench_$1_kernel_stats.csv; done
  rm -rf $E/prof
done
ls -la $E; du -sh $R/gpurun_out
