#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command; summaries copied to gpurun_out/prof_summary/
mkdir -p gpurun_out/prof gpurun_out/prof_summary
export TMPDIR=/tmp
STEPS=${1:-10}
MODE=--eager   # kernels one at a time: their own durations; "default" as 2nd argument = the default command (two samples in flight)
[ "$2" = "default" ] && MODE=
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof $GRAFT_REPO_ROOT/gpurun_out/prof_summary; mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof $GRAFT_REPO_ROOT/gpurun_out/prof_summary
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup 3 --no-cpu-baseline $MODE > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*stats*" | head
for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do cp $f gpurun_out/prof_summary/; done
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/prof_summary/*kernel_stats.csv'):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    print(f, 'total ms', tot/1e6)
    for r in rows[:28]:
        print('%-90s calls %6s avg_us %10.1f total_ms %9.2f  %5.1f%%'%(r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, 100*float(r['TotalDurationNs'])/tot))
PY
tail -2 gpurun_out/prof_bench.log
