#!/bin/bash
# Lean variant of gpu_profile.sh for the end of a round's GPU budget: rocprofv3 kernel trace + stats of `bench.py --eager` with the
# settle / steady loops cut short; only the kernel_stats summary is kept (gpurun_out/prof_summary/), the trace is deleted on the box.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof $R/gpurun_out/prof_summary; mkdir -p $R/gpurun_out/prof $R/gpurun_out/prof_summary
cd /tmp && timeout ${PROF_TIMEOUT:-70} rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- \
    python $R/bench.py --steps ${1:-40} --warmup 3 --no-cpu-baseline --eager --settle-seconds 0.2 --steady-seconds 0 > $R/gpurun_out/prof_bench.log 2>&1
cd $R
for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do cp $f gpurun_out/prof_summary/; done
rm -rf gpurun_out/prof
ls -la gpurun_out/prof_summary; tail -1 gpurun_out/prof_bench.log | cut -c1-300
