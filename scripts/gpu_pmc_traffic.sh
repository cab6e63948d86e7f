#!/bin/bash
# HBM traffic of the pmn_warp_correlate launches for bench.py's `roofline.traffic`: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3
# --pmc passes (counters only with --kernel-trace; MI355X_MICROARCH.md, HBM section) over the SAME command the roofline figure is
# measured with (bench.py --eager: a real forward on the bench's samples), then scripts/make_traffic_json.py (gfx950 correction:
# FETCH_SIZE x 2).  One entry per bench configuration (default: BASELINE configs[1], [2], [4]).
# Result: gpurun_out/pmc_traffic.json (copy to profiles/pmc_traffic.json; it carries the kernel sources' hash).
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic
rm -rf $OUT $GRAFT_REPO_ROOT/gpurun_out/pmc_traffic.json; mkdir -p $OUT
for cfg in ${CONFIGS:-"1600 1200 5" "1920 1056 7" "3072 2048 10"}; do
  set -- $cfg
  i=0
  for c in FETCH_SIZE WRITE_SIZE; do
    i=$((i+1))
    (cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --width $1 --height $2 --views $3 --samples 2 --steps 4 --warmup 2 --no-cpu-baseline --eager --roofline-steps 4 --steady-seconds 0 --settle-seconds 0 > $OUT/p$i.log 2>&1)
  done
  python $GRAFT_REPO_ROOT/scripts/make_traffic_json.py $OUT $GRAFT_REPO_ROOT/gpurun_out/pmc_traffic.json ${1}x${2}_N${3} | head -3
  rm -rf $OUT/p1 $OUT/p2
done
