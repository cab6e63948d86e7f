#!/bin/bash
# exact per-kernel durations of the micro-benchmark from rocprofv3's kernel trace
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ktrace
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o kt -- python $GRAFT_REPO_ROOT/scripts/kernel_bench.py ${KB_ARGS:---only all --reps 5} > $OUT/log.txt 2>&1)
python - <<'PY'
import csv,glob,collections,os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/ktrace'
rows=list(csv.DictReader(open(glob.glob(out+'/*kernel_trace.csv')[0])))
agg=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if any(k in n for k in ('gather_corr','aggregate','init_hyp','nchw','confidence')):
        agg[(n[:60], r['Grid_Size_X'], r['LDS_Block_Size'], r['VGPR_Count'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(agg.items()):
    v=sorted(v); print('%-62s grid %8s lds %6s vgpr %4s  n=%3d  median %8.1f us  min %8.1f us'%(k[0],k[1],k[2],k[3],len(v),v[len(v)//2],v[0]))
PY
