#!/bin/bash
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ktrace_fn
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && FN_ONLY_HIP=1 rocprofv3 --kernel-trace --output-format csv -d $OUT -o kt -- python $GRAFT_REPO_ROOT/scripts/featurenet_bench.py > $OUT/log.txt 2>&1)
python - <<'PY'
import csv,glob,collections,os,re
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/ktrace_fn'
rows=list(csv.DictReader(open(glob.glob(out+'/*kernel_trace.csv')[0])))
rows=[r for r in rows if 'conv_kernel' in r['Kernel_Name'] or 'conv_tiled' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last forward = last 16 conv kernels
last=rows[-16:]
tot=0
for r in last:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3; tot+=d
    m=re.search(r'(conv_\w+<[^>]*)>', r['Kernel_Name'])
    print('%-40s grid %8s x%s  vgpr %4s sgpr %4s  %9.1f us'%(m.group(1), r['Grid_Size_X'], r['Grid_Size_Y'], r['VGPR_Count'], r['SGPR_Count'], d))
print('total conv us', tot)
print(open(out+'/log.txt').read()[-300:])
PY
