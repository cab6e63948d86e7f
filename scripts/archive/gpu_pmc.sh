#!/bin/bash
# PMC passes over the kernel micro-benchmark (counters only with --kernel-trace, one pass per counter group).
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
ARGS=${PMC_ARGS:---only warp --reps 2}
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python $GRAFT_REPO_ROOT/scripts/kernel_bench.py $ARGS > $OUT/p$i.log 2>&1)
done <<LIST
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
FETCH_SIZE
WRITE_SIZE
LIST
python - <<'PY'
import csv,glob,collections,os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc'
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out+'/p*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'gather_corr' not in k and 'aggregate' not in k: continue
        key=(k[:40], r.get('Grid_Size','') or r.get('Grid_Size_X',''), r.get('LDS_Block_Size',''))
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out+'/summary.txt','w') as fo:
    for key,cs in agg.items():
        line='%s grid=%s lds=%s\n'%key + ''.join('   %-32s %16.0f (n=%d)\n'%(c, sum(v)/len(v), len(v)) for c,v in sorted(cs.items()))
        print(line); fo.write(line)
PY
