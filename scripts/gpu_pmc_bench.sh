#!/bin/bash
# PMC passes over a short bench run: per-kernel SQ counters for every kernel of the forward (development aid).
# (Four counters per pass for the first three: on some boxes of this pool rocprofv3 segfaults inside its dispatch callback with the
#  8-counter sets round 2 used -- at the first pmn_aggregate_regress launch, product code untouched -- or hangs until the timeout.)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_bench
rm -rf $OUT; mkdir -p $OUT
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  for attempt in 1 2 3; do  # (a pass that crashes or hangs leaves no csv: up to three attempts -- it is the profiler, not the product)
    rm -rf $OUT/p$i
    (cd /tmp && timeout 150 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --eager --roofline-steps 4 > $OUT/p$i.log 2>&1)
    [ -n "$(find $OUT/p$i -name '*counter_collection.csv' 2>/dev/null)" ] && break
    echo "pass $i ($line): attempt $attempt left no counter file"
  done
done <<LIST
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS
SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD
LIST
python - <<'PY'
import csv,glob,collections,os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_bench'
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out+'/p*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if k.startswith('void at::') or 'rocclr' in k: continue
        agg[k[:64]][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out+'/summary.txt','w') as fo:
    for key,cs in agg.items():
        g=lambda n: (sum(cs[n])/len(cs[n])) if n in cs else 0.0
        if 'GRBM_GUI_ACTIVE' not in cs or 'SQ_WAVES' not in cs:  # the pass with the denominators failed: no ratios rather than wrong ones
            line='%-66s (the GRBM_GUI_ACTIVE / SQ_WAVES pass left no counters: ratios not computable)\n'%key
            print(line,end=''); fo.write(line); continue
        cyc=g('GRBM_GUI_ACTIVE')/8.0
        line='%-66s cyc %8.0f  valu_busy %4.0f%%  valu/wave %6.0f  lds_busy %4.0f%%  lds_conf %4.0f%%  wait_inst %4.0f%%  wait_lds %4.0f%%  mfma/wave %5.0f  mfma16_util %4.0f%%  mfma_busy_raw %10.0f  waves %7.0f\n'%(
            key, cyc, 100*g('SQ_ACTIVE_INST_VALU')*4/1024/max(cyc,1), g('SQ_INSTS_VALU')/max(g('SQ_WAVES'),1),
            100*g('SQ_LDS_IDX_ACTIVE')/256/max(cyc,1), 100*g('SQ_LDS_BANK_CONFLICT')/max(g('SQ_LDS_IDX_ACTIVE'),1),
            100*g('SQ_WAIT_INST_ANY')/max(g('SQ_WAVE_CYCLES'),1), 100*g('SQ_WAIT_INST_LDS')/max(g('SQ_WAVE_CYCLES'),1),
            g('SQ_INSTS_MFMA')/max(g('SQ_WAVES'),1), 100*g('SQ_INSTS_MFMA')*16/1024/max(cyc,1), g('SQ_VALU_MFMA_BUSY_CYCLES'), g('SQ_WAVES'))
        print(line,end=''); fo.write(line)
PY
