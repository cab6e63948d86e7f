#!/bin/bash
# gpurun payload: window statistics + SQ counters of the matrix-core pmn_warp_correlate on a real forward's arguments
export TMPDIR=/tmp
export PMN_EXPERIMENTAL=1  # the matrix-core formulation lives in the research build
OUT=$GRAFT_REPO_ROOT/gpurun_out/corr_pmc
rm -rf $OUT; mkdir -p $OUT
timeout 600 python scripts/corr_ab.py --windows 2>&1 | grep -v amdgpu.ids | tee $OUT/windows.txt
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python $GRAFT_REPO_ROOT/scripts/corr_ab.py --impls ${IMPLS:-mfma} --reps 5 > $OUT/p$i.log 2>&1)
done <<LIST
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS
SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD
SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR
LIST
python - <<'PY'
import csv,glob,collections,os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/corr_pmc'
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out+'/p*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'corr_mfma' not in k and 'gather_corr' not in k: continue
        agg[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out+'/summary.txt','w') as fo:
    for key,cs in agg.items():
        g=lambda n: (sum(cs[n])/len(cs[n])) if n in cs else 0.0
        cyc=g('GRBM_GUI_ACTIVE')/8.0
        wc=max(g('SQ_WAVE_CYCLES'),1)
        line='%-72s cyc %8.0f valu_busy %4.0f%% valu/wave %6.0f salu/wave %6.0f lds/wave %5.0f vmem_rd/wave %5.0f lds_busy %4.0f%% lds_conf %4.0f%% wait_any %4.0f%% wait_inst %4.0f%% active %4.0f%% mfma/wave %5.0f mfma_util %4.0f%% waves %7.0f wavecyc/wave %8.0f\n'%(
            key, cyc, 100*g('SQ_ACTIVE_INST_VALU')*4/1024/max(cyc,1), g('SQ_INSTS_VALU')/max(g('SQ_WAVES'),1), g('SQ_INSTS_SALU')/max(g('SQ_WAVES'),1),
            g('SQ_INSTS_LDS')/max(g('SQ_WAVES'),1), g('SQ_INSTS_VMEM_RD')/max(g('SQ_WAVES'),1),
            100*g('SQ_LDS_IDX_ACTIVE')/256/max(cyc,1), 100*g('SQ_LDS_BANK_CONFLICT')/max(g('SQ_LDS_IDX_ACTIVE'),1),
            100*g('SQ_WAIT_ANY')/wc, 100*g('SQ_WAIT_INST_ANY')/wc, 100*g('SQ_ACTIVE_INST_ANY')/wc,
            g('SQ_INSTS_MFMA')/max(g('SQ_WAVES'),1), 100*g('SQ_INSTS_MFMA')*32/1024/max(cyc,1), g('SQ_WAVES'), 4*g('SQ_WAVE_CYCLES')/max(g('SQ_WAVES'),1))
        print(line,end=''); fo.write(line)
PY
rm -rf $OUT/p[0-9]*
