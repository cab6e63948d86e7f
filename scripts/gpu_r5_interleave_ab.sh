#!/bin/bash
# gpurun payload (round 5): PixelwiseNet launch, block vs interleaved hypothesis assignment of the two lane groups of a pixel --
# time at configs[1] and configs[4] (bench.py --eager, HIP events), bits, and the L2-miss traffic of the configs[4] launch (FETCH_SIZE)
export TMPDIR=/tmp
O=gpurun_out/r05_il; mkdir -p $O
for r in 1 2; do for v in g2 g2il; do
  timeout 300 python scripts/bench_with_lib.py build/pw/libpmn_hip_$v.so --eager --steps 30 --warmup 5 --roofline-steps 40 --no-cpu-baseline --steady-seconds 0 --settle-seconds 0.5 2>/dev/null | grep '^{' > $O/cfg2_${v}_$r.json
done; done
for v in g2 g2il; do
  timeout 300 python scripts/bench_with_lib.py build/pw/libpmn_hip_$v.so --eager --width 3072 --height 2048 --views 10 --samples 2 --steps 8 --warmup 2 --roofline-steps 8 --no-cpu-baseline --steady-seconds 0 --settle-seconds 0.3 2>/dev/null | grep '^{' > $O/cfg5_$v.json
  timeout 300 python scripts/ab_forward_bits.py --lib build/pw/libpmn_hip_$v.so --out $O/bits_$v.npz > /dev/null 2>&1
  rm -rf $O/pmc_$v; (cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$v -o pmc -- python $GRAFT_REPO_ROOT/scripts/bench_with_lib.py $GRAFT_REPO_ROOT/build/pw/libpmn_hip_$v.so --eager --width 3072 --height 2048 --views 10 --samples 2 --steps 3 --warmup 1 --roofline-steps 4 --no-cpu-baseline --steady-seconds 0 --settle-seconds 0 > /dev/null 2>&1)
done
python scripts/ab_forward_bits.py --compare $O/bits_g2.npz $O/bits_g2il.npz | grep -v "equal bits"; echo "bits rc=$?"
rm -f $O/bits_*.npz
python - <<'PY'
import json, glob, csv, os
O='gpurun_out/r05_il/'
for f in sorted(glob.glob(O+'cfg*_*.json')):
    try: j=json.load(open(f))
    except Exception as e: print(f,'no line'); continue
    per=j['roofline']['per_shape']
    print(os.path.basename(f), 'kernel_ms %.4f'%j['roofline']['kernel_ms_per_step'], {k.split('_')[0]+k.split('_')[1]+('p' if k.endswith('pixelwise') else ''): round(v['ms_avg']*1e3,1) for k,v in per.items()})
for v in ('g2','g2il'):
    vals=[]
    for f in glob.glob(O+'pmc_%s/**/*counter_collection.csv'%v, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'pixelwise_wave_kernel' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE': vals.append(float(r['Counter_Value']))
    if vals: print(v,'cfg5 pixelwise FETCH_SIZE x2 = %.1f MB per launch (%d launches)'%(2*sum(vals)/len(vals)*1024/1e6,len(vals)))
PY
rm -rf $O/pmc_g2 $O/pmc_g2il
