#!/bin/bash
# gpurun payload (round 5): the PixelwiseNet launch A/B on ONE box -- scripts/build_pw_variants.sh's five libraries, each timed twice
# (interleaved) through bench.py --eager (HIP events around every pmn_warp_correlate launch), then the same forward's BITS compared
# (scripts/ab_forward_bits.py), then the new parity test against the reference on ROCm and the bench-contract tests.
export TMPDIR=/tmp
OUT=gpurun_out/r05_pw
mkdir -p $OUT
VARIANTS="base g0 g4 g2 g2w5"
for round in 1 2; do
  for v in $VARIANTS; do
    timeout 300 python scripts/bench_with_lib.py build/pw/libpmn_hip_$v.so --eager --steps 40 --warmup 5 --roofline-steps 40 --no-cpu-baseline \
        --steady-seconds 0 --settle-seconds 0.5 2>$OUT/bench_${v}_$round.err | grep '^{' > $OUT/bench_${v}_$round.json
  done
done
python - <<'PY' | tee gpurun_out/r05_pw/summary.txt
import json, glob, os
rows = {}
for f in sorted(glob.glob('gpurun_out/r05_pw/bench_*_?.json')):
    name = os.path.basename(f)[6:-5]
    try:
        j = json.load(open(f))
    except Exception as e:
        print(name, 'no line', e); continue
    per = j['roofline']['per_shape']
    print(name, 'eager %.1f/s' % j['value'], 'kernel_ms %.4f' % j['roofline']['kernel_ms_per_step'],
          ' '.join('%s=%.1f' % (k.split('_')[0] + k.split('_')[1] + ('p' if k.endswith('pixelwise') else ''), v['ms_avg'] * 1e3) for k, v in per.items()))
PY
for v in $VARIANTS; do
  timeout 300 python scripts/ab_forward_bits.py --lib build/pw/libpmn_hip_$v.so --out $OUT/bits_$v.npz > $OUT/bits_$v.log 2>&1
done
for v in g0 g4 g2 g2w5; do
  echo "== bits $v vs base" | tee -a $OUT/summary.txt
  python scripts/ab_forward_bits.py --compare $OUT/bits_base.npz $OUT/bits_$v.npz 2>&1 | grep -v "equal bits" | tee -a $OUT/summary.txt
  echo "rc=$?" | tee -a $OUT/summary.txt
done
rm -f $OUT/bits_*.npz
echo "== rocm parity + archive + bench contract tests" | tee -a $OUT/summary.txt
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests/test_fullsize_parity.py tests/test_bench_gpu.py tests/test_reference_archive.py -q -m gpu -k "rocm or bench or archive" --durations=5 2>&1 | tail -30 | tee -a $OUT/summary.txt
