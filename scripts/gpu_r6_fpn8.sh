#!/bin/bash
# the cc32 form of the FPN's 1/8 level: same bits as before (whole forward), parity + overlap + plan tests, a bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r06_fpn8_check.log
: > $L
python scripts/ab_forward_bits.py --lib build/libpmn_hip_before_fpn8.so --out /tmp/a.npz > /dev/null 2>&1
python scripts/ab_forward_bits.py --out /tmp/b.npz > /dev/null 2>&1
python scripts/ab_forward_bits.py --compare /tmp/a.npz /tmp/b.npz 2>&1 | tail -12 >> $L
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_plan_gpu.py tests/test_overlap_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -av "Warning\|warnings.warn\|^$" | tail -3 >> $L
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | grep -a '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('bench', j['value'], 'steady', j['steady_state']['value'], 'verified', j['outputs_verified']['steps'], 'differ', j['outputs_verified']['steps_that_differ_from_the_eager_forward'])" >> $L
cat $L
