#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/r06_flight.log
: > $L
timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -k "f16_domain or feature_weight or featurenet_hip_matches_miopen" 2>&1 | tail -3 | tee -a $L
for r in 1 2 3; do for f in 2 3; do
  timeout 600 python bench.py --no-cpu-baseline --in-flight $f --verify-steps 24 --roofline-steps 4 --steady-seconds 3 2>/dev/null | grep -a '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('in-flight $f run $r: value', j['value'], 'steady', j['steady_state']['value'], 'other', j['value_other_input_mode']['value'], 'differ', j['outputs_verified']['steps_that_differ_from_the_eager_forward'])" | tee -a $L
done; done
