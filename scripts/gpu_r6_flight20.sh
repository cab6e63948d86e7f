#!/bin/bash
# The driver's K = 20 window for 2 / 3 / 4 samples in flight, five runs each, interleaved -> gpurun_out/r06_flight20.log
mkdir -p gpurun_out
L=gpurun_out/r06_flight20.log
: > $L
for run in 1 2 3 4 5; do
  for f in 2 3 4; do
    timeout 300 python bench.py --steps 20 --warmup 5 --in-flight $f --no-cpu-baseline --verify-steps 24 2>/dev/null | grep -a '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('in-flight $f run $run: value %.1f steady %.1f other %.1f differ %d' % (j['value'], j['steady_state']['value'], j['value_other_input_mode']['value'], j['outputs_verified']['steps_that_differ_from_the_eager_forward']))" >> $L
  done
done
cat $L
