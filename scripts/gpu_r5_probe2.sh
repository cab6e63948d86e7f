#!/bin/bash
# gpurun payload (round 5): the ROCm-parity probe on the attribution build that takes divisions by host scalars as ATen's GPU kernels do
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_parity
for v in atendiv; do
  timeout 600 python scripts/rocm_parity_probe.py --tag $v --lib build/pw/libpmn_hip_$v.so 2>gpurun_out/r05_parity/probe_$v.err | grep '^{' > gpurun_out/r05_parity/probe_$v.json
done
python - <<'PY'
import json
for v in ("atendiv",):
    j = json.load(open(f"gpurun_out/r05_parity/probe_{v}.json"))
    for leg in ("engine_projections", "torch_rocm_projections"):
        print(v, leg, {k: "%.2e/%.1e" % (x["frac_over_1e-3"], x["max"]) for k, x in j[leg].items()})
PY
