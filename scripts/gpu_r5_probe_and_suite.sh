#!/bin/bash
# gpurun payload (round 5): the ROCm-parity attribution probe on two libraries, then the whole GPU suite on the product library
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_parity
for v in default posefma; do
  lib=""; [ $v = posefma ] && lib="--lib build/pw/libpmn_hip_posefma.so"
  timeout 600 python scripts/rocm_parity_probe.py --tag $v $lib 2>gpurun_out/r05_parity/probe_$v.err | grep '^{' > gpurun_out/r05_parity/probe_$v.json
done
python - <<'PY'
import json
for v in ("default", "posefma"):
    try:
        j = json.load(open(f"gpurun_out/r05_parity/probe_{v}.json"))
    except Exception as e:
        print(v, "no result", e); continue
    for leg in ("engine_projections", "torch_rocm_projections"):
        print(v, leg, {k: "%.2e/%.1e" % (x["frac_over_1e-3"], x["max"]) for k, x in j[leg].items()})
    print(v, j["rel_proj_max_diff_over_scale"])
PY
bash scripts/gpu_check.sh
