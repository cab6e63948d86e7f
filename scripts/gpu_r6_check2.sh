#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python scripts/scene_sigcheck.py 2>&1 | grep -a "cfg5" | tee gpurun_out/r06_check2.log
timeout 1500 python -m pytest tests/test_fullsize_parity.py tests/test_overlap_gpu.py tests/test_plan_gpu.py tests/test_hip_parity.py -q -m gpu -x -k "cfg3_scene_end or cfg5_scene_end or overlap or plan or feature_weight or golden or kernels" 2>&1 | grep -av "Warning\|warnings.warn\|return \|^$\|pin_memory" | tail -15 | tee -a gpurun_out/r06_check2.log
timeout 900 python bench.py --steps 60 --warmup 5 > gpurun_out/r06_bench_full.out 2> gpurun_out/r06_bench_full.err; echo "bench rc=$?" | tee -a gpurun_out/r06_check2.log
grep -a '^{' gpurun_out/r06_bench_full.out > gpurun_out/r06_bench_full.json
python - <<'PY' | tee -a gpurun_out/r06_check2.log
import json
j = json.load(open("gpurun_out/r06_bench_full.json"))
print("value", j["value"], "steady", j["steady_state"]["value"], "eager", j["single_stream_eager"]["value"], "verified", j["outputs_verified"]["steps"], "differ", j["outputs_verified"]["steps_that_differ_from_the_eager_forward"])
print("eval leg", j.get("eval_end_to_end"))
print("roofline", {k: j["roofline"][k] for k in ("achieved", "frac", "l1_frac", "kernel_ms_per_step", "traffic")})
print("cpu", {k: j["cpu_baseline"].get(k) for k in ("value", "cores", "kind")}, "ref rocm", (j.get("reference_rocm") or {}).get("value"), (j.get("reference_rocm") or {}).get("this_engine_over_reference"))
PY
