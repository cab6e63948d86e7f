#!/bin/bash
# gpurun payload (round 6, first GPU session): launch plans -- parity with the eager forward on the default hardware queues, the
# default bench line in plan mode beside round 5's graph mode, and the torch-free HIP-graph reproducer over a small env matrix.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r06_plan.log
: > $L
echo "== pytest tests/test_plan_gpu.py" | tee -a $L
timeout 1500 python -m pytest tests/test_plan_gpu.py -q -m gpu -x --durations=8 2>&1 | tail -25 | tee -a $L
echo "== bench, plan mode (default)" | tee -a $L
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r06_bench_plan.out 2> gpurun_out/r06_bench_plan.err; echo "rc=$?" | tee -a $L
grep -a '^{' gpurun_out/r06_bench_plan.out > gpurun_out/r06_bench_plan.json; tail -5 gpurun_out/r06_bench_plan.err | tee -a $L
echo "== bench, graph mode (round 5)" | tee -a $L
timeout 900 python bench.py --no-cpu-baseline --launch graph > gpurun_out/r06_bench_graph.out 2> gpurun_out/r06_bench_graph.err; echo "rc=$?" | tee -a $L
grep -a '^{' gpurun_out/r06_bench_graph.out > gpurun_out/r06_bench_graph.json
python - <<'PY' | tee -a $L
import json
for m in ("plan", "graph"):
    try:
        j = json.load(open(f"gpurun_out/r06_bench_{m}.json"))
        print(m, "value", j["value"], "steady", j["steady_state"]["value"], "eager", j["single_stream_eager"]["value"], "other", (j.get("value_other_input_mode") or {}).get("value"),
              "verified", j["outputs_verified"]["steps"], "differ", j["outputs_verified"]["steps_that_differ_from_the_eager_forward"], j["config"]["hardware_queues"], j["config"]["launch"][:60])
    except Exception as e:
        print(m, "no line", e)
PY
echo "== torch-free HIP-graph reproducer" | tee -a $L
R=scripts/repro/graph_queue_repro
for mode in graph manual eager; do
  timeout 300 $R $mode 3 60 32 12 2>&1 | tail -1 | tee -a $L
done
GPU_MAX_HW_QUEUES=1 timeout 300 $R graph 3 60 32 12 2>&1 | tail -1 | tee -a $L
HSA_ENABLE_SDMA=0 timeout 300 $R graph 3 60 32 12 2>&1 | tail -1 | tee -a $L
timeout 300 $R graph 3 60 128 24 2>&1 | tail -1 | tee -a $L
