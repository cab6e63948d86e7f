#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r06_graphq.log
: > $L
for q in 4 8; do
GPU_MAX_HW_QUEUES=$q timeout 900 python bench.py --no-cpu-baseline --launch graph --verify-steps 192 > gpurun_out/r06_bench_graph_q$q.out 2> gpurun_out/r06_bench_graph_q$q.err; echo "q=$q rc=$?" | tee -a $L
grep -a '^{' gpurun_out/r06_bench_graph_q$q.out > gpurun_out/r06_bench_graph_q$q.json
done


python - <<'PY' | tee -a $L
import json
for m in ("graph_q4", "graph_q8"):
    try:
        j = json.load(open(f"gpurun_out/r06_bench_{m}.json"))
        print(m, "value", j["value"], "steady", j["steady_state"]["value"], "other", (j.get("value_other_input_mode") or {}).get("value"),
              "verified", j["outputs_verified"]["steps"], "differ", j["outputs_verified"]["steps_that_differ_from_the_eager_forward"], j["config"]["hardware_queues"], j["config"]["launch"][:40])
    except Exception as e:
        print(m, "no line", e)
PY
