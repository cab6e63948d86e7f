#!/bin/bash
# gpurun payload: the whole GPU suite on the product library, the opt-in research-build tests, smoke(), a default bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
echo "== pytest -m gpu (product)" | tee gpurun_out/pytest_gpu.log
timeout 2400 python -m pytest tests/ -q -m gpu -x --durations=8 2>&1 | tail -25 | tee -a gpurun_out/pytest_gpu.log
echo "== pytest research build" | tee -a gpurun_out/pytest_gpu.log
PMN_EXPERIMENTAL=1 timeout 1500 python -m pytest tests/test_corr_mfma.py tests/test_gather_win.py tests/test_hip_parity.py -q -m gpu -k "corr or gather or windowed or winograd or mfma or research" 2>&1 | tail -6 | tee -a gpurun_out/pytest_gpu.log
echo "== smoke" | tee -a gpurun_out/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a gpurun_out/pytest_gpu.log
echo "== bench (no cpu baseline)" | tee -a gpurun_out/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep '^{' > gpurun_out/bench_quick.json; python - <<'PY'
import json
j=json.load(open('gpurun_out/bench_quick.json'))
print('value',j['value'],'steady',j['steady_state']['value'],'eager',j['single_stream_eager']['value'],'frac',j['roofline']['frac'],'traffic',j['roofline']['traffic'])
print(j['roofline']['per_shape'])
PY
