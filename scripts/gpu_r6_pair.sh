#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_overlap_gpu.py tests/test_plan_gpu.py -q -m gpu -x -k "pair or fused_conv34 or featurenet_hip_matches_miopen or overlap or plan" 2>&1 | grep -av "Warning\|warnings.warn\|^$" | tail -8 | tee gpurun_out/r06_pair.log
python - <<'PY' 2>&1 | grep -av Warn | tee -a gpurun_out/r06_pair.log
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import bench, patchmatchnet_amd as P
dev = torch.device("cuda", 0)
model = P.PatchmatchNet(**bench.DEFAULT_KW); bench.load_weights(model); model = model.to(dev).eval()
s = bench.make_samples(1, 6, 1200, 1600, dev, 0)[0]
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
with torch.no_grad():
    for fuse in (True, False, True, False):
        model.feature.fuse_conv34 = fuse
        print("FeatureNet, six 1600x1200 views, fuse_conv34 =", fuse, ": %.1f us" % t(lambda: model.feature.forward_hip(s["images"])))
PY
timeout 600 python bench.py --no-cpu-baseline --verify-steps 48 2>/dev/null | grep -a '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('bench value', j['value'], 'steady', j['steady_state']['value'], 'eager', j['single_stream_eager']['value'], 'differ', j['outputs_verified']['steps_that_differ_from_the_eager_forward'])" | tee -a gpurun_out/r06_pair.log
