#!/bin/bash
# Focused GPU pass for the FeatureNet convolutions: parity tests of the conv kernels, FeatureNet wall time with the fp16-split
# kernels vs the fp32 Winograd kernels, rocprofv3 kernel stats of the eager bench.  Outputs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_parity.py -q -x -k "conv or winograd or f16 or featurenet or FeatureNet or stem or fpn" --durations=8 > gpurun_out/pytest_conv.log 2>&1; echo "exit $?" >> gpurun_out/pytest_conv.log
tail -15 gpurun_out/pytest_conv.log
timeout 300 python - > gpurun_out/featurenet_ab.log 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import patchmatchnet_amd as P
dev = "cuda:0"
with np.load("tests/golden/params_000007.npz") as z:
    sd = {k: torch.from_numpy(z[k]) for k in z.files}
m = P.PatchmatchNet(patchmatch_interval_scale=[0.005, 0.0125, 0.025], propagation_range=[6, 4, 2], patchmatch_iteration=[1, 2, 2],
                    patchmatch_num_sample=[8, 8, 16], propagate_neighbors=[0, 8, 16], evaluate_neighbors=[9, 9, 9])
m.load_state_dict(sd); m = m.to(dev).eval()
fn = m.feature
x = torch.rand(6, 3, 1200, 1600, device=dev)
def timeit(f, n=20):
    with torch.no_grad():
        for _ in range(5): f()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
with torch.no_grad():
    fn.f16_split = False
    ref = {s: t.clone() for s, t in fn.forward_hip(x).items()}
    t_old = timeit(lambda: fn.forward_hip(x))
    fn.f16_split = True
    new = fn.forward_hip(x)
    err = {s: float((new[s] - ref[s]).abs().max() / ref[s].abs().max()) for s in (1, 2, 3)}
    t_new = timeit(lambda: fn.forward_hip(x))
    mi = fn(x[:1])
    err_mi = {s: float((new[s][:1].permute(0, 3, 1, 2) - mi[s]).abs().max() / mi[s].abs().max()) for s in (1, 2, 3)}
print("FeatureNet 6x1200x1600: fp32 Winograd/MFMA kernels %.3f ms, fp16-split kernels %.3f ms" % (t_old, t_new))
print("max rel diff f16-split vs fp32 kernels", err, " vs MIOpen (image 0)", err_mi)
# per layer
pk = fn._packed()
from patchmatchnet_amd import ops
with torch.no_grad():
    t = torch.empty((6, 1200, 1600, 8), device=dev)
    for i in range(6): ops.stem(x[i:i+1].contiguous(), *pk["conv0"], *pk["conv1"], out=t[i:i+1])
    print("stem x6 %.1f us" % (1e3 * timeit(lambda: [ops.stem(x[i:i+1].contiguous(), *pk["conv0"], *pk["conv1"], out=t[i:i+1]) for i in range(6)])))
    for i, (k, s, p) in enumerate(fn._SPEC):
        if i < 2: continue
        a = 1e3 * timeit(lambda: ops.conv2d_f16s(t, *pk[f"conv{i}_f16s"], k, s, relu=True))
        if f"conv{i}_wino" in pk:
            b = 1e3 * timeit(lambda: ops.conv3x3_wino(t, *pk[f"conv{i}_wino"], relu=True))
        else:
            b = 1e3 * timeit(lambda: ops.conv5x5s2_wino(t, *pk[f"conv{i}_wino5"], relu=True))
        nb = t.numel() * 4
        t2 = ops.conv2d_f16s(t, *pk[f"conv{i}_f16s"], k, s, relu=True)
        gb = (nb + t2.numel() * 4) / 1e9
        print("conv%d in %s: f16-split %.1f us (%.2f TB/s in+out)   fp32 winograd %.1f us" % (i, tuple(t.shape), a, gb / a * 1e3, b))
        t = t2
PY
cat gpurun_out/featurenet_ab.log
bash scripts/gpu_profile.sh 20 > gpurun_out/profile_eager.log 2>&1; tail -45 gpurun_out/profile_eager.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_f16s.log 2>&1; tail -1 gpurun_out/bench_f16s.log | cut -c1-900
