#!/usr/bin/env python
"""Times FeatureNet (the MIOpen part of the forward) in a few PyTorch-ROCm formulations at cfg-2 shape (6 x 3x1200x1600)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.nn as nn, torch.nn.functional as F
import patchmatchnet_amd as P

dev = "cuda:0"
with np.load(os.path.join(ROOT, "tests", "golden", "params_000007.npz")) as z:
    sd = {k: torch.from_numpy(z[k]) for k in z.files}
m = P.PatchmatchNet(patchmatch_interval_scale=[0.005, 0.0125, 0.025], propagation_range=[6, 4, 2], patchmatch_iteration=[1, 2, 2],
                    patchmatch_num_sample=[8, 8, 16], propagate_neighbors=[0, 8, 16], evaluate_neighbors=[9, 9, 9])
m.load_state_dict(sd); m = m.to(dev).eval()
fn = m.feature
x = torch.rand(6, 3, 1200, 1600, device=dev)

def timeit(f, n=10):
    with torch.no_grad():
        for _ in range(3): f()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

def fold(cbr):
    w = cbr.conv.weight.double(); bn = cbr.bn
    s = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
    return (w * s.view(-1, 1, 1, 1)).float(), (bn.bias.double() - bn.running_mean.double() * s).float()

folded = {i: fold(getattr(fn, f"conv{i}")) for i in range(11)}
spec = {i: (getattr(fn, f"conv{i}").conv.stride, getattr(fn, f"conv{i}").conv.padding) for i in range(11)}

def run_folded(x, fused):
    def cbr(i, t):
        w, b = folded[i]; st, pd = spec[i]
        if fused:
            return torch.ops.aten.miopen_convolution_relu(t, w, b, st, pd, (1, 1), 1)
        return F.relu(F.conv2d(t, w, b, st, pd), inplace=True)
    t = x
    for i in range(0, 5): t = cbr(i, t)
    half = t
    for i in range(5, 8): t = cbr(i, t)
    quarter = t
    for i in range(8, 11): t = cbr(i, t)
    eighth = t
    out3 = fn.output1(eighth)
    top = F.interpolate(eighth, scale_factor=2.0, mode="bilinear", align_corners=False) + fn.inner1(quarter)
    out2 = fn.output2(top)
    top = F.interpolate(top, scale_factor=2.0, mode="bilinear", align_corners=False) + fn.inner2(half)
    return {3: out3, 2: out2, 1: fn.output3(top)}

ref = None
with torch.no_grad():
    ref = fn(x)
print("baseline nchw            %.3f ms" % timeit(lambda: fn(x)))
with torch.no_grad():
    o = fn.forward_hip(x)
err = max(float((o[s].permute(0, 3, 1, 2) - ref[s]).abs().max() / ref[s].abs().max()) for s in (1, 2, 3))
print("HIP pmn_conv2d           %.3f ms   max rel diff vs baseline %.2e" % (timeit(lambda: fn.forward_hip(x)), err))
if os.environ.get("FN_ONLY_HIP"):
    sys.exit(0)
for name, f in [("folded conv+bias, relu  ", lambda: run_folded(x, False)), ("folded miopen conv_relu ", lambda: run_folded(x, True))]:
    try:
        with torch.no_grad(): o = f()
        err = max(float((o[s] - ref[s]).abs().max()) for s in (1, 2, 3))
        print("%s %.3f ms   max abs diff vs baseline %.2e" % (name, timeit(f), err))
    except Exception as e:
        print(name, "FAILED", type(e).__name__, str(e)[:200])
xcl = x.contiguous(memory_format=torch.channels_last)
fncl = fn.to(memory_format=torch.channels_last)
try:
    with torch.no_grad(): o = fncl(xcl)
    err = max(float((o[s] - ref[s]).abs().max()) for s in (1, 2, 3))
    print("channels_last            %.3f ms   max abs diff %.2e  out strides %s" % (timeit(lambda: fncl(xcl)), err, o[3].stride()))
    for name, f in [("cl folded conv+bias,relu", lambda: run_folded(xcl, False)), ("cl folded conv_relu     ", lambda: run_folded(xcl, True))]:
        with torch.no_grad(): o = f()
        err = max(float((o[s] - ref[s]).abs().max()) for s in (1, 2, 3))
        print("%s %.3f ms   max abs diff %.2e" % (name, timeit(f), err))
except Exception as e:
    print("channels_last FAILED", type(e).__name__, str(e)[:200])
# refinement
rf = m.upsample_net.to(memory_format=torch.contiguous_format)
img = x[:1].contiguous(); d0 = 425 + 510 * torch.rand(1, 1, 600, 800, device=dev)
dmin = torch.tensor([425.0], device=dev); dmax = torch.tensor([935.0], device=dev)
print("refinement               %.3f ms" % timeit(lambda: rf(img, d0, dmin, dmax)))
