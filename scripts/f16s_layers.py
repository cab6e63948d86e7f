#!/usr/bin/env python
"""Per-layer timing of pmn_conv2d_f16s on FeatureNet's own shapes (6 x 1200x1600 input) + the whole FeatureNet, and its difference
from the fp32 Winograd path.  Used by scripts/gpu_f16s_ab.sh with library variants (PMN_F16S_VARIANT selects the matching packing)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import patchmatchnet_amd as P
from patchmatchnet_amd import ops
dev = "cuda:0"
with np.load(os.path.join(ROOT, "tests/golden/params_000007.npz")) as z:
    sd = {k: torch.from_numpy(z[k]) for k in z.files}
m = P.PatchmatchNet(patchmatch_interval_scale=[0.005, 0.0125, 0.025], propagation_range=[6, 4, 2], patchmatch_iteration=[1, 2, 2],
                    patchmatch_num_sample=[8, 8, 16], propagate_neighbors=[0, 8, 16], evaluate_neighbors=[9, 9, 9])
m.load_state_dict(sd); m = m.to(dev).eval()
fn = m.feature
x = torch.rand(6, 3, 1200, 1600, device=dev)
def timeit(f, n=30):
    with torch.no_grad():
        for _ in range(5): f()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6
with torch.no_grad():
    fn.f16_split = False
    ref = {s: t.clone() for s, t in fn.forward_hip(x).items()}
    fn.f16_split = True
    new = fn.forward_hip(x)
    err = max(float((new[s] - ref[s]).abs().max() / ref[s].abs().max()) for s in (1, 2, 3))
    tot = timeit(lambda: fn.forward_hip(x), 20)
    pk = fn._packed()
    t = torch.empty((6, 1200, 1600, 8), device=dev)
    for i in range(6): ops.stem(x[i:i+1].contiguous(), *pk["conv0"], *pk["conv1"], out=t[i:i+1])
    per = []
    for i, (k, s, p) in enumerate(fn._SPEC):
        if i < 2: continue
        per.append("c%d %.1f" % (i, timeit(lambda: ops.conv2d_f16s(t, *pk[f"conv{i}_f16s"], k, s, relu=True))))
        t = ops.conv2d_f16s(t, *pk[f"conv{i}_f16s"], k, s, relu=True)
print("variant %s: FeatureNet %.1f us  maxdiff vs fp32 path %.2e | %s" % (os.environ.get("PMN_F16S_VARIANT", "0"), tot, err, "  ".join(per)))
