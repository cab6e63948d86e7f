#!/usr/bin/env python
"""A/B of the fused stem (pmn_stem_f16s, six 1600x1200 views): one process per library build; prints the median launch time by HIP
events, the whole FeatureNet's time, and a SHA-256 of the stem's output bits (two builds must print the same digest).
    python scripts/stem_ab.py [--lib build/pw/libpmn_hip_stem16.so]"""
import argparse
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--height", type=int, default=1200)
ap.add_argument("--width", type=int, default=1600)
a = ap.parse_args()
if a.lib:
    from patchmatchnet_amd import _lib
    _lib.LIB_PATH = os.path.abspath(a.lib)
import bench  # noqa: E402
import patchmatchnet_amd as P  # noqa: E402
from patchmatchnet_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
model = P.PatchmatchNet(**bench.DEFAULT_KW)
bench.load_weights(model)
model = model.to(dev).eval()
fn = model.feature
pk = fn._packed()
g = torch.Generator().manual_seed(5)
imgs = [torch.rand(1, 3, a.height, a.width, generator=g).to(dev) for _ in range(6)]


def timed(f, reps=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


with torch.no_grad():
    out = torch.empty((6, a.height, a.width, 8), dtype=torch.float32, device=dev)

    def stem_six():
        for i, im in enumerate(imgs):
            ops.stem_f16s(im, *pk["conv0"], *pk["conv1_f16s"], out=out[i:i + 1])

    t_stem = timed(stem_six)
    digest = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
    t_fn = timed(lambda: fn.forward_hip(imgs), reps=15)
    # odd sizes: the scalar staging path and partial tiles
    odd = torch.rand(2, 3, 70, 90, generator=g).to(dev)
    o2 = ops.stem_f16s(odd, *pk["conv0"], *pk["conv1_f16s"])
    d2 = hashlib.sha256(o2.cpu().numpy().tobytes()).hexdigest()[:16]
print(f"lib={a.lib or 'tree'} stem_six_views_us={t_stem:.1f} featurenet_us={t_fn:.1f} digest={digest} odd_digest={d2}")
