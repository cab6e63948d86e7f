#!/usr/bin/env python
"""FeatureNet.forward_hip per VIEW as a function of the views per pass: 6 (one sample, what the forward does) vs 12 / 18 (the views of
two / three samples in flight in one pass) at 1600x1200.  Data point for DESIGN.md section 9 (not a product path)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
import patchmatchnet_amd as P

dev = torch.device("cuda", 0)
model = P.PatchmatchNet(**bench.DEFAULT_KW)
bench.load_weights(model)
fn = model.to(dev).eval().feature
x = torch.rand(18, 3, 1200, 1600, device=dev)


def timeit(f, n=20):
    with torch.no_grad():
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            f()
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for rnd in range(2):
    for v in (6, 12, 18):
        ms = timeit(lambda: fn.forward_hip(x[:v]))
        print(f"round {rnd}: {v:2d} views per pass: {ms:7.3f} ms = {ms / v * 1e3:6.1f} us per view ({ms / v * 6:6.3f} ms per six)", flush=True)
