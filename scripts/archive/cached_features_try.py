#!/usr/bin/env python
"""Rate of the forward when the feature pyramids of all views are already on the device (eval.py's encode-once path: FeatureNet
runs once per VIEW of a scan, the samples then run cascade + refinement only): graph replay, 1 and 3 samples in flight."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
import patchmatchnet_amd as P
from patchmatchnet_amd.graph import GraphedForward

dev = torch.device("cuda", 0)
model = P.PatchmatchNet(**bench.DEFAULT_KW); bench.load_weights(model); model = model.to(dev).eval()
samples = bench.make_samples(3, 6, 1200, 1600, dev, 0)
STEPS = 120
with torch.no_grad():
    feats = [model.extract_features(list(s["images"])) for s in samples]
    for S in (1, 3):
        streams = [torch.cuda.Stream(dev) for _ in range(S)]
        slots = [GraphedForward(model) for _ in range(S)]
        def run(n):
            for i in range(n):
                k, s, f = i % S, samples[i % 3], feats[i % 3]
                with torch.cuda.stream(streams[k]):
                    slots[k]([s["images"][0]] * 6, s["intrinsics"], s["extrinsics"], s["depth_min"], s["depth_max"], features=f)
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        run(2 * S)
        torch.cuda.synchronize()
        t = time.perf_counter()
        run(STEPS)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print(f"cached features, {S} in flight: {dt / STEPS * 1e3:.3f} ms per depth map, {STEPS / dt:.1f} depth-maps/s", flush=True)
