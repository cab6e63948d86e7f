#!/usr/bin/env python
"""Two (or more) independent samples in flight: one HIP-graph replay stream per sample slot, so that kernels bound by different
units (vector-memory pipe for the gathers, matrix cores for the convolutions, VALU for the stem / aggregation) can share the
CUs.  Throughput of S concurrent replay streams vs one.  Development aid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
import patchmatchnet_amd as P
from patchmatchnet_amd.graph import GraphedForward

dev = torch.device("cuda", 0)
model = P.PatchmatchNet(**bench.DEFAULT_KW); bench.load_weights(model); model = model.to(dev).eval()
samples = bench.make_samples(4, 6, 1200, 1600, dev, 0)
STEPS = int(os.environ.get("STEPS", "96"))
with torch.no_grad():
    for S in (1, 2, 3, 4):
        streams = [torch.cuda.Stream(dev) for _ in range(S)]
        graphed = [GraphedForward(model) for _ in range(S)]
        for k in range(S):  # capture (on the stream it will replay on)
            with torch.cuda.stream(streams[k]):
                s = samples[k % 4]
                graphed[k](list(s["images"]), s["intrinsics"], s["extrinsics"], s["depth_min"], s["depth_max"])
        torch.cuda.synchronize()
        def run(n):
            for i in range(n):
                k = i % S
                s = samples[i % 4]
                with torch.cuda.stream(streams[k]):
                    graphed[k](list(s["images"]), s["intrinsics"], s["extrinsics"], s["depth_min"], s["depth_max"])
        run(2 * S)
        torch.cuda.synchronize()
        t = time.perf_counter()
        run(STEPS)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print(f"{S} stream(s): {dt / STEPS * 1e3:.3f} ms per depth map, {STEPS / dt:.1f} depth-maps/s", flush=True)
        del graphed
        torch.cuda.empty_cache()
