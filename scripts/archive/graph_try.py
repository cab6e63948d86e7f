#!/usr/bin/env python
"""HIP-graph capture of the whole PatchmatchNet.forward (torch.cuda.CUDAGraph over the ctypes launches): eager vs replay
step time at the bench workload.  Development aid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
import patchmatchnet_amd as P

dev = torch.device("cuda", 0)
model = P.PatchmatchNet(**bench.DEFAULT_KW); bench.load_weights(model); model = model.to(dev).eval()
samples = bench.make_samples(4, 6, 1200, 1600, dev, 0)
def step(s):
    return model(list(s["images"]), s["intrinsics"].clone(), s["extrinsics"], s["depth_min"], s["depth_max"])
with torch.no_grad():
    for s in samples: step(s)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(40): step(samples[i % 4])
    torch.cuda.synchronize()
    print(f"eager : {(time.perf_counter() - t) / 40 * 1e3:.3f} ms/step", flush=True)
    graphs, outs = [], []
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for s in samples: step(s)
    torch.cuda.current_stream().wait_stream(side)
    for s in samples:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            o = step(s)
        graphs.append(g); outs.append(o)
    torch.cuda.synchronize()
    for g in graphs: g.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(40): graphs[i % 4].replay()
    torch.cuda.synchronize()
    print(f"graph : {(time.perf_counter() - t) / 40 * 1e3:.3f} ms/step", flush=True)
    ref = step(samples[0]); graphs[0].replay(); torch.cuda.synchronize()
    print("depth max |eager - graph| (stage-3 noise differs run to run):", float((ref[0] - outs[0][0]).abs().max()))
