#!/usr/bin/env python
"""What do the per-sample input copies of a graph replay cost?  bench.py's timed loop (3 samples in flight, one HIP-graph slot each)
A: GraphedForward(model) -- every step copies its six images (138 MB) into the slot's static buffers (rounds 2-4);
B: GraphedForward(model, inputs_in_place=True) -- FeatureNet reads the samples where they are (pmn_stem_f16s_views), one image copied.
Alternates A B A B on one box.   python scripts/inplace_probe.py [--seconds 3]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import bench
import patchmatchnet_amd as P
from patchmatchnet_amd.graph import GraphedForward

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=3.0)
ap.add_argument("--in-flight", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda", 0)
model = P.PatchmatchNet(**bench.DEFAULT_KW)
bench.load_weights(model)
model = model.to(dev).eval()
samples = bench.make_samples(12, 6, 1200, 1600, dev, 0)
S = a.in_flight
streams = [torch.cuda.Stream(dev) for _ in range(S)]
slots = {False: [GraphedForward(model) for _ in range(S)], True: [GraphedForward(model, inputs_in_place=True) for _ in range(S)]}
main = torch.cuda.current_stream(dev)

def replay(i, in_place):
    k, s = i % S, samples[i % len(samples)]
    with torch.cuda.stream(streams[k]):
        return slots[in_place][k]([im for im in s["images"]], s["intrinsics"], s["extrinsics"], s["depth_min"], s["depth_max"])

with torch.no_grad():
    for st in streams:
        st.wait_stream(main)
    for i in range(2 * S):
        replay(i, False)
        replay(i, True)
    torch.cuda.synchronize()
    def run(in_place, seconds):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(6):
                replay(n, in_place); n += 1
            if n % 24 == 0:
                streams[n % S].synchronize()
        torch.cuda.synchronize()
        return n / (time.perf_counter() - t0)
    run(False, 1.5)
    for r in range(3):
        va = run(False, a.seconds); vb = run(True, a.seconds)
        print(f"round {r}: A copies {va:7.2f} depth-maps/s   B in place {vb:7.2f}   B/A {vb / va:.4f}", flush=True)
    print("captures per slot:", [s.captures for m in slots.values() for s in m])
