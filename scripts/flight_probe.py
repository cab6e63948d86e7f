#!/usr/bin/env python
"""bench.py's timed loop under variations of HOW the samples in flight share the GPU: number of slots, HIP stream priorities.
One box, alternating rounds.   python scripts/flight_probe.py [--seconds 2.5]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import bench
import patchmatchnet_amd as P
from patchmatchnet_amd.graph import GraphedForward

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=2.5)
ap.add_argument("--only", nargs="*", default=None, help="configuration names to run (default: all)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
model = P.PatchmatchNet(**bench.DEFAULT_KW)
bench.load_weights(model)
model = model.to(dev).eval()
samples = bench.make_samples(12, 6, 1200, 1600, dev, 0)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range", lo, hi, flush=True)
CONFIGS = {"3 equal": [0, 0, 0], "3 high/normal/normal": [-1, 0, 0], "3 descending": [-1, 0, 1] if lo >= 1 else [-1, -1, 0],
           "2 equal": [0, 0], "4 equal": [0, 0, 0, 0], "2 high/normal": [-1, 0]}
if a.only:
    CONFIGS = {k: v for k, v in CONFIGS.items() if k in a.only}
main = torch.cuda.current_stream(dev)
setups = {}
with torch.no_grad():
    for name, prios in CONFIGS.items():
        streams = [torch.cuda.Stream(dev, priority=p) for p in prios]
        slots = [GraphedForward(model, inputs_in_place=True) for _ in prios]
        for st in streams:
            st.wait_stream(main)
        setups[name] = (streams, slots)

    def replay(name, i):
        streams, slots = setups[name]
        S = len(streams)
        k, s = i % S, samples[i % len(samples)]
        with torch.cuda.stream(streams[k]):
            return slots[k]([im for im in s["images"]], s["intrinsics"], s["extrinsics"], s["depth_min"], s["depth_max"])

    def run(name, seconds):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(6):
                replay(name, n); n += 1
            if n % 24 == 0:
                setups[name][0][0].synchronize()
        torch.cuda.synchronize()
        return n / (time.perf_counter() - t0)

    for name in CONFIGS:
        for i in range(8):
            replay(name, i)
        torch.cuda.synchronize()
    run(next(iter(CONFIGS)), 1.5)
    for r in range(2):
        print(f"round {r}: " + "   ".join(f"{name}: {run(name, a.seconds):6.1f}" for name in CONFIGS), flush=True)
