#!/usr/bin/env python
"""torch.profiler view of one bench step: which ATen ops / memcpys surround the pmn_* launches (development aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
import patchmatchnet_amd as P
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
model = P.PatchmatchNet(**bench.DEFAULT_KW); bench.load_weights(model); model = model.to(dev).eval()
s = bench.make_samples(1, 6, 1200, 1600, dev, 0)[0]
def step():
    return model(list(s["images"]), s["intrinsics"].clone(), s["extrinsics"], s["depth_min"], s["depth_max"])
with torch.no_grad():
    for _ in range(3): step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60))
ev = [e for e in prof.events() if "copy" in e.name.lower() or "Memcpy" in e.name]
from collections import Counter
c = Counter()
for e in ev:
    st = [f for f in (e.stack or []) if "patchmatchnet_amd" in f or "bench" in f]
    c[(e.name, tuple(st[:3]))] += 1
for k, v in c.most_common(40):
    print(v, k)
