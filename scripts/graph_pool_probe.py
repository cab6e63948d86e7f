#!/usr/bin/env python
"""(1) Do the graphs of three GraphedForward slots live in DISTINCT allocator pools / address ranges?  (2) Do three pure-torch graphs
(no kernel of this repository) replayed concurrently on their own streams reproduce their eager results?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

dev = torch.device("cuda", 0)
# ---- (2) pure torch ----
torch.manual_seed(0)
S, R = 3, 300
xs = [torch.randn(512, 512, device=dev) for _ in range(S)]
w = [torch.randn(512, 512, device=dev) * 0.05 for _ in range(6)]


def net(x):
    for i in range(40):
        x = torch.tanh(x @ w[i % 6]) + 0.1 * x
        x = torch.nn.functional.avg_pool2d(x[None, None], 3, 1, 1)[0, 0] * 1.01
    return x


want = [net(x).clone() for x in xs]
torch.cuda.synchronize()
graphs, outs, streams = [], [], [torch.cuda.Stream(dev) for _ in range(S)]
for k in range(S):
    side = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        net(xs[k])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        o = net(xs[k])
    graphs.append(g)
    outs.append(o)
torch.cuda.synchronize()
bad = {}
for r in range(R):
    got = []
    for k in range(S):
        with torch.cuda.stream(streams[k]):
            graphs[k].replay()
            got.append(outs[k].clone())
    torch.cuda.synchronize()
    for k in range(S):
        if not torch.equal(got[k], want[k]):
            bad.setdefault(k, []).append(r)
print("pure-torch graphs replayed concurrently x%d:" % R, {k: len(v) for k, v in bad.items()} if bad else "every replay equals eager")

# ---- (1) pools of the engine's graphs ----
import goldenutil as GU  # noqa: E402
import synth  # noqa: E402
import patchmatchnet_amd as P  # noqa: E402
from patchmatchnet_amd.graph import GraphedForward  # noqa: E402
_, params, kw = GU.load_case("default")
model = P.PatchmatchNet(**kw)
model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
model = model.to(dev).eval()
H, W, nv = 96, 128, 3
intr, extr = synth.synthetic_cameras(nv, H, W)
slots = [GraphedForward(model) for _ in range(S)]
res = []
with torch.no_grad():
    for k in range(S):
        g = torch.Generator().manual_seed(100 + k)
        imgs = [torch.rand(1, 3, H, W, generator=g).to(dev) for _ in range(nv)]
        torch.cuda.synchronize()
        with torch.cuda.stream(streams[k]):
            d, c = slots[k](imgs, torch.from_numpy(intr).to(dev), torch.from_numpy(extr).to(dev), torch.tensor([425.0], device=dev),
                            torch.tensor([935.0], device=dev))
        torch.cuda.synchronize()
        res.append((d, c))
snap = torch.cuda.memory_snapshot()
pools = {}
for seg in snap:
    pools.setdefault(tuple(seg.get("segment_pool_id", (0, 0))), []).append((seg["address"], seg["total_size"], seg["stream"]))
for pid, segs in sorted(pools.items()):
    print("pool", pid, "segments", len(segs), "bytes", sum(s[1] for s in segs), "streams", sorted({s[2] for s in segs})[:4])


def pool_of(ptr):
    for pid, segs in pools.items():
        for a, n, _ in segs:
            if a <= ptr < a + n:
                return pid
    return None


for k, (d, c) in enumerate(res):
    print("slot", k, "depth output in pool", pool_of(d.data_ptr()), "confidence in pool", pool_of(c.data_ptr()))
# overlapping segments between different pools?
allsegs = sorted((a, a + n, pid) for pid, segs in pools.items() for a, n, _ in segs)
ov = [(x, y) for x, y in zip(allsegs, allsegs[1:]) if y[0] < x[1]]
print("overlapping segments:", ov[:3] if ov else "none")
