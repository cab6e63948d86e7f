#!/usr/bin/env python
"""Does GraphedForward with several slots in flight reproduce the eager forward bit for bit when the graphs are captured in the middle
of the run (as eval.py's plain path does)?  No DataLoader, no prefetcher, no writer: resident inputs, S slots on S streams.
    PMN_PROBE_ROOT=<tree> python scripts/graph_flake_probe.py [trials] [sync_before_replay]"""
import os
import sys

ROOT = os.environ.get("PMN_PROBE_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import goldenutil as GU  # noqa: E402
import synth  # noqa: E402
import patchmatchnet_amd as P  # noqa: E402
from patchmatchnet_amd.graph import GraphedForward  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 30
mode = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda", 0)
_, params, kw = GU.load_case("default")
model = P.PatchmatchNet(**kw)
model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
model = model.to(dev).eval()
H, W, nv, NS = 96, 128, 3, 4
intr, extr = synth.synthetic_cameras(nv, H, W)
samples = []
for s in range(NS):
    g = torch.Generator().manual_seed(100 + s)
    samples.append(dict(images=[torch.rand(1, 3, H, W, generator=g).to(dev) for _ in range(nv)], intrinsics=torch.from_numpy(intr).to(dev),
                        extrinsics=torch.from_numpy(extr).to(dev), dmin=torch.tensor([425.0], device=dev), dmax=torch.tensor([935.0], device=dev)))
torch.manual_seed(3)
want = []
with torch.no_grad():
    for s in samples:
        d, c, _ = model(list(s["images"]), s["intrinsics"].clone(), s["extrinsics"], s["dmin"], s["dmax"])
        want.append((d.clone(), c.clone()))
torch.cuda.synchronize()
main = torch.cuda.current_stream(dev)
for S in (2, 3):
    bad = {}
    for t in range(trials):
        slots = [GraphedForward(model) for _ in range(S)]
        streams = [torch.cuda.Stream(dev) for _ in range(S)]
        torch.manual_seed(3)
        got = []
        with torch.no_grad():
            for i, s in enumerate(samples):
                k = i % S
                streams[k].wait_stream(main)
                with torch.cuda.stream(streams[k]):
                    if mode == "sync":
                        torch.cuda.synchronize()
                    d, c = slots[k](list(s["images"]), s["intrinsics"], s["extrinsics"], s["dmin"], s["dmax"])
                    got.append((d.clone(), c.clone()))
        torch.cuda.synchronize()
        for i, ((d, c), (wd, wc)) in enumerate(zip(got, want)):
            if not torch.equal(d, wd) or not torch.equal(c, wc):
                bad.setdefault(i, []).append((t, int((d != wd).sum()), int((c != wc).sum())))
        del slots
    print(f"S={S} mode={mode or 'default'} trials={trials}: deviations per sample:", bad if bad else "none")
