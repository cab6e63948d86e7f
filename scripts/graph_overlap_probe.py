#!/usr/bin/env python
"""Two (or three) ALREADY CAPTURED forwards replayed concurrently on their own streams, many times, fixed inputs and noise: does every
replay reproduce the eager forward bit for bit?  Separates "a capture beside a replay" from "two replays beside each other".
    PMN_PROBE_ROOT=<tree> python scripts/graph_overlap_probe.py [rounds] [features]"""
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

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
inject = len(sys.argv) > 2 and "features" in sys.argv[2:]
eager = len(sys.argv) > 2 and "eager" in sys.argv[2:]
bare = len(sys.argv) > 2 and "bare" in sys.argv[2:]  # replay the captured graphs directly: no fill / draw / clone between launches
dev = torch.device("cuda", 0)
_, params, kw = GU.load_case("default")
model = P.PatchmatchNet(**kw)
model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
model = model.to(dev).eval()
H, W, nv, S = (int(os.environ.get("PMN_PROBE_H", 96)), int(os.environ.get("PMN_PROBE_W", 128)), int(os.environ.get("PMN_PROBE_NV", 3)), 3)
intr, extr = synth.synthetic_cameras(nv, H, W)
samples = []
for s in range(S):
    g = torch.Generator().manual_seed(100 + s)
    samples.append(dict(images=[torch.rand(1, 3, H, W, generator=g).to(dev) for _ in range(nv)], intrinsics=torch.from_numpy(intr).to(dev),
                        extrinsics=torch.from_numpy(extr).to(dev), dmin=torch.tensor([425.0], device=dev), dmax=torch.tensor([935.0], device=dev)))
want, feats = [], []
with torch.no_grad():
    for s in samples:
        f = model.extract_features(list(s["images"])) if inject else None
        feats.append(f)
        torch.manual_seed(3)  # every forward of this probe draws the same stage-3 noise
        d, c, _ = model(list(s["images"]), s["intrinsics"].clone(), s["extrinsics"], s["dmin"], s["dmax"], features=f)
        want.append((d.clone(), c.clone()))
torch.cuda.synchronize()
slots = [GraphedForward(model) for _ in range(S)]
streams = [torch.cuda.Stream(dev) for _ in range(S)]
with torch.no_grad():
    for k in range(S):  # capture one after the other, device idle in between
        torch.cuda.synchronize()
        torch.manual_seed(3)
        with torch.cuda.stream(streams[k]):
            s = samples[k]
            slots[k](list(s["images"]), s["intrinsics"], s["extrinsics"], s["dmin"], s["dmax"], features=feats[k])
        torch.cuda.synchronize()
    bad = {}
    if bare:
        held = [slots[k].cache[next(iter(slots[k].cache))] for k in range(S)]  # (graph, static, (depth, confidence))
        for r in range(rounds):
            for k in range(S):
                with torch.cuda.stream(streams[k]):
                    held[k][0].replay()
            if r % 8 == 7:  # look at the outputs now and then (after a full synchronisation)
                torch.cuda.synchronize()
                for k in range(S):
                    d, c = held[k][2]
                    if not torch.equal(d, want[k][0]) or not torch.equal(c, want[k][1]):
                        bad.setdefault(k, []).append((r, int((d != want[k][0]).sum())))
        rounds_done = rounds
        print(f"{S} captured forwards replayed BARE (no eager op between launches) x{rounds}:", {k: (len(v), v[:3]) for k, v in bad.items()} if bad else "every checked output equals the eager forward")
        sys.exit(0)
    for r in range(rounds):
        outs = []
        for k in range(S):
            torch.manual_seed(3)
            with torch.cuda.stream(streams[k]):
                s = samples[k]
                if eager:
                    d, c, _ = model(list(s["images"]), s["intrinsics"].clone(), s["extrinsics"], s["dmin"], s["dmax"], features=feats[k])
                else:
                    d, c = slots[k](list(s["images"]), s["intrinsics"], s["extrinsics"], s["dmin"], s["dmax"], features=feats[k])
                outs.append((d.clone(), c.clone()))
        torch.cuda.synchronize()
        for k, ((d, c), (wd, wc)) in enumerate(zip(outs, want)):
            if not torch.equal(d, wd) or not torch.equal(c, wc):
                bad.setdefault(k, []).append((r, int((d != wd).sum())))
print(f"{S} {'eager' if eager else 'captured'} forwards concurrently x{rounds} (features injected: {inject}):", {k: (len(v), v[:3]) for k, v in bad.items()} if bad else "every replay equals the eager forward")
