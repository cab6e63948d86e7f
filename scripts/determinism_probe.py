#!/usr/bin/env python
"""Is a forward bit-reproducible run after run (same inputs, same stage-3 draw, one process)?  Repeats an eager forward with the debug
hooks N times at two sizes and reports, per recorded tensor, how many repetitions differ from the first.
    python scripts/determinism_probe.py [--lib other.so] [--reps 30]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--reps", type=int, default=30)
a = ap.parse_args()
import torch  # noqa: E402
if a.lib:
    from patchmatchnet_amd import _lib
    _lib.LIB_PATH = os.path.abspath(a.lib)
import bench  # noqa: E402
import patchmatchnet_amd as P  # noqa: E402

dev = torch.device("cuda", 0)
model = P.PatchmatchNet(**bench.DEFAULT_KW)
bench.load_weights(model)
model = model.to(dev).eval()
for (H, W, nv) in ((96, 128, 3), (1200, 1600, 6)):
    s = bench.make_samples(1, nv, H, W, dev, 0)[0]
    noise = torch.rand(1, 48, H // 8, W // 8, generator=torch.Generator().manual_seed(7)).to(dev)
    first, diffs = None, {}
    side = torch.cuda.Stream(dev)
    for r in range(a.reps):
        dbg = {}
        with torch.no_grad():
            if r % 2:  # every other repetition with unrelated work in flight on another stream (what eval.py's pipeline looks like)
                with torch.cuda.stream(side):
                    junk = torch.rand(64, 1024, 1024, device=dev).sum()
            depth, conf, dpm = model([im for im in s["images"]], s["intrinsics"].clone(), s["extrinsics"], s["depth_min"], s["depth_max"],
                                     noise=noise, debug=dbg)
        torch.cuda.synchronize()
        rec = {"depth": depth, "confidence": conf}
        for st in (3, 2, 1):
            for it, x in enumerate(dbg[st]):
                for k in ("depth", "view_weights", "score", "depth_sample"):
                    if k in x and x[k] is not None and torch.is_tensor(x[k]):
                        rec[f"s{st}_it{it + 1}_{k}"] = x[k]
        rec = {k: v.clone() for k, v in rec.items()}
        if first is None:
            first = rec
        else:
            for k, v in rec.items():
                if not torch.equal(v, first[k]):
                    diffs.setdefault(k, []).append((r, int((v != first[k]).sum())))
    print(f"{W}x{H} lib={a.lib or 'tree'} reps={a.reps}:", "bit-reproducible" if not diffs else {k: v[:4] for k, v in diffs.items()})
