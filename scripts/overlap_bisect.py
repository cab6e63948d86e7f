#!/usr/bin/env python
"""Which kernel's output changes when another forward runs beside it?  (DESIGN_LESSONS.md lessons 45-46.)

Round 6's first GPU session showed that plain launches replayed from C on three streams differ from the eager forward just like
round 5's HIP-graph replays did (89 of 96 steps at 1600x1200): it is not the graphs, it is two forwards of this library sharing the
device.  This probe runs ONE forward eagerly on stream A with every ops.* call's outputs cloned, once alone (twice: determinism) and
then with a disturber -- launch-plan replays of another sample queued on stream B, which keep the device busy for the whole pass --
and reports, call by call, the first outputs that differ.

    python scripts/overlap_bisect.py [--height 1200 --width 1600 --views 5 --trials 6 --disturb 40]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
import patchmatchnet_amd as P  # noqa: E402
from patchmatchnet_amd import ops  # noqa: E402
from patchmatchnet_amd.graph import PlannedForward  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--height", type=int, default=1200)
ap.add_argument("--width", type=int, default=1600)
ap.add_argument("--views", type=int, default=5)
ap.add_argument("--trials", type=int, default=6)
ap.add_argument("--disturb", type=int, default=40, help="plan replays queued on the second stream before the recorded pass")
ap.add_argument("--disturber", choices=("forward", "featurenet", "cascade"), default="forward")
args = ap.parse_args()

dev = torch.device("cuda", 0)
model = P.PatchmatchNet(**bench.DEFAULT_KW)
bench.load_weights(model)
model = model.to(dev).eval()
samples = bench.make_samples(2, args.views + 1, args.height, args.width, dev, 0)
noise = torch.rand((1, 48, args.height // 8, args.width // 8), device=dev)

NAMES = ["stem_f16s_views", "stem_f16s", "stem", "conv2d_f16s", "conv2d_f16s_pair", "pointwise_split_mfma", "fpn_level", "stage_projections", "offset_heads_f16s",
         "feature_weight", "init_hypotheses", "warp_correlate", "aggregate_regress", "normalize_depth", "conv2d", "refine_fused",
         "confidence", "nchw_to_nhwc"]
ORIG = {n: getattr(ops, n) for n in NAMES}
REC = None


def wrap(name):
    f = ORIG[name]

    def g(*a, **kw):
        out = f(*a, **kw)
        if REC is not None:
            outs = out if isinstance(out, (tuple, list)) else (out,)
            REC.append((name, [o.clone() for o in outs if isinstance(o, torch.Tensor) and o.numel() > 0]))
        return out
    return g


for n in NAMES:
    setattr(ops, n, wrap(n))
import patchmatchnet_amd.module as M  # noqa: E402  (module.py calls ops.<fn> through the same module object)


def forward(s):
    return model([im for im in s["images"]], s["intrinsics"].clone(), s["extrinsics"], s["depth_min"], s["depth_max"], noise=noise)


def recorded(s):
    global REC
    REC = []
    d, c, _ = forward(s)
    rec, REC = REC, None
    rec.append(("OUTPUT", [d.clone(), c.clone()]))
    return rec


A, B = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
slotB = PlannedForward(model, inputs_in_place=True)
with torch.no_grad():
    with torch.cuda.stream(B):
        slotB([im for im in samples[1]["images"]], samples[1]["intrinsics"].clone(), samples[1]["extrinsics"], samples[1]["depth_min"],
              samples[1]["depth_max"])
    torch.cuda.synchronize()
    with torch.cuda.stream(A):
        forward(samples[0])
        base = recorded(samples[0])
        again = recorded(samples[0])
    torch.cuda.synchronize()
    print(f"{len(base)} recorded calls per forward")
    same = all(all(torch.equal(x, y) for x, y in zip(a[1], b[1])) for a, b in zip(base, again))
    print("alone, twice: identical" if same else "alone, twice: DIFFERENT (not deterministic even without a disturber)")
    del again

    def disturb(n):
        with torch.cuda.stream(B):
            if args.disturber == "forward":
                for _ in range(n):
                    slotB([im for im in samples[1]["images"]], samples[1]["intrinsics"].clone(), samples[1]["extrinsics"],
                          samples[1]["depth_min"], samples[1]["depth_max"])
            elif args.disturber == "featurenet":
                for _ in range(3 * n):
                    ORIG_FN(samples[1]["images"])
            else:
                raise SystemExit("cascade disturber: not implemented")

    ORIG_FN = model.feature.forward_hip
    first_counts = {}
    for t in range(args.trials):
        torch.cuda.synchronize()
        disturb(args.disturb)
        with torch.cuda.stream(A):
            rec = recorded(samples[0])
        still_busy = not B.query()
        torch.cuda.synchronize()
        diffs = []
        for k, (a, b) in enumerate(zip(base, rec)):
            assert a[0] == b[0]
            for j, (x, y) in enumerate(zip(a[1], b[1])):
                if not torch.equal(x, y):
                    xf, yf = x.float(), y.float()
                    bad = (xf != yf) & ~(torch.isnan(xf) & torch.isnan(yf))
                    n_bad = int(bad.sum())
                    rel = float(((xf - yf).abs() / xf.abs().clamp_min(1e-12))[bad].max()) if n_bad else 0.0
                    idx = bad.reshape(-1).nonzero()[:3].reshape(-1).tolist()
                    diffs.append((k, a[0], j, tuple(x.shape), n_bad, rel, idx))
        print(f"trial {t}: disturber still running at the end of the pass: {still_busy}; calls whose outputs differ: {len(diffs)}")
        for d in diffs[:6]:
            print("   call %d %s out %d %s: %d elements differ, max rel %.3e, first flat indices %s" % d)
        if diffs:
            first_counts[(diffs[0][0], diffs[0][1])] = first_counts.get((diffs[0][0], diffs[0][1]), 0) + 1
    print("first differing call over the trials:", first_counts)
    print("call list:", [(k, n) for k, (n, _) in enumerate(base)])
