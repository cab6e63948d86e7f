#!/usr/bin/env python
"""Per-call sensitivity table for DESIGN_LESSONS.md lesson 46: every ops.* call of ONE forward is captured with its arguments, then
re-issued on stream A (same inputs, fresh outputs) alone and while stream B keeps the device busy with a disturber; the table says
for which calls the output stops being the solo output, how often, and under which disturber.

    python scripts/overlap_victims.py [--lib other/libpmn_hip.so] [--height 1200 --width 1600 --views 5 --reps 30]
disturbers: forward (launch-plan replays of another sample), stream (ATen adds over 256 MB: pure HBM traffic), matmul (ATen GEMMs),
featurenet (this library's FeatureNet only)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--height", type=int, default=1200)
ap.add_argument("--width", type=int, default=1600)
ap.add_argument("--views", type=int, default=5)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--disturbers", default="forward,stream,matmul,featurenet")
ap.add_argument("--only", default="", help="comma list of op names to test (default: all)")
args = ap.parse_args()
from patchmatchnet_amd import _lib  # noqa: E402
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
import torch  # noqa: E402

import bench  # noqa: E402
import patchmatchnet_amd as P  # noqa: E402
from patchmatchnet_amd import ops  # noqa: E402
from patchmatchnet_amd.graph import PlannedForward  # noqa: E402

dev = torch.device("cuda", 0)
model = P.PatchmatchNet(**bench.DEFAULT_KW)
bench.load_weights(model)
model = model.to(dev).eval()
samples = bench.make_samples(2, args.views + 1, args.height, args.width, dev, 0)
noise = torch.rand((1, 48, args.height // 8, args.width // 8), device=dev)
NAMES = ["stem_f16s", "conv2d_f16s", "conv2d_f16s_pair", "pointwise_split_mfma", "fpn_level", "stage_projections", "offset_heads_f16s", "feature_weight",
         "init_hypotheses", "warp_correlate", "aggregate_regress", "normalize_depth", "conv2d", "refine_fused", "confidence"]
ORIG = {n: getattr(ops, n) for n in NAMES}
CALLS = []
capturing = [False]


def wrap(name):
    f = ORIG[name]

    def g(*a, **kw):
        out = f(*a, **kw)
        if capturing[0] and "out" not in kw:
            outs = out if isinstance(out, (tuple, list)) else (out,)
            CALLS.append((name, a, kw, [o.clone() if isinstance(o, torch.Tensor) and o.numel() else None for o in outs]))
        return out
    return g


for n in NAMES:
    setattr(ops, n, wrap(n))
s0, s1 = samples
A, B = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
slotB = PlannedForward(model, inputs_in_place=True)
big = torch.rand(64 * 1024 * 1024, device=dev)
big2 = torch.empty_like(big)
m1 = torch.rand(4096, 4096, device=dev)
m2 = torch.empty_like(m1)


def disturb(kind, n):
    with torch.cuda.stream(B):
        for _ in range(n):
            if kind == "forward":
                slotB([im for im in s1["images"]], s1["intrinsics"].clone(), s1["extrinsics"], s1["depth_min"], s1["depth_max"])
            elif kind == "stream":
                for _ in range(12):
                    torch.add(big, 1.0, out=big2)
            elif kind == "matmul":
                for _ in range(4):
                    torch.matmul(m1, m1, out=m2)
            elif kind == "featurenet":
                for _ in range(2):
                    FN(s1["images"])


def same(got, want):
    got = got if isinstance(got, (tuple, list)) else (got,)
    return all(w is None or torch.equal(g, w) for g, w in zip(got, want))


with torch.no_grad():
    FN = model.feature.forward_hip
    with torch.cuda.stream(B):
        slotB([im for im in s1["images"]], s1["intrinsics"].clone(), s1["extrinsics"], s1["depth_min"], s1["depth_max"])
    torch.cuda.synchronize()
    with torch.cuda.stream(A):
        model([im for im in s0["images"]], s0["intrinsics"].clone(), s0["extrinsics"], s0["depth_min"], s0["depth_max"], noise=noise)
        capturing[0] = True
        model([im for im in s0["images"]], s0["intrinsics"].clone(), s0["extrinsics"], s0["depth_min"], s0["depth_max"], noise=noise)
        capturing[0] = False
    torch.cuda.synchronize()
    only = set(x for x in args.only.split(",") if x)
    kinds = [k for k in args.disturbers.split(",") if k]
    print(f"lib = {_lib.LIB_PATH}")
    print(f"{len(CALLS)} captured calls; reps = {args.reps}; columns: solo | " + " | ".join(kinds))
    for k, (name, a, kw, want) in enumerate(CALLS):
        if only and name not in only:
            continue
        shape = next((tuple(w.shape) for w in want if w is not None), None)
        row = []
        with torch.cuda.stream(A):
            bad = sum(0 if same(ORIG[name](*a, **kw), want) else 1 for _ in range(args.reps))
        torch.cuda.synchronize()
        row.append(bad)
        for kind in kinds:
            torch.cuda.synchronize()
            disturb(kind, 12)
            outs = []
            with torch.cuda.stream(A):
                for _ in range(args.reps):
                    outs.append(ORIG[name](*a, **kw))
            busy = not B.query()
            torch.cuda.synchronize()
            bad = sum(0 if same(o, want) else 1 for o in outs)
            row.append(f"{bad}{'' if busy else '(idle)'}")
            del outs
        print(f"call {k:2d} {name:22s} {str(shape):24s} differ of {args.reps}: " + " | ".join(str(x) for x in row), flush=True)
