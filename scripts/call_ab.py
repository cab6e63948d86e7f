#!/usr/bin/env python
"""Time selected ops.* calls of one captured 1600x1200 forward against several builds of libpmn_hip.so, one process per build (a process
loads one library), same inputs (seeded), interleaved rounds; prints per call the microseconds per build and whether the outputs are
bit-identical to the first build's.

    python scripts/call_ab.py --ops feature_weight --libs build/fwn/libpmn_hip_lds.so,build/fwn/libpmn_hip_dpp5.so [--reps 40 --rounds 3]"""
import argparse
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--libs", required=True)
ap.add_argument("--ops", default="feature_weight")
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--height", type=int, default=1200)
ap.add_argument("--width", type=int, default=1600)
ap.add_argument("--views", type=int, default=5)
ap.add_argument("--child", default=None)
args = ap.parse_args()

if args.child is None:
    libs = args.libs.split(",")
    res = {}
    for r in range(args.rounds):
        for lib in libs:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--libs", lib, "--child", lib, "--ops", args.ops, "--reps", str(args.reps),
                                  "--height", str(args.height), "--width", str(args.width), "--views", str(args.views)],
                                 capture_output=True, text=True, cwd=ROOT)
            line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
            if not line:
                print(lib, "FAILED", out.stderr[-400:])
                continue
            for k, (us, dig) in json.loads(line[-1]).items():
                res.setdefault(k, {}).setdefault(lib, []).append((us, dig))
    for k in sorted(res, key=lambda x: int(x.split(":")[0])):
        first = None
        row = []
        for lib in libs:
            v = res[k].get(lib, [])
            if not v:
                continue
            first = first or v[0][1]
            row.append(f"{os.path.basename(lib).replace('libpmn_hip_', '').replace('.so', '')} {min(u for u, _ in v):.1f} us"
                       f"{'' if all(d == first for _, d in v) else ' (DIFFERENT BITS)'}")
        print(f"call {k:28s} " + " | ".join(row))
    sys.exit(0)

for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from patchmatchnet_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.abspath(args.child)
import torch  # noqa: E402

import bench  # noqa: E402
import patchmatchnet_amd as P  # noqa: E402
from patchmatchnet_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
model = P.PatchmatchNet(**bench.DEFAULT_KW)
bench.load_weights(model)
model = model.to(dev).eval()
s0 = bench.make_samples(1, args.views + 1, args.height, args.width, dev, 0)[0]
noise = torch.rand((1, 48, args.height // 8, args.width // 8), generator=torch.Generator().manual_seed(3)).to(dev)
names = args.ops.split(",")
orig = {n: getattr(ops, n) for n in names}
calls = []
cap = [False]


def wrap(name):
    def g(*a, **kw):
        out = orig[name](*a, **kw)
        if cap[0]:
            calls.append((name, a, {k: v for k, v in kw.items() if k != "out"}))
        return out
    return g


for n in names:
    setattr(ops, n, wrap(n))
with torch.no_grad():
    model([im for im in s0["images"]], s0["intrinsics"].clone(), s0["extrinsics"], s0["depth_min"], s0["depth_max"], noise=noise)
    cap[0] = True
    model([im for im in s0["images"]], s0["intrinsics"].clone(), s0["extrinsics"], s0["depth_min"], s0["depth_max"], noise=noise)
    cap[0] = False
    torch.cuda.synchronize()
    out = {}
    for k, (name, a, kw) in enumerate(calls):
        for _ in range(5):
            o = orig[name](*a, **kw)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(args.reps):
            o = orig[name](*a, **kw)
        t1.record()
        torch.cuda.synchronize()
        outs = o if isinstance(o, (tuple, list)) else (o,)
        h = hashlib.sha256()
        for t in outs:
            if isinstance(t, torch.Tensor) and t.numel():
                h.update(t.contiguous().cpu().numpy().tobytes())
        shape = next(tuple(t.shape) for t in outs if isinstance(t, torch.Tensor) and t.numel())
        out[f"{k}:{name}{list(shape)}"] = (t0.elapsed_time(t1) * 1e3 / args.reps, h.hexdigest()[:16])
print(json.dumps(out))
