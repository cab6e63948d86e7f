#!/usr/bin/env python
"""Victim x disturber matrix at kernel granularity (DESIGN_LESSONS.md lesson 46): every captured ops.* call of one forward can be the
disturber (re-issued in a loop on stream B) while a victim call is re-issued on stream A and compared with its solo output.

    python scripts/overlap_pairs.py [--victims feature_weight,warp_correlate] [--lib ...]"""
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
ap.add_argument("--reps", type=int, default=16)
ap.add_argument("--victims", default="20,22,28,36,25", help="indices of the captured calls to use as victims")
ap.add_argument("--disturbers", default="", help="comma list of disturber name prefixes (default: all)")
ap.add_argument("--loops", type=int, default=40, help="disturber launches queued before the victim's reps")
args = ap.parse_args()
from patchmatchnet_amd import _lib  # noqa: E402
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
import torch  # noqa: E402

import bench  # noqa: E402
import patchmatchnet_amd as P  # noqa: E402
from patchmatchnet_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
model = P.PatchmatchNet(**bench.DEFAULT_KW)
bench.load_weights(model)
model = model.to(dev).eval()
s0, s1 = bench.make_samples(2, args.views + 1, args.height, args.width, dev, 0)
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
        if capturing[0]:
            outs = out if isinstance(out, (tuple, list)) else (out,)
            CALLS.append((name, a, kw, [o.clone() if isinstance(o, torch.Tensor) and o.numel() else None for o in outs]))
        return out
    return g


for n in NAMES:
    setattr(ops, n, wrap(n))
A, B = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


PLANES = int(os.environ.get("PMN_PLANES", "0"))  # compare only the first PLANES channels of every output (probe builds)


def same(got, want):
    got = got if isinstance(got, (tuple, list)) else (got,)
    if PLANES:
        return all(w is None or torch.equal(g[:, :PLANES], w[:, :PLANES]) for g, w in zip(got, want))
    return all(w is None or torch.equal(g, w) for g, w in zip(got, want))


with torch.no_grad():
    with torch.cuda.stream(A):
        model([im for im in s0["images"]], s0["intrinsics"].clone(), s0["extrinsics"], s0["depth_min"], s0["depth_max"], noise=noise)
        capturing[0] = True
        model([im for im in s0["images"]], s0["intrinsics"].clone(), s0["extrinsics"], s0["depth_min"], s0["depth_max"], noise=noise)
        capturing[0] = False
    torch.cuda.synchronize()
    print(f"lib = {_lib.LIB_PATH}")
    print("calls:", [(k, c[0]) for k, c in enumerate(CALLS)])
    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(n):
            fn()
        t1.record()
        torch.cuda.synchronize()
        return t0.elapsed_time(t1) / n

    # disturbers: every captured call, plus kernels that are not this library's
    half = torch.rand(4096, 4096, device=dev).half()
    bf = half.bfloat16()
    f32 = half.float()
    big = torch.rand(64 * 1024 * 1024, device=dev)
    extra = [("torch.matmul fp16 4096^3", lambda: torch.matmul(half, half)), ("torch.matmul bf16 4096^3", lambda: torch.matmul(bf, bf)),
             ("torch.matmul fp32 4096^3", lambda: torch.matmul(f32, f32)), ("torch.add 256 MB", lambda: torch.add(big, 1.0)),
             ("torch.exp 256 MB", lambda: torch.exp(big)), ("torch conv2d fp16 (MIOpen)", None)]
    xh = torch.rand(6, 64, 150, 200, device=dev).half()
    wh = torch.rand(64, 64, 3, 3, device=dev).half()
    extra[-1] = ("torch conv2d fp16 (MIOpen)", lambda: torch.nn.functional.conv2d(xh, wh, padding=1))
    # synthetic one-feature disturbers (scripts/repro/disturbers.hip)
    dl_path = os.path.join(ROOT, "scripts", "repro", "libdisturb.so")
    if os.path.isfile(dl_path):
        import ctypes
        DL = ctypes.CDLL(dl_path)
        DL.disturb_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        dbuf = torch.zeros(32 * 1024 * 1024, device=dev)

        def micro(which, iters, lds=0, blocks=2048):
            def f():
                rc = DL.disturb_launch(which, dbuf.data_ptr(), dbuf.numel(), blocks, iters, lds, torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc
            return f
        extra += [("micro: fp16 MFMA 16x16x32 only", micro(0, 4000)), ("micro: fp32 MFMA 16x16x4 only", micro(1, 4000)),
                  ("micro: cvt f32<->f16 VALU only", micro(2, 4000)), ("micro: LDS 32 KB write_b64/read_b128 + barriers", micro(3, 40, 32768)),
                  ("micro: LDS 8 KB write_b64/read_b128 + barriers", micro(3, 160, 8192)),
                  ("micro: LDS 32 KB -> fp16 MFMA", micro(4, 4000, 32768)), ("micro: global float4 copy", micro(5, 0))]
    disturbers = [(f"{di}:{c[0]}", (lambda c=c: ORIG[c[0]](*c[1], **c[2]))) for di, c in enumerate(CALLS)] + extra
    wanted = [x for x in args.disturbers.split(",") if x]
    if wanted:
        disturbers = [d for d in disturbers if any(d[0].startswith(w) for w in wanted)]
    victims = [int(x) for x in args.victims.split(",")]
    for vi in victims:
        vname, va, vkw, vwant = CALLS[vi]
        vkw = {k: v for k, v in vkw.items() if k != "out"}
        with torch.cuda.stream(A):
            tv = timed(lambda: ORIG[vname](*va, **vkw), 4)
        hits = []
        for dn, dfn in disturbers:
            with torch.cuda.stream(B):
                td = timed(dfn, 3)
            loops = int(min(max(3.0 * tv * args.reps / max(td, 1e-3), 8), 4000))
            torch.cuda.synchronize()
            with torch.cuda.stream(B):
                for _ in range(loops):
                    dfn()
            outs = []
            with torch.cuda.stream(A):
                for _ in range(args.reps):
                    outs.append(ORIG[vname](*va, **vkw))
                A.synchronize()
            busy = not B.query()
            torch.cuda.synchronize()
            bad = sum(0 if same(o, vwant) else 1 for o in outs)
            del outs
            hits.append((dn, bad, busy, td))
        print(f"victim call {vi} {vname} ({tv * 1e3:.0f} us): MISMATCHES under: " +
              ", ".join(f"{dn}={bad}/{args.reps}{'' if busy else '(B idle at end)'}" for dn, bad, busy, td in hits if bad) +
              " | clean, B busy throughout: " + ", ".join(dn for dn, bad, busy, td in hits if not bad and busy) +
              " | clean but B finished early: " + ", ".join(dn for dn, bad, busy, td in hits if not bad and not busy), flush=True)

    # ---- anatomy of one corrupted output: where, and what the wrong values are --------------------------------------------------
    if os.environ.get("PMN_ANATOMY"):
        vi, dname = [int(x) if i == 0 else x for i, x in enumerate(os.environ["PMN_ANATOMY"].split(":", 1))]
        vname, va, vkw, vwant = CALLS[vi]
        dfn = next(f for n_, f in disturbers if n_.startswith(dname))
        for trial in range(3):
            torch.cuda.synchronize()
            with torch.cuda.stream(B):
                for _ in range(200):
                    dfn()
            with torch.cuda.stream(A):
                outs = [ORIG[vname](*va, **vkw) for _ in range(4)]
            torch.cuda.synchronize()
            for o in outs:
                o = o if isinstance(o, (tuple, list)) else (o,)
                if len(o) >= 2 and vwant[1] is not None and o[0].dim() == 4 and o[1].dim() == 4 and not torch.equal(o[0], vwant[0]):
                    # warp_correlate: pixels whose cost differs vs pixels where some view weight differs
                    pc = (o[0] != vwant[0]).any(1)[0]
                    pv = (o[1] != vwant[1]).any(1)[0]
                    print(f"trial {trial}: pixels with a different cost {int(pc.sum())}, with a different view weight {int(pv.sum())}, "
                          f"cost differs but no view weight does {int((pc & ~pv).sum())}, per-view counts "
                          f"{[int((o[1][0, v] != vwant[1][0, v]).sum()) for v in range(o[1].shape[1])]}")
                    dv = (o[1] - vwant[1])[o[1] != vwant[1]]
                    print(f"trial {trial}: view-weight differences: {int((dv < 0).sum())} lower, {int((dv > 0).sum())} higher; "
                          f"first few got/want {[(round(float(a), 6), round(float(b), 6)) for a, b in zip(o[1][o[1] != vwant[1]][:8], vwant[1][o[1] != vwant[1]][:8])]}")
                for j, (g, w_) in enumerate(zip(o, vwant)):
                    if w_ is None or torch.equal(g, w_):
                        continue
                    bad = (g != w_).nonzero()
                    print(f"trial {trial} out {j} shape {tuple(g.shape)}: {bad.shape[0]} elements differ")
                    flat_want = w_.reshape(-1)
                    for row in bad[:24].tolist():
                        gv, wv = float(g[tuple(row)]), float(w_[tuple(row)])
                        elsewhere = (flat_want == gv).nonzero().reshape(-1)[:4].tolist()
                        print(f"    at {row}: got {gv:.9g} want {wv:.9g}; got-value occurs in the solo output at flat {elsewhere} "
                              f"(this element's flat index {int(sum(r * s for r, s in zip(row, w_.stride())))})")
                    break
                else:
                    continue
                break
