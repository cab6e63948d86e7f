#!/usr/bin/env python
"""Same-box A/B of pmn_warp_correlate's two formulations on the REAL arguments of a forward.

Runs one PatchmatchNet.forward on bench.py's sample (photo-consistent scene, reference checkpoint), records the arguments of
every pmn_warp_correlate call (five per depth map at the default iterations), then replays each call under
the streaming kernel (gather_corr.hip) and the research build's matrix-core formulation (experimental/corr_mfma.hip, pmn_set_tuning
key 1 bit 6): HIP-event time per launch (median / min of --reps), the algorithmic bytes of SURVEY.md 8(d), and the difference
between the two outputs.  Needs the research build: `make -C patchmatchnet_amd/csrc EXPERIMENTAL=1` and PMN_EXPERIMENTAL=1.

    PMN_EXPERIMENTAL=1 python scripts/corr_ab.py [--width 1600 --height 1200 --views 5] [--reps 30] [--impls stream mfma]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(np.min(ts))


def window_stats(a):
    """Bounding boxes of the live taps of every (16-pixel tile, view, chunk of DCH hypotheses): what corr_mfma.hip calls a window."""
    ref, src, rel, depth = a[0], a[1], a[2], a[3]
    B, h, w, C = ref.shape
    N, _, hs, ws, _ = src.shape
    D = depth.shape[1]
    hw = h * w
    dev = ref.device
    ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32),
                            indexing="ij")
    ntile = (hw + 15) // 16
    pad = ntile * 16 - hw
    print(f"== C{C} D{D} {h}x{w} N{N}: {ntile} tiles", flush=True)
    for dch in sorted({8, 16, D}):
        if dch > D or D % dch:
            continue
        qs = []
        for v in range(N):
            P = rel[0, v]
            rx = (P[0, 0] * xs + P[0, 1] * ys + P[0, 2]) * ((ws - 1) / (w - 1))
            ry = (P[1, 0] * xs + P[1, 1] * ys + P[1, 2]) * ((hs - 1) / (h - 1))
            rz = P[2, 0] * xs + P[2, 1] * ys + P[2, 2]
            pz = rz[None] * depth[0] + P[2, 3]
            gx = (rx[None] * depth[0] + P[0, 3] * ((ws - 1) / (w - 1))) / pz
            gy = (ry[None] * depth[0] + P[1, 3] * ((hs - 1) / (h - 1))) / pz
            live = (pz > 1e-3) & (gx > -1) & (gx < ws) & (gy > -1) & (gy < hs)
            x0 = gx.floor().clamp(0, ws - 2)
            y0 = gy.floor().clamp(0, hs - 2)
            big = 1e9
            def tiles(t, fill):
                t = torch.where(live, t, torch.full_like(t, fill)).reshape(D, hw)
                t = torch.nn.functional.pad(t, (0, pad), value=fill)
                return t.reshape(D // dch, dch, ntile, 16)
            xmin = tiles(x0, big).amin(dim=(1, 3)); xmax = tiles(x0, -big).amax(dim=(1, 3))
            ymin = tiles(y0, big).amin(dim=(1, 3)); ymax = tiles(y0, -big).amax(dim=(1, 3))
            ok = xmax >= xmin
            q = ((xmax - xmin + 2) * (ymax - ymin + 2))[ok]
            qs.append(q)
        q = torch.cat(qs).double()
        pct = lambda p: float(torch.quantile(q[torch.randperm(q.numel(), device=dev)[:1000000]], p))
        fr = lambda c: float((q > c).double().mean())
        print(f"   DCH {dch:2d}: Q mean {float(q.mean()):7.1f} p50 {pct(0.5):5.0f} p90 {pct(0.9):5.0f} p99 {pct(0.99):6.0f} max {float(q.max()):8.0f}"
              f" | N-tiles/window {float(((q + 15) // 16).mean()):5.2f} | frac > 64/96/128/192/256: {fr(64):.3f} {fr(96):.3f} "
              f"{fr(128):.3f} {fr(192):.3f} {fr(256):.3f} | windows {q.numel()}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--height", type=int, default=1200)
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--impls", nargs="+", default=["stream", "mfma"])
    ap.add_argument("--json", default=None)
    ap.add_argument("--windows", action="store_true", help="window statistics of the 16-pixel tiles instead of timings")
    args = ap.parse_args()
    import bench
    import patchmatchnet_amd as P
    from patchmatchnet_amd import ops
    dev = "cuda:0"
    model = P.PatchmatchNet(**bench.DEFAULT_KW)
    bench.load_weights(model)
    model = model.to(dev).eval()
    s = bench.make_samples(1, args.views + 1, args.height, args.width, dev, 0)[0]

    calls = []
    real = ops.warp_correlate

    def recorder(*a, **kw):
        calls.append((a, kw))
        return real(*a, **kw)

    def select(impl):
        ops.set_tuning(ops.TUNE_FLAGS, ops.FLAG_MFMA if impl == "mfma" else 0)

    ops.warp_correlate = recorder
    select("stream")
    torch.manual_seed(0)
    with torch.no_grad():
        model(s["images"], s["intrinsics"], s["extrinsics"], s["depth_min"], s["depth_max"])
    torch.cuda.synchronize()
    ops.warp_correlate = real
    print(f"{len(calls)} pmn_warp_correlate calls recorded", flush=True)

    if args.windows:
        for a, kw in calls:
            window_stats(a)
        return
    rows, tot = [], {i: 0.0 for i in args.impls}
    tot_bytes = 0
    for a, kw in calls:
        ref, src, rel, depth, vw = a[0], a[1], a[2], a[3], a[4]
        B, h, w, C = ref.shape
        N, D, G = src.shape[0], depth.shape[1], a[8]
        nbytes = 4 * B * h * w * ((1 + N) * C + D + N + G * D)
        tot_bytes += nbytes
        name = f"C{C}_D{D}_{h}x{w}_N{N}_{'vw' if vw is not None else 'pixelwise'}"
        outs = {}
        row = dict(shape=name, bytes=nbytes)
        kw2 = dict(kw)
        kw2["want_similarity"] = True
        for impl in args.impls:
            select(impl)
            med, mn = timed(lambda: real(*a, **kw), args.reps)
            cost, vwo, argmax, sim = real(*a, **kw2)
            torch.cuda.synchronize()
            outs[impl] = (cost.clone(), sim.clone(), None if vw is not None else vwo.clone())
            row[impl] = dict(median_us=med * 1e3, min_us=mn * 1e3, gbps=nbytes / med / 1e6)
            tot[impl] += med
        if len(args.impls) > 1:
            x, y = outs[args.impls[0]], outs[args.impls[1]]
            row["max_abs_diff_sim"] = float((x[1] - y[1]).abs().max())
            row["max_abs_diff_cost"] = float((x[0] - y[0]).abs().max())
            if x[2] is not None:
                row["max_abs_diff_vw"] = float((x[2] - y[2]).abs().max())
        rows.append(row)
        print(json.dumps(row), flush=True)
    summary = {i: dict(ms_per_depth_map=tot[i], gbps=tot_bytes / tot[i] / 1e6, frac_of_8TBs=tot_bytes / tot[i] / 1e6 / 8000)
               for i in args.impls}
    print(json.dumps(dict(summary=summary, algorithmic_MB=tot_bytes / 1e6)), flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(dict(rows=rows, summary=summary, algorithmic_MB=tot_bytes / 1e6), f, indent=1)


if __name__ == "__main__":
    main()
