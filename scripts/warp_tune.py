#!/usr/bin/env python
"""Replay the five pmn_warp_correlate launches of a REAL forward (bench.py's synthetic sample through the whole cascade, so the
hypotheses / view weights are what the kernel sees in the benchmark) under different kernel families and window sizes.

    python scripts/warp_tune.py [--reps 20] [--configs stream win12 win8 win16 ...]

Prints one row per (launch, configuration): median / min microseconds (HIP events on the launch stream), algorithmic GB/s
(SURVEY.md 8(d) bytes) and whether the outputs are bit-identical to the streaming family's.
"""
import argparse
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


def parse_config(name, ops):
    """stream | tile<KB> | win<KB>[p<KB>][n][@<ablation bits>] (first windowed form; n = no quad rotation) | lane<KB>[w<2|3>][@<bits>]
    e.g. win12, win16p6, win12n, win12@3, lane12, lane16w2, lane12@3"""
    if name == "stream":
        return dict(flags=0, cap=12288, cap_pix=8192, dbg=0, lane=False, wps=3)
    if name.startswith("tile"):  # tile<KB>: the tile-window kernel (known-weights launches), window buffer of <KB> KB
        return dict(flags=ops.FLAG_TILE, cap=12288, cap_pix=8192, dbg=0, lane=False, wps=3, tile_cap=int(name[4:] or 20) * 1024)
    if name.startswith("lane"):
        body = name[4:]
        dbg = 0
        if "@" in body:
            body, d = body.split("@")
            dbg = int(d)
        wps = 3
        if "w" in body:
            body, ww = body.split("w")
            wps = int(ww)
        return dict(flags=ops.FLAG_WINDOWED, cap=int(body) * 1024, cap_pix=8192, dbg=dbg, lane=True, wps=wps)
    assert name.startswith("win"), name
    body = name[3:]
    dbg = 0
    if "@" in body:
        body, d = body.split("@")
        dbg = int(d)
    norot = body.endswith("n")
    if norot:
        body = body[:-1]
    if "p" in body:
        c, cp = body.split("p")
    else:
        c, cp = body, "8"
    return dict(flags=ops.FLAG_WINDOWED | ops.FLAG_WIN_V1 | (ops.FLAG_NO_ROTATION if norot else 0), cap=int(c) * 1024,
                cap_pix=int(cp) * 1024, dbg=dbg, lane=False, wps=3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--height", type=int, default=1200)
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--configs", nargs="+", default=["stream", "win12", "win8", "win16", "win12n", "win12p4", "win12p6"])
    args = ap.parse_args()
    import bench
    import patchmatchnet_amd as P
    from patchmatchnet_amd import ops
    dev = torch.device("cuda", 0)
    model = P.PatchmatchNet(**bench.DEFAULT_KW)
    bench.load_weights(model)
    model = model.to(dev).eval()
    s = bench.make_samples(1, args.views + 1, args.height, args.width, dev, 0)[0]

    calls = []
    real = ops.warp_correlate

    def recorder(*a, **k):
        calls.append((a, k))
        return real(*a, **k)

    ops.warp_correlate = recorder
    with torch.no_grad():
        torch.manual_seed(1234)
        model([im for im in s["images"]], s["intrinsics"].clone(), s["extrinsics"], s["depth_min"], s["depth_max"])
    ops.warp_correlate = real
    torch.cuda.synchronize()
    print(f"{len(calls)} pmn_warp_correlate launches recorded")

    def apply(cfg):
        ops.set_tuning(ops.TUNE_FLAGS, cfg["flags"])
        if "tile_cap" in cfg:
            ops.set_tuning(ops.TUNE_TILE_WINDOW_BYTES, cfg["tile_cap"])
        if cfg["lane"]:
            ops.set_tuning(ops.TUNE_LANE_WINDOW_BYTES, cfg["cap"])
            ops.set_tuning(ops.TUNE_LANE_ABLATE, cfg["dbg"])
            ops.set_tuning(ops.TUNE_LANE_WAVES_PER_SIMD, cfg["wps"])
        else:
            ops.set_tuning(ops.TUNE_WINDOW_BYTES, cfg["cap"])
            ops.set_tuning(ops.TUNE_WINDOW_BYTES_PIXELWISE, cfg["cap_pix"])
            ops.set_tuning(ops.TUNE_ABLATE, cfg["dbg"])

    totals = {}
    for ci, (a, k) in enumerate(calls):
        ref, src, rel, depth, vw = a[0], a[1], a[2], a[3], a[4]
        B, h, w, C = ref.shape
        N, D = src.shape[0], depth.shape[1]
        G = a[8]
        nbytes = 4 * B * h * w * ((1 + N) * C + D + N + G * D)
        apply(parse_config("stream", ops))
        want = real(*a, **k)
        torch.cuda.synchronize()
        want = [t.clone() for t in want if isinstance(t, torch.Tensor)]
        for name in args.configs:
            cfg = parse_config(name, ops)
            apply(cfg)
            got = real(*a, **k)
            torch.cuda.synchronize()
            got = [t for t in got if isinstance(t, torch.Tensor)]
            same = all(torch.equal(x, y) for x, y in zip(want, got))
            med, mn = timed(lambda: real(*a, **k), args.reps)
            totals[name] = totals.get(name, 0.0) + med
            print(f"launch {ci}: C{C} D{D} {h}x{w} N{N} {'vw' if vw is not None else 'pixelwise':9s} {name:10s} "
                  f"median {med * 1e3:8.1f} us  min {mn * 1e3:8.1f} us  {nbytes / med / 1e6:8.1f} GB/s(alg)  "
                  f"{'bit-identical' if same else 'DIFFERENT'}", flush=True)
    ops.set_tuning(ops.TUNE_FLAGS, ops.DEFAULT_FLAGS)
    total_bytes = 0
    for a, k in calls:
        B, h, w, C = a[0].shape
        total_bytes += 4 * B * h * w * ((1 + a[1].shape[0]) * C + a[3].shape[1] + a[1].shape[0] + a[8] * a[3].shape[1])
    for name, ms in totals.items():
        print(f"total {name:10s} {ms:.3f} ms per depth map  {total_bytes / ms / 1e6:8.1f} GB/s = "
              f"{total_bytes / ms / 1e6 / 8000 * 100:.2f} % of 8 TB/s")


if __name__ == "__main__":
    main()
