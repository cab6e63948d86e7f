#!/usr/bin/env python
"""Design study (CPU, no GPU): how big is the source-map footprint of a pixel tile in the warp+correlation kernel?

Runs bench.py's synthetic sample through the torch FeatureNet and the CPU oracle cascade (test infrastructure, not the
product), then for every Evaluation call (stage, iteration) and source view projects all hypotheses of every TWxTH pixel
tile and reports the bounding box of the tap texels per chunk of DCH consecutive hypotheses: width/height percentiles,
texel counts, and the tap re-use factor a tile-private LDS window would get (taps / window texels).

    python tests/studies/footprint_study.py [--width 1600 --height 1200] [--tw 16 --th 4 --dch 8]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import synth  # noqa: E402


def tap_origin(P, depth, hs, ws):
    """north-west tap texel (x0, y0) of every (d, y, x): models/module.py:148-181 positions, floor."""
    D, h, w = depth.shape
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    rx = P[0, 0] * xs + P[0, 1] * ys + P[0, 2]
    ry = P[1, 0] * xs + P[1, 1] * ys + P[1, 2]
    rz = P[2, 0] * xs + P[2, 1] * ys + P[2, 2]
    px = rx[None] * depth + P[0, 3]
    py = ry[None] * depth + P[1, 3]
    pz = rz[None] * depth + P[2, 3]
    gx = px / pz * (ws - 1) / (w - 1)
    gy = py / pz * (hs - 1) / (h - 1)
    x0 = np.clip(np.floor(gx), 0, ws - 2).astype(np.int64)
    y0 = np.clip(np.floor(gy), 0, hs - 2).astype(np.int64)
    return x0, y0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--height", type=int, default=1200)
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--tw", type=int, default=16)
    ap.add_argument("--th", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--dch", type=int, nargs="+", default=[4, 8])
    ap.add_argument("--features", default="net", choices=["net", "synth"])
    ap.add_argument("--share", action="store_true", help="report register-level tap sharing statistics instead of windows")
    args = ap.parse_args()
    H, W, N = args.height, args.width, args.views

    import bench
    import patchmatchnet_amd as P
    from oracle import oracle as O
    with np.load(os.path.join(ROOT, "tests", "golden", "params_000007.npz")) as z:
        params = {k: z[k] for k in z.files}
    intr, extr = synth.synthetic_cameras(N + 1, H, W)
    if args.features == "net":
        model = P.PatchmatchNet(**bench.DEFAULT_KW)
        bench.load_weights(model)
        model.eval()
        s = bench.make_samples(1, N + 1, H, W, "cpu", 0)[0]
        with torch.no_grad():
            feats = [{k: v.numpy() for k, v in model.feature(im).items()} for im in s["images"]]
    else:
        f3 = synth.synthetic_features(N + 1, 64, H // 8, W // 8, 0)
        f2 = synth.synthetic_features(N + 1, 32, H // 4, W // 4, 1)
        f1 = synth.synthetic_features(N + 1, 16, H // 2, W // 2, 2)
        feats = [{3: f3[i].numpy(), 2: f2[i].numpy(), 1: f1[i].numpy()} for i in range(N + 1)]
    noise = torch.rand(1, 48, H // 8, W // 8, generator=torch.Generator().manual_seed(1234)).numpy()
    O.set_num_threads(os.cpu_count() or 1)
    trace = {}
    O.cascade(params, feats, intr, extr, np.array([425.0], np.float32), np.array([935.0], np.float32), noise, trace=trace)

    scale = {3: 0.125, 2: 0.25, 1: 0.5}
    for stage in (3, 2, 1):
        proj = O.stage_projections(intr, extr, scale[stage]).astype(np.float64)
        for it, rec in enumerate(trace[stage]):
            ds = rec["depth_sample"][0].astype(np.float64)  # [D,h,w]
            D, h, w = ds.shape
            dmap = rec["depth"][0]
            gx = np.abs(np.diff(dmap, axis=1)).mean() / (935 - 425)
            print(f"== stage {stage} iter {it + 1}: D={D} {h}x{w}; mean |d depth/dx| = {gx * 100:.3f}% of range")
            for v in range(1, N + 1):
                rel = proj[0, v] @ np.linalg.inv(proj[0, 0])
                x0, y0 = tap_origin(rel, ds, h, w)
                if args.share:
                    # per-pixel reuse across hypotheses and across horizontally adjacent pixels (register-level tap sharing)
                    key = y0 * 65536 + x0
                    same_prev = (key[1:] == key[:-1]).mean()
                    srt = np.sort(key, axis=0)
                    distinct = 1 + (srt[1:] != srt[:-1]).sum(axis=0)
                    cols = np.sort(y0 * 65536 * 4 + x0, axis=0)  # distinct 2x2 blocks vs distinct columns
                    east = ((x0[:, :, 1:] == x0[:, :, :-1] + 1) & (y0[:, :, 1:] == y0[:, :, :-1])).mean()
                    eq = ((x0[:, :, 1:] == x0[:, :, :-1]) & (y0[:, :, 1:] == y0[:, :, :-1])).mean()
                    # union of texels over the D hypotheses of one pixel (what a lane walking its own pixel would load once)
                    tex = set()
                    print(f"   view {v}: NW texel same as previous hypothesis {same_prev * 100:.1f}%; distinct NW texels per pixel over D={D}: "
                          f"mean {distinct.mean():.2f} p90 {np.percentile(distinct, 90):.0f}; next pixel NW = mine+1: {east * 100:.1f}%, = mine: {eq * 100:.1f}%")
                    continue
                for th in args.th:
                    for dch in args.dch:
                        if dch > D:
                            continue
                        tw = args.tw
                        hh, ww = (h // th) * th, (w // tw) * tw
                        bw, bh = [], []
                        for c0 in range(0, D, dch):
                            xs = x0[c0:c0 + dch, :hh, :ww].reshape(-1, hh // th, th, ww // tw, tw)
                            ysb = y0[c0:c0 + dch, :hh, :ww].reshape(-1, hh // th, th, ww // tw, tw)
                            bw.append(xs.max(axis=(0, 2, 4)) - xs.min(axis=(0, 2, 4)) + 2)
                            bh.append(ysb.max(axis=(0, 2, 4)) - ysb.min(axis=(0, 2, 4)) + 2)
                        bw, bh = np.stack(bw).ravel(), np.stack(bh).ravel()
                        tex = bw * bh
                        taps = tw * th * dch * 4
                        q = lambda a, p: float(np.percentile(a, p))
                        print(f"   view {v} tile {tw}x{th} dch {dch}: W p50/p90/p99/max {q(bw,50):.0f}/{q(bw,90):.0f}/{q(bw,99):.0f}/{bw.max()}"
                              f"  H {q(bh,50):.0f}/{q(bh,99):.0f}/{bh.max()}  texels p50/p90/p99 {q(tex,50):.0f}/{q(tex,90):.0f}/{q(tex,99):.0f}"
                              f"  reuse(mean) {taps / tex.mean():.1f}")


if __name__ == "__main__":
    main()
