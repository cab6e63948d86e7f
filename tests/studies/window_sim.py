#!/usr/bin/env python
"""CPU simulation of gather_win.hip's window policy on the real cascade's hypotheses (bench.py sample through the torch
FeatureNet + CPU oracle): per launch and view, the share of (tile, round) windows that had to be cut down, the share of items
that fall outside their window (global path), and the share of wave-steps (64 items) with at least one such item.

    python tests/studies/window_sim.py [--cap 192] [--th 4] [--dch 8]
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "scripts")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch
import synth
from footprint_study import tap_origin


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cap", type=int, nargs="+", default=[192])
    ap.add_argument("--th", type=int, default=4)
    ap.add_argument("--dch", type=int, default=8)
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--height", type=int, default=1200)
    args = ap.parse_args()
    H, W, N = args.height, args.width, 5
    import bench
    import patchmatchnet_amd as P
    from oracle import oracle as O
    with np.load(os.path.join(ROOT, "tests", "golden", "params_000007.npz")) as z:
        params = {k: z[k] for k in z.files}
    intr, extr = synth.synthetic_cameras(N + 1, H, W)
    model = P.PatchmatchNet(**bench.DEFAULT_KW)
    bench.load_weights(model)
    model.eval()
    s = bench.make_samples(1, N + 1, H, W, "cpu", 0)[0]
    with torch.no_grad():
        feats = [{k: v.numpy() for k, v in model.feature(im).items()} for im in s["images"]]
    noise = torch.rand(1, 48, H // 8, W // 8, generator=torch.Generator().manual_seed(1234)).numpy()
    O.set_num_threads(os.cpu_count() or 1)
    trace = {}
    O.cascade(params, feats, intr, extr, np.array([425.0], np.float32), np.array([935.0], np.float32), noise, trace=trace)
    scale = {3: 0.125, 2: 0.25, 1: 0.5}
    tw, th, dch = 16, args.th, args.dch
    for cap in args.cap:
        print(f"##### window capacity {cap} texels ({cap * 64} B per wave), tile {tw}x{th}, {dch} hypotheses per round")
        for stage in (3, 2, 1):
            proj = O.stage_projections(intr, extr, scale[stage]).astype(np.float64)
            for it, rec in enumerate(trace[stage]):
                ds = rec["depth_sample"][0].astype(np.float64)
                D, h, w = ds.shape
                hh, ww = (h // th) * th, (w // tw) * tw
                line = f"stage {stage} it {it + 1} D={D}:"
                tot_items = tot_out = tot_steps = tot_bad_steps = 0
                for v in range(1, N + 1):
                    rel = proj[0, v] @ np.linalg.inv(proj[0, 0])
                    x0, y0 = tap_origin(rel, ds, h, w)
                    n_out = n_items = n_steps = n_bad = n_cut = n_rounds = 0
                    for c0 in range(0, D, dch):
                        xs = x0[c0:c0 + dch, :hh, :ww].reshape(-1, hh // th, th, ww // tw, tw)  # [d, ty, r, tx, c]
                        ys = y0[c0:c0 + dch, :hh, :ww].reshape(-1, hh // th, th, ww // tw, tw)
                        ends = [0, xs.shape[0] - 1]
                        sx0 = xs[ends].min(axis=(0, 2, 4)); sx1 = xs[ends].max(axis=(0, 2, 4))
                        sy0 = ys[ends].min(axis=(0, 2, 4)); sy1 = ys[ends].max(axis=(0, 2, 4))
                        bw = sx1 - sx0 + 2; bh = sy1 - sy0 + 2
                        cut = bw * bh > cap
                        bh2 = np.where(cut, np.minimum(bh, 8), bh)
                        bw2 = np.where(cut, np.maximum(np.minimum(bw, cap // bh2), 2), bw)
                        cx = (xs[0, :, th // 2, :, tw // 2] + xs[-1, :, th // 2, :, tw // 2]) // 2
                        cy = (ys[0, :, th // 2, :, tw // 2] + ys[-1, :, th // 2, :, tw // 2]) // 2
                        bx0 = np.where(cut, np.minimum(np.maximum(cx - bw2 // 2 + 1, sx0), sx1 + 2 - bw2), sx0)
                        by0 = np.where(cut, np.minimum(np.maximum(cy - bh2 // 2 + 1, sy0), sy1 + 2 - bh2), sy0)
                        lx = xs - bx0[None, :, None, :, None]; ly = ys - by0[None, :, None, :, None]
                        outside = (lx < 0) | (lx >= (bw2 - 1)[None, :, None, :, None]) | (ly < 0) | (ly >= (bh2 - 1)[None, :, None, :, None])
                        n_out += outside.sum(); n_items += outside.size
                        st = outside.any(axis=(2, 4))  # [d, ty, tx]: a wave-step = one d of one tile
                        n_bad += st.sum(); n_steps += st.size
                        n_cut += cut.sum(); n_rounds += cut.size
                    line += f"  v{v}: cut {100 * n_cut / n_rounds:4.1f}% out {100 * n_out / n_items:5.2f}% steps {100 * n_bad / n_steps:4.1f}%"
                    tot_items += n_items; tot_out += n_out; tot_steps += n_steps; tot_bad_steps += n_bad
                print(line + f"  | all: out {100 * tot_out / tot_items:5.2f}% steps {100 * tot_bad_steps / tot_steps:4.1f}%")


if __name__ == "__main__":
    main()
