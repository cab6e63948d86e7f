#!/usr/bin/env python
"""Micro-benchmark of the hot-path kernels at BASELINE cfg-2 shapes (1600x1200, N=5): one row per launch shape.

    python scripts/kernel_bench.py [--reps 20] [--only warp|agg|all] [--views 5] [--width 1600 --height 1200]

Timing = HIP events on the launch stream around single launches (median of --reps after 3 warm-ups), random data.
Algorithmic bytes of pmn_warp_correlate per SURVEY.md 8(d): 4*h*w*[(1+N)*C + D + N + G*D].
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

import synth  # noqa: E402


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="all")
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--height", type=int, default=1200)
    ap.add_argument("--flat", action="store_true", help="all hypotheses of a pixel equal: ~100%% L1 hits (upper bound probe)")
    ap.add_argument("--lib", default=None, help="A/B: load this build of libpmn_hip.so instead of the tree's")
    args = ap.parse_args()
    if args.lib:
        from patchmatchnet_amd import _lib
        _lib.LIB_PATH = os.path.abspath(args.lib)
    import patchmatchnet_amd as P
    from patchmatchnet_amd import ops, params
    dev = "cuda:0"
    H, W, N = args.height, args.width, args.views
    with np.load(os.path.join(ROOT, "tests", "golden", "params_000007.npz")) as z:
        sd = {k: torch.from_numpy(z[k]) for k in z.files}
    model = P.PatchmatchNet(patchmatch_interval_scale=[0.005, 0.0125, 0.025], propagation_range=[6, 4, 2],
                            patchmatch_iteration=[1, 2, 2], patchmatch_num_sample=[8, 8, 16],
                            propagate_neighbors=[0, 8, 16], evaluate_neighbors=[9, 9, 9])
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    intr, extr = synth.synthetic_cameras(N + 1, H, W)
    gen = torch.Generator().manual_seed(0)
    rows = []
    total_ms, total_bytes = 0.0, 0
    # (stage, D, pixelwise?) for the five Evaluation calls of one forward
    for stage, D, pixelwise, count in [(3, 64, True, 1), (3, 32, False, 1), (2, 16, False, 2), (1, 8, False, 1)]:
        scale = {3: 8, 2: 4, 1: 2}[stage]
        C, G = {3: (64, 8), 2: (32, 8), 1: (16, 4)}[stage]
        h, w = H // scale, W // scale
        pm = getattr(model, f"patchmatch_{stage}")
        feats = [f.to(dev) for f in synth.synthetic_features(N + 1, C, h, w, seed=stage)]
        proj = synth.stage_projections(intr, extr, 1.0 / scale)
        ref_nhwc = ops.nchw_to_nhwc(feats[0])
        src_nhwc = ops.stack_sources_nhwc(feats[1:])
        rel = ops.relative_projection([torch.from_numpy(proj[:, i]).to(dev) for i in range(1, N + 1)],
                                      torch.from_numpy(proj[:, 0]).to(dev))
        # hypotheses: sorted uniform in inverse depth over [1/935, 1/425] (stage 3) or a +-k band (others)
        inv = 1 / 935.0 + torch.rand(1, D, h, w, generator=gen) * (1 / 425.0 - 1 / 935.0)
        if stage != 3 or not pixelwise:
            centre = 1 / 935.0 + torch.rand(1, 1, h, w, generator=gen) * (1 / 425.0 - 1 / 935.0)
            band = {3: 0.025, 2: 0.0125, 1: 0.005}[stage] * (1 / 425.0 - 1 / 935.0)
            k = (torch.arange(D).float() - D // 2).view(1, D, 1, 1) * (8.0 / D if stage == 1 else 1.0)
            inv = (centre + band * k).clamp(1 / 935.0, 1 / 425.0)
        if args.flat:
            inv = inv[:, :1].expand(-1, D, -1, -1)
        hyp = (1.0 / inv).sort(dim=1)[0].contiguous().to(dev)
        vw = None if pixelwise else torch.rand(1, N, h, w, generator=gen).to(dev)
        sim_mlp = pm.evaluation.similarity_net.packed_device()
        pix_mlp = pm.evaluation.pixel_wise_net.packed_device()
        nbytes = 4 * h * w * ((1 + N) * C + D + N + G * D)
        if args.only in ("all", "warp"):
            med, mn = timed(lambda: ops.warp_correlate(ref_nhwc, src_nhwc, rel, hyp, vw, 0, sim_mlp,
                                                       pix_mlp if pixelwise else None, G), args.reps)
            rows.append((f"warp_correlate s{stage} D{D} {'pixelwise' if pixelwise else 'vw'} x{count}", med, mn,
                         nbytes / med / 1e6))
            total_ms += med * count
            total_bytes += nbytes * count
        if args.only in ("all", "agg"):
            K = 9
            eval_off = (0.5 * torch.randn(1, 2 * K, h, w, generator=gen)).to(dev)
            fw = torch.rand(1, K, h, w, generator=gen).to(dev)
            # cost / xnorm hypothesis-last, as pmn_warp_correlate / pmn_init_hypotheses hand them over
            cost = torch.randn(1, h, w, D, generator=gen).to(dev).permute(0, 3, 1, 2)
            xn = (((1.0 / hyp) - 1 / 935.0) / (1 / 425.0 - 1 / 935.0)).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
            med, mn = timed(lambda: ops.aggregate_regress(cost, hyp, xn, fw, eval_off, pm._etable,
                                                          pm.patchmatch_interval_scale, stage == 1), args.reps)
            rows.append((f"aggregate_regress s{stage} D{D} x{count}", med, mn, 0.0))
            med, mn = timed(lambda: ops.feature_weight(ref_nhwc, eval_off, pm._etable,
                                                       pm.feature_weight_net.packed_device(), G), args.reps)
            rows.append((f"feature_weight s{stage}", med, mn, 0.0))
    for name, med, mn, gbps in rows:
        print(f"{name:48s} median {med * 1e3:9.1f} us   min {mn * 1e3:9.1f} us   {gbps:8.1f} GB/s(alg)")
    if total_ms:
        print(f"warp_correlate per depth map: {total_ms:.3f} ms, {total_bytes / 1e6:.1f} MB algorithmic, "
              f"{total_bytes / total_ms / 1e6:.1f} GB/s = {total_bytes / total_ms / 1e6 / 8000 * 100:.2f}% of 8 TB/s")


if __name__ == "__main__":
    main()
