"""OPT-IN (research build): runs only with PMN_EXPERIMENTAL=1 and `make -C patchmatchnet_amd/csrc EXPERIMENTAL=1`
(libpmn_hip_experimental.so); skipped in the product configuration, whose library has the streaming family alone.

pmn_warp_correlate has three kernel families -- lane = item (csrc/gather_lane.hip, the default) and the first windowed
form (csrc/gather_win.hip), both taking their taps from wave-private LDS windows of the source maps, and streaming (csrc/gather_corr.hip: taps straight from HBM/L1).  They implement the same arithmetic in the
same order (reference models/module.py:130-181, models/patchmatch.py:192-217, :570, :695-702), so they must agree BIT FOR
BIT on any input; the streaming family is the one pinned against the oracle / the reference's golden tensors in
tests/test_hip_parity.py, and this file ties the windowed family to it on shapes and data that exercise every path:
window fits / is cut down / is useless (unsorted hypotheses -> per-lane global fallback), ragged tile edges, partial
hypothesis chunks, behind-camera hypotheses, batch > 1, half-resolution view weights.
"""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu

CG = {64: 8, 32: 8, 16: 4}  # channels -> groups (reference models/net.py:153-158)


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    from patchmatchnet_amd import _lib
    if not _lib.experimental():
        pytest.skip("research kernel families: opt in with PMN_EXPERIMENTAL=1 (+ make EXPERIMENTAL=1)")
    import patchmatchnet_amd as P
    from patchmatchnet_amd import ops
    P.lib()
    return P, ops


def _mlp(ops, seed):
    from patchmatchnet_amd import _lib
    g = torch.Generator().manual_seed(seed)
    return (0.4 * torch.randn(_lib.MLP_FLOATS, generator=g)).cuda()


def _case(C, D, h, w, N, B, hyp, seed, vw_shift=0, pixelwise=False):
    """Random stage inputs with the synthetic DTU-like cameras; ``hyp`` picks the hypothesis pattern."""
    g = torch.Generator().manual_seed(seed)
    H, W = h * 8, w * 8
    intr, extr = synth.synthetic_cameras(N + 1, H, W)
    proj = synth.stage_projections(intr, extr, 0.125)
    ref = (0.5 * torch.randn(B, h, w, C, generator=g)).cuda()
    src = (0.5 * torch.randn(N, B, h, w, C, generator=g)).cuda()
    P0 = torch.from_numpy(proj[0, 0]).double()
    rel = torch.stack([torch.from_numpy(proj[0, i]).double() @ torch.inverse(P0) for i in range(1, N + 1)], 0)
    rel = rel.float()[None].repeat(B, 1, 1, 1).contiguous().cuda()
    lo, hi = 1 / 935.0, 1 / 425.0
    if hyp == "sorted_band":  # what the cascade produces: a sorted band around a smooth-ish centre
        yy, xx = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
        centre = lo + (hi - lo) * (0.5 + 0.3 * torch.sin(xx / 9.0) * torch.cos(yy / 7.0))[None, None]
        centre = centre + 0.01 * (hi - lo) * torch.randn(B, 1, h, w, generator=g)
        k = (torch.arange(D).float() - D // 2).view(1, D, 1, 1)
        inv = (centre + 0.02 * (hi - lo) * k).clamp(lo, hi)
        depth = (1.0 / inv).sort(dim=1)[0]
    elif hyp == "full_range":  # stage-3 first iteration: the whole range, per-pixel jitter
        u = torch.rand(B, D, h, w, generator=g) + torch.arange(D).float().view(1, D, 1, 1)
        depth = 1.0 / (lo + u / D * (hi - lo))
    elif hyp == "random":  # unsorted, anywhere in the range: windows are useless, every path still has to be right
        depth = 1.0 / (lo + torch.rand(B, D, h, w, generator=g) * (hi - lo))
    elif hyp == "behind":  # some hypotheses behind the source cameras / negative (reference sentinel path)
        depth = 1.0 / (lo + torch.rand(B, D, h, w, generator=g) * (hi - lo))
        depth[:, ::3] = -depth[:, ::3]
        depth[:, 1, : h // 2] = 0.0
    else:
        raise ValueError(hyp)
    depth = depth.contiguous().cuda()
    vw = None
    if not pixelwise:
        vw = torch.rand(B, N, h >> vw_shift, w >> vw_shift, generator=g).cuda()
    return ref, src, rel, depth, vw


def _run(ops, case, C, pixelwise, sim_mlp, pix_mlp, vw_shift=0):
    ref, src, rel, depth, vw = case
    cost, vwo, argmax, sim = ops.warp_correlate(ref, src, rel, depth, vw, vw_shift, sim_mlp, pix_mlp if pixelwise else None,
                                                CG[C], want_similarity=True, want_argmax=pixelwise)
    torch.cuda.synchronize()
    out = [cost.clone(), sim.clone()]
    if pixelwise:
        out += [vwo.clone(), argmax.clone()]
    return out


def _compare(ops, case, C, pixelwise, vw_shift=0, caps=(12288,), seed=0):
    sim_mlp, pix_mlp = _mlp(ops, 100 + seed), _mlp(ops, 200 + seed)
    try:
        ops.set_tuning(ops.TUNE_FLAGS, 0)  # streaming kernels
        want = _run(ops, case, C, pixelwise, sim_mlp, pix_mlp, vw_shift)
        # families: lane = item engine built for 3 / 2 waves per SIMD, first windowed form with / without quad rotation
        variants = [(ops.FLAG_WINDOWED, 3), (ops.FLAG_WINDOWED, 2), (ops.FLAG_WINDOWED | ops.FLAG_WIN_V1, 3),
                    (ops.FLAG_WINDOWED | ops.FLAG_WIN_V1 | ops.FLAG_NO_ROTATION, 3)]
        if not pixelwise:  # the tile-window kernel (gather_tile.hip) covers the known-weights launches
            variants.append((ops.FLAG_TILE, 3))
        for cap in caps:
            for flags, wps in variants:
                ops.set_tuning(ops.TUNE_FLAGS, flags)
                ops.set_tuning(ops.TUNE_WINDOW_BYTES, cap)
                ops.set_tuning(ops.TUNE_WINDOW_BYTES_PIXELWISE, min(cap, 8192))
                ops.set_tuning(ops.TUNE_LANE_WINDOW_BYTES, cap)
                ops.set_tuning(ops.TUNE_LANE_WAVES_PER_SIMD, wps)
                ops.set_tuning(ops.TUNE_TILE_WINDOW_BYTES, max(16384, cap))  # 16 KB: small enough for stragglers on the tests' data
                got = _run(ops, case, C, pixelwise, sim_mlp, pix_mlp, vw_shift)
                for i, (a, b) in enumerate(zip(want, got)):
                    assert torch.isfinite(a.float()).all()
                    same = torch.equal(a, b)
                    if not same:
                        bad = (a != b)
                        idx = bad.nonzero()[0].tolist()
                        raise AssertionError(f"output {i}: {int(bad.sum())} of {a.numel()} elements differ (cap {cap}, flags "
                                             f"{flags}, wps {wps}); first at {idx}: streaming {a[tuple(idx)].item()!r} windowed "
                                             f"{b[tuple(idx)].item()!r}")
    finally:
        ops.set_tuning(ops.TUNE_FLAGS, ops.DEFAULT_FLAGS)
        ops.set_tuning(ops.TUNE_WINDOW_BYTES, 12288)
        ops.set_tuning(ops.TUNE_WINDOW_BYTES_PIXELWISE, 8192)
        ops.set_tuning(ops.TUNE_LANE_WINDOW_BYTES, 12288)
        ops.set_tuning(ops.TUNE_LANE_WAVES_PER_SIMD, 3)
        ops.set_tuning(ops.TUNE_TILE_WINDOW_BYTES, 20480)


@pytest.mark.parametrize("C,D,h,w,N,B,hyp", [
    (16, 8, 60, 80, 3, 1, "sorted_band"),     # stage-1 shape class: one chunk, tiles divide the map
    (16, 8, 37, 53, 2, 2, "sorted_band"),     # ragged right / bottom tile edges, batch of two
    (32, 16, 45, 70, 3, 1, "sorted_band"),    # stage 2: two hypothesis chunks, two channel slices
    (64, 32, 30, 41, 2, 1, "sorted_band"),    # stage 3 second iteration: four slices
    (64, 32, 22, 37, 5, 1, "random"),         # windows useless: per-lane global fallback everywhere
    (32, 12, 19, 33, 2, 1, "sorted_band"),    # partial last chunk (D % 8 != 0: variant neighbour counts)
    (16, 5, 21, 18, 2, 1, "random"),          # D < 8
    (16, 8, 26, 40, 3, 1, "behind"),          # behind-camera / non-positive hypotheses
    (64, 20, 17, 29, 3, 2, "full_range"),
])
def test_views_windowed_equals_streaming(C, D, h, w, N, B, hyp):
    _, ops = _gpu()
    case = _case(C, D, h, w, N, B, hyp, seed=C + D + h)
    _compare(ops, case, C, pixelwise=False, caps=(12288, 8192, 32768))


def test_views_half_resolution_view_weights():
    _, ops = _gpu()
    case = _case(32, 16, 36, 52, 3, 1, "sorted_band", seed=5, vw_shift=1)
    _compare(ops, case, 32, pixelwise=False, vw_shift=1)


@pytest.mark.parametrize("C,D,h,w,N,B,hyp", [
    (64, 64, 24, 40, 3, 1, "full_range"),  # the stage-3 first-iteration launch
    (64, 64, 19, 27, 2, 2, "full_range"),  # ragged tiles, batch of two
    (64, 52, 15, 33, 2, 1, "full_range"),  # D % 8 != 0
    (64, 64, 13, 21, 2, 1, "random"),
    (64, 32, 14, 19, 2, 1, "behind"),
    (32, 24, 21, 35, 2, 1, "full_range"),  # PixelwiseNet at another stage's width (API allows it)
    (16, 16, 18, 30, 2, 1, "sorted_band"),
])
def test_pixelwise_windowed_equals_streaming(C, D, h, w, N, B, hyp):
    _, ops = _gpu()
    case = _case(C, D, h, w, N, B, hyp, seed=C + D + w, pixelwise=True)
    _compare(ops, case, C, pixelwise=True, caps=(8192, 4096))


def test_fullsize_windowed_equals_streaming():
    """BASELINE cfg-2 stage shapes (1600x1200, N=5): every launch of the cascade, both families, bit-identical."""
    _, ops = _gpu()
    for C, D, scale, pixelwise, hyp in [(64, 64, 8, True, "full_range"), (64, 32, 8, False, "sorted_band"),
                                        (32, 16, 4, False, "sorted_band"), (16, 8, 2, False, "sorted_band")]:
        case = _case(C, D, 1200 // scale, 1600 // scale, 5, 1, hyp, seed=scale, pixelwise=pixelwise)
        _compare(ops, case, C, pixelwise=pixelwise, caps=(12288,) if not pixelwise else (8192,))
        del case
        torch.cuda.empty_cache()
