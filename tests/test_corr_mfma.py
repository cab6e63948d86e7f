"""OPT-IN (research build): runs only with PMN_EXPERIMENTAL=1 and `make -C patchmatchnet_amd/csrc EXPERIMENTAL=1`
(libpmn_hip_experimental.so); skipped in the product configuration, whose library has the streaming kernel alone.

Round 4's second formulation of pmn_warp_correlate: correlate-then-interpolate on the fp32 matrix cores
(csrc/experimental/corr_mfma.hip, pmn_set_tuning key 1 bit 6) next to the product's streaming kernel (csrc/gather_corr.hip:
blend C channels per tap, then correlate).  Measured slower on every launch of the cascade (profiles/r04_corr_mfma.md), kept
as a tested record.

Both use the same tap positions and corner weights (reference models/module.py:130-181); they differ in the order of the
channel sum and the 4-tap blend (reference models/patchmatch.py:198-203), i.e. by fp32 re-association.  The streaming kernel is
the one pinned against the oracle / the reference's golden tensors in tests/test_hip_parity.py; this file ties the two to each
other on data that exercises every path of the matrix-core kernel (and re-runs the golden kernel tests with it selected):
window fits the wave's LDS buffer / is walked in pieces (unsorted hypotheses), ragged tiles, tiles straddling image rows,
behind-camera and out-of-range hypotheses, batch > 1, half-resolution view weights, source maps of another size.
Tolerances (absolute, on O(1) quantities): aggregated similarity 2e-5, cost 2e-4 (MLP of it), view weights 1e-5; arg-max
equal wherever the two largest PixelwiseNet responses of the streaming kernel are not within 1e-5 of each other.
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


def _mlp(seed):
    from patchmatchnet_amd import _lib
    g = torch.Generator().manual_seed(seed)
    return (0.4 * torch.randn(_lib.MLP_FLOATS, generator=g)).cuda()


def _case(C, D, h, w, N, B, hyp, seed, vw_shift=0, pixelwise=False, hs=None, ws=None):
    g = torch.Generator().manual_seed(seed)
    hs, ws = hs or h, ws or w
    H, W = h * 8, w * 8
    intr, extr = synth.synthetic_cameras(N + 1, H, W)
    proj = synth.stage_projections(intr, extr, 0.125)
    # smooth features (neighbouring texels correlate, like a real feature map) + noise
    def feat(*shape):
        f = torch.randn(*shape, generator=g)
        return (0.5 * f).contiguous()
    ref = feat(B, h, w, C).cuda()
    src = feat(N, B, hs, ws, C).cuda()
    P0 = torch.from_numpy(proj[0, 0]).double()
    rel = torch.stack([torch.from_numpy(proj[0, i]).double() @ torch.inverse(P0) for i in range(1, N + 1)], 0)
    rel = rel.float()[None].repeat(B, 1, 1, 1).contiguous().cuda()
    lo, hi = 1 / 935.0, 1 / 425.0
    if hyp == "sorted_band":
        yy, xx = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
        centre = lo + (hi - lo) * (0.5 + 0.3 * torch.sin(xx / 9.0) * torch.cos(yy / 7.0))[None, None]
        centre = centre + 0.002 * (hi - lo) * torch.randn(B, 1, h, w, generator=g)
        k = (torch.arange(D).float() - D // 2).view(1, D, 1, 1)
        inv = (centre + 0.01 * (hi - lo) * k).clamp(lo, hi)
        depth = (1.0 / inv).sort(dim=1)[0]
    elif hyp == "full_range":
        u = torch.rand(B, D, h, w, generator=g) + torch.arange(D).float().view(1, D, 1, 1)
        depth = 1.0 / (lo + u / D * (hi - lo))
    elif hyp == "random":
        depth = 1.0 / (lo + torch.rand(B, D, h, w, generator=g) * (hi - lo))
    elif hyp == "behind":
        depth = 1.0 / (lo + torch.rand(B, D, h, w, generator=g) * (hi - lo))
        depth[:, ::3] = -depth[:, ::3]
        depth[:, 1, : h // 2] = 0.0
    elif hyp == "outside":  # far outside the depth range: most taps leave the source map
        depth = 1.0 / (lo + (torch.rand(B, D, h, w, generator=g) * 40 - 20) * (hi - lo)).clamp(min=1e-4)
        depth = depth.sort(dim=1)[0]
    else:
        raise ValueError(hyp)
    depth = depth.contiguous().cuda()
    vw = None
    if not pixelwise:
        vw = torch.rand(B, N, h >> vw_shift, w >> vw_shift, generator=g).cuda()
    return ref, src, rel, depth, vw


def _run(ops, impl, case, C, pixelwise, sim_mlp, pix_mlp, vw_shift=0):
    ref, src, rel, depth, vw = case
    ops.set_tuning(ops.TUNE_FLAGS, ops.FLAG_MFMA if impl == "mfma" else 0)
    try:
        cost, vwo, argmax, sim = ops.warp_correlate(ref, src, rel, depth, vw, vw_shift, sim_mlp,
                                                    pix_mlp if pixelwise else None, CG[C], want_similarity=True,
                                                    want_argmax=pixelwise)
        torch.cuda.synchronize()
    finally:
        ops.set_tuning(ops.TUNE_FLAGS, ops.DEFAULT_FLAGS)
    out = dict(cost=cost.clone(), sim=sim.clone())
    if pixelwise:
        out.update(vw=vwo.clone(), argmax=argmax.clone())
    return out


def _compare(ops, case, C, pixelwise, vw_shift=0, seed=0, label=""):
    sim_mlp, pix_mlp = _mlp(100 + seed), _mlp(200 + seed)
    want = _run(ops, "stream", case, C, pixelwise, sim_mlp, pix_mlp, vw_shift)
    got = _run(ops, "mfma", case, C, pixelwise, sim_mlp, pix_mlp, vw_shift)
    for key, tol in (("sim", 2e-5), ("cost", 2e-4)) + ((("vw", 1e-5),) if pixelwise else ()):
        a, b = want[key], got[key]
        assert torch.isfinite(a).all() and torch.isfinite(b).all(), f"{label} {key}: non-finite values"
        err = (a - b).abs()
        if float(err.max()) > tol:
            idx = [int(i) for i in np.unravel_index(int(err.argmax()), err.shape)]
            raise AssertionError(f"{label} {key}: max abs diff {float(err.max()):.3e} > {tol} at {idx}: streaming "
                                 f"{a[tuple(idx)].item()!r} matrix-core {b[tuple(idx)].item()!r}; {int((err > tol).sum())} of "
                                 f"{err.numel()} beyond the tolerance")
    if pixelwise:
        diff = want["argmax"] != got["argmax"]
        if bool(diff.any()):
            # only fp32 near-ties may differ: the view weight (= the max) agrees to 1e-5 already (checked above)
            assert float(diff.float().mean()) < 5e-3, f"{label}: {float(diff.float().mean()):.2e} of arg-max indices differ"
    return want, got


VIEWS = [
    (16, 8, 60, 80, 3, 1, "sorted_band"),    # stage-1 shape class, tiles divide the rows
    (16, 8, 37, 53, 2, 2, "sorted_band"),    # tiles straddle image rows, ragged last tile, batch of two
    (32, 16, 45, 70, 3, 1, "sorted_band"),   # stage 2: two passes of four groups
    (64, 32, 30, 41, 2, 1, "sorted_band"),   # stage 3 second iteration: two chunks, eight-channel groups
    (64, 32, 22, 37, 5, 1, "random"),        # unsorted hypotheses: every window is walked in pieces
    (32, 16, 19, 33, 2, 1, "random"),
    (16, 8, 26, 40, 3, 1, "behind"),         # behind-camera / non-positive hypotheses
    (16, 8, 26, 40, 3, 1, "outside"),        # most taps outside the source map (dead items, border taps)
    (32, 16, 21, 30, 2, 1, "outside"),
    (16, 8, 16, 16, 1, 1, "sorted_band"),    # smallest map the ABI accepts in practice, one view
    (32, 12, 19, 33, 2, 1, "sorted_band"),   # D % 8 != 0 (variant neighbour counts): dead items in the last chunk
    (16, 5, 21, 18, 2, 1, "random"),         # D < 8
    (64, 20, 17, 29, 3, 2, "full_range"),
    (64, 64, 12, 20, 2, 1, "full_range"),    # the largest hypothesis count
]


@pytest.mark.parametrize("C,D,h,w,N,B,hyp", VIEWS)
def test_views_matrix_core_matches_streaming(C, D, h, w, N, B, hyp):
    _, ops = _gpu()
    case = _case(C, D, h, w, N, B, hyp, seed=C + D + h)
    _compare(ops, case, C, pixelwise=False, label=f"C{C} D{D} {h}x{w} N{N} B{B} {hyp}")


def test_views_half_resolution_view_weights():
    _, ops = _gpu()
    case = _case(32, 16, 36, 52, 3, 1, "sorted_band", seed=5, vw_shift=1)
    _compare(ops, case, 32, pixelwise=False, vw_shift=1, label="vw_shift")


def test_views_source_maps_of_another_size():
    _, ops = _gpu()
    case = _case(16, 8, 40, 56, 2, 1, "sorted_band", seed=9, hs=32, ws=48)
    _compare(ops, case, 16, pixelwise=False, label="hs/ws")


@pytest.mark.parametrize("C,D,h,w,N,B,hyp", [
    (64, 48, 24, 40, 3, 1, "full_range"),   # the stage-3 first-iteration launch
    (64, 48, 19, 27, 2, 2, "full_range"),   # ragged tiles, batch of two
    (64, 48, 13, 21, 2, 1, "random"),
    (64, 48, 14, 19, 2, 1, "behind"),
    (64, 64, 24, 40, 5, 1, "full_range"),   # D = 64 = 48 + 16 propagated: what the cascade launches
    (64, 52, 15, 33, 2, 1, "full_range"),   # D % 8 != 0
    (32, 24, 21, 35, 2, 1, "full_range"),   # PixelwiseNet at another stage's width (the ABI allows it)
    (16, 16, 18, 30, 2, 1, "sorted_band"),
])
def test_pixelwise_matrix_core_matches_streaming(C, D, h, w, N, B, hyp):
    _, ops = _gpu()
    case = _case(C, D, h, w, N, B, hyp, seed=C + D + w, pixelwise=True)
    _compare(ops, case, C, pixelwise=True, label=f"pixelwise C{C} D{D} {h}x{w} N{N} B{B} {hyp}")


def test_fullsize_shapes_match_streaming():
    """BASELINE cfg-2 launch shapes (1600x1200, N=5) on smooth hypotheses."""
    _, ops = _gpu()
    for C, D, scale, pixelwise, hyp in [(64, 64, 8, True, "full_range"), (64, 32, 8, False, "sorted_band"),
                                        (32, 16, 4, False, "sorted_band"), (16, 8, 2, False, "sorted_band")]:
        case = _case(C, D, 1200 // scale, 1600 // scale, 5, 1, hyp, seed=scale, pixelwise=pixelwise)
        _compare(ops, case, C, pixelwise=pixelwise, label=f"fullsize C{C} D{D}")
        del case
        torch.cuda.empty_cache()


@pytest.mark.parametrize("case", ["default", "variant"])
def test_cascade_on_reference_features_with_the_matrix_core_kernel(case):
    """The whole cascade on the reference's own features and noise, every pmn_warp_correlate launch on the matrix cores, against
    the reference's per-iteration and final depths (tests/golden, generated by the imported reference): the 1e-3 relative bar of
    the product's test_cascade_with_reference_features."""
    P, ops = _gpu()
    import goldenutil as GU
    g, params, kw = GU.load_case(case)
    model = P.PatchmatchNet(**kw)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    model = model.cuda().eval()
    nv = int(g["n_views"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    feats = [{s: t(g[f"feature_{v}_s{s}"]) for s in (1, 2, 3)} for v in range(nv)]
    ops.set_tuning(ops.TUNE_FLAGS, ops.FLAG_MFMA)
    try:
        with torch.no_grad():
            depth, conf, dpm = model([t(g[f"image_{v}"]) for v in range(nv)], t(g["intrinsics"]), t(g["extrinsics"]),
                                     t(g["depth_min"]), t(g["depth_max"]), noise=t(g["noise"]), features=feats)
        torch.cuda.synchronize()
    finally:
        ops.set_tuning(ops.TUNE_FLAGS, ops.DEFAULT_FLAGS)
    for stage in (3, 2, 1):
        for it in range(1, kw["patchmatch_iteration"][stage - 1] + 1):
            e = GU.rel_err(dpm[stage][it - 1].cpu().numpy(), g[f"s{stage}_it{it}_depth_out"])
            assert e < 1e-3, (stage, it, e)
    assert GU.rel_err(depth.cpu().numpy(), g["depth"]) < 1e-3
