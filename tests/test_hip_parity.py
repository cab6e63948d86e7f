"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the reference-generated golden vectors.

Run with ``pytest -m gpu`` on an MI355X.  Tolerances: north_star asks <= 1e-3 relative on depth and exact view-weight
indices; kernel-level comparisons (same inputs on both sides) are held much tighter than that.
"""
import numpy as np
import pytest
import torch

import goldenutil as GU
import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _research():
    """Rounds 1-2's fp32 alternatives live in the research build since round 4 (ABI 17): opt in with PMN_EXPERIMENTAL=1."""
    from patchmatchnet_amd import _lib
    if not _lib.experimental():
        pytest.skip("research build only: PMN_EXPERIMENTAL=1 (+ make -C patchmatchnet_amd/csrc EXPERIMENTAL=1)")


def _gpu():
    # fail loudly (never skip) when selected with -m gpu on a box without a usable GPU / library
    assert torch.cuda.is_available(), "GPU tests selected but no ROCm device is visible"
    import patchmatchnet_amd as P
    P.lib()
    return P


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def n(x):
    return x.detach().cpu().numpy()


def _model(P, params, kw):
    m = P.PatchmatchNet(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return m.to(DEV).eval()


def _configs(kw):
    return O.default_stage_configs(kw["patchmatch_interval_scale"], kw["propagation_range"],
                                   kw["patchmatch_iteration"], kw["patchmatch_num_sample"],
                                   kw["propagate_neighbors"], kw["evaluate_neighbors"])


def test_library_is_the_compute_path():
    P = _gpu()
    assert P.lib().pmn_abi_version() == P._lib.ABI_VERSION
    with pytest.raises(P.PmnError):  # CPU tensors are refused, there is no fallback
        P.ops.nchw_to_nhwc(torch.zeros(1, 4, 4, 4))


@pytest.mark.parametrize("name", ["A", "B"])
def test_differentiable_warping_op(name):
    P = _gpu()
    g = GU.load_npz("ops_small.npz")
    out = P.differentiable_warping(t(g[f"{name}_src"]), t(g[f"{name}_src_proj"]), t(g[f"{name}_ref_proj"]),
                                   t(g[f"{name}_depth"]))
    ref = g[f"{name}_warped"]
    assert GU.abs_err(n(out), ref) < 5e-5
    oracle = O.differentiable_warping(g[f"{name}_src"], g[f"{name}_src_proj"], g[f"{name}_ref_proj"], g[f"{name}_depth"])
    frac_zero_mismatch = float(((n(out) == 0) != (oracle == 0)).mean())
    assert frac_zero_mismatch < 1e-3  # the inverse may differ in the last bit between LAPACK and the GPU solver


def test_nchw_to_nhwc_roundtrip():
    P = _gpu()
    x = torch.randn(2, 32, 37, 53, device=DEV)
    y = P.ops.nchw_to_nhwc(x)
    assert torch.equal(y, x.permute(0, 2, 3, 1).contiguous())


@pytest.mark.parametrize("case", ["default", "variant", "counts", "dilations", "rig"])
@pytest.mark.parametrize("stage", [3, 2, 1])
def test_kernels_against_golden(case, stage):
    """Every kernel of one stage, fed the reference's own tensors (incl. its conv offsets), against the reference's
    intermediates (golden) and against the oracle's arg-max indices."""
    P = _gpu()
    g, params, kw = GU.load_case(case)
    cfg = _configs(kw)[stage]
    feats, proj, depth, vw = GU.stage_inputs(g, kw, stage)
    model = _model(P, params, kw)
    pm = getattr(model, f"patchmatch_{stage}")
    B, C, h, w = feats[0].shape

    ref_nhwc = P.ops.nchw_to_nhwc(t(feats[0]))
    src_nhwc = P.ops.stack_sources_nhwc([t(f) for f in feats[1:]])
    # relative projections computed exactly as the oracle does (numpy fp32) so both sides sample identical positions;
    # ops.relative_projection (torch.inverse on the device) is exercised by the module-level tests below
    rel = t(np.stack([np.matmul(proj[:, i], np.linalg.inv(proj[:, 0])) for i in range(1, proj.shape[1])], 1)
            .astype(np.float32))
    eval_off = t(g[f"s{stage}_eval_offsets"])
    propa_off = t(g[f"s{stage}_propa_offsets"]) if f"s{stage}_propa_offsets" in g else None
    dmin, dmax = t(g["depth_min"]), t(g["depth_max"])

    fw = pm.feature_weight_net(ref_nhwc, eval_off, pm._etable)
    assert GU.abs_err(n(fw), g[f"s{stage}_feature_weight"]) < 2e-5

    # oracle run for the arg-max indices (not stored in the fixtures)
    otr = []
    O.patchmatch_stage(cfg, params, feats[0], feats[1:], proj[:, 0], [proj[:, i] for i in range(1, proj.shape[1])],
                       g["depth_min"], g["depth_max"], depth, vw, noise=g["noise"] if stage == 3 else None,
                       propa_offsets=g.get(f"s{stage}_propa_offsets"), eval_offsets=g[f"s{stage}_eval_offsets"],
                       trace=otr)

    for it in range(1, cfg.iterations + 1):
        key = f"s{stage}_it{it}_"
        is_inverse = stage == 1 and it == cfg.iterations
        propagate = cfg.propagate_neighbors > 0 and not (stage == 1 and it == cfg.iterations)
        # inputs of this iteration are the GOLDEN outputs of the previous one: kernels are compared in isolation
        if it == 1:
            d_in = None if depth is None else t(depth)
        else:
            d_in = t(g[f"s{stage}_it{it - 1}_depth_out"])
        first = stage == 3 and it == 1
        hyp, xn = P.ops.init_hypotheses(t(g["noise"]) if first else None, d_in, 0, dmin, dmax, cfg.num_sample,
                                        cfg.interval_scale, propa_off if propagate else None,
                                        pm._ptable if propagate else None, h, w)
        assert GU.rel_err(n(hyp), g[key + "depth_sample"]) < 2e-6
        hyp = t(g[key + "depth_sample"])  # continue from the golden hypotheses
        inv_min, inv_max = 1.0 / g["depth_min"].reshape(-1, 1, 1, 1), 1.0 / g["depth_max"].reshape(-1, 1, 1, 1)
        xn_ref = (1.0 / g[key + "depth_sample"] - inv_max) / (inv_min - inv_max)
        if first:
            vw_in = None
        elif it == 1:
            vw_in = t(vw)
        else:
            vw_in = t(g[f"s{stage}_it{it - 1}_view_weights"])
        cost, vw_out, argmax, sim = P.ops.warp_correlate(
            ref_nhwc, src_nhwc, rel, hyp, vw_in, 0, pm.evaluation.similarity_net.packed_device(),
            pm.evaluation.pixel_wise_net.packed_device() if vw_in is None else None, cfg.G, want_similarity=True,
            want_argmax=vw_in is None)
        # (since round 4 the tap positions come out of the reference's own IEEE chain, pmn_pose_position: what is left is the fp32
        #  summation order of the blend / group mean -- full-size maxima on random features: <= 2.4e-5, profiles/r0*_parity_report.jsonl)
        assert GU.abs_err(n(sim), g[key + "similarity"]) < 3e-5
        assert GU.abs_err(n(vw_out), g[key + "view_weights"]) < 1e-5
        if vw_in is None:
            # "bit-exact on view_weights indices": arg-max over D of the PixelwiseNet response == oracle's
            np.testing.assert_array_equal(n(argmax), otr[it - 1]["view_weight_argmax"])
        # SimilarityNet MLP in isolation: the oracle MLP applied to the kernel's own aggregated similarity
        # (the MLP output spans tens of units, so errors are scaled by max(|ref|, 1))
        cost_ref = O.pointwise_mlp(n(sim), params, f"patchmatch_{stage}.evaluation.similarity_net", "similarity", False)
        assert float((np.abs(n(cost) - cost_ref) / np.maximum(np.abs(cost_ref), 1.0)).max()) < 5e-5
        # ... and end to end (fp32 rounding of the similarity is amplified by the MLP's weights)
        assert float((np.abs(n(cost) - otr[it - 1]["cost"]) / np.maximum(np.abs(otr[it - 1]["cost"]), 1.0)).max()) < 1e-3
        score, dep = P.ops.aggregate_regress(t(otr[it - 1]["cost"]), hyp, t(xn_ref.astype(np.float32)), fw, eval_off,
                                             pm._etable, cfg.interval_scale, is_inverse)
        assert GU.abs_err(n(score), g[key + "score"]) < 2e-4
        assert GU.rel_err(n(dep), g[key + "depth"]) < 2e-5


@pytest.mark.parametrize("case", ["default", "variant", "counts", "dilations", "rig"])
def test_cascade_with_reference_features(case):
    """PatchmatchNet.forward fed the reference's FeatureNet outputs and noise: whole hot path + MIOpen offset heads +
    refinement + confidence vs the reference's final outputs (north_star tolerance 1e-3 relative on depth)."""
    P = _gpu()
    g, params, kw = GU.load_case(case)
    model = _model(P, params, kw)
    nv = int(g["n_views"])
    feats = [{s: t(g[f"feature_{v}_s{s}"]) for s in (1, 2, 3)} for v in range(nv)]
    imgs = [t(g[f"image_{v}"]) for v in range(nv)]
    dbg = {}
    with torch.no_grad():
        depth, conf, dpm = model(imgs, t(g["intrinsics"]), t(g["extrinsics"]), t(g["depth_min"]), t(g["depth_max"]),
                                 noise=t(g["noise"]), features=feats, debug=dbg)
    last = kw["patchmatch_iteration"][0]
    for stage in (3, 2, 1):
        for it in range(1, kw["patchmatch_iteration"][stage - 1] + 1):
            e = GU.rel_err(n(dpm[stage][it - 1]), g[f"s{stage}_it{it}_depth_out"])
            assert e < 1e-3, (stage, it, e)
    assert GU.rel_err(n(dpm[1][last - 1]), g[f"s1_it{last}_depth_out"]) < 1e-3
    assert GU.rel_err(n(depth), g["depth"]) < 1e-3
    assert depth.shape == g["depth"].shape and conf.shape == g["confidence"].shape
    mism = float((np.abs(n(conf) - g["confidence"]) > 1e-3).mean())
    assert mism < 5e-3, mism  # confidence hinges on trunc(sum d*p): statistical criterion


def test_end_to_end_from_images():
    """HIP FeatureNet (the default, hip_feature_net=True: stem / Winograd / MFMA convolutions) + hot path + HIP refinement from
    raw images vs the reference's CPU result.  FeatureNet rounding differs from the CPU backend's and amplifies down the
    cascade, so the criterion is statistical; tests/test_fullsize_parity.py runs both FeatureNet paths and records the maxima."""
    P = _gpu()
    g, params, kw = GU.load_case("default")
    model = _model(P, params, kw)
    nv = int(g["n_views"])
    imgs = [t(g[f"image_{v}"]) for v in range(nv)]
    with torch.no_grad():
        depth, conf, _ = model(imgs, t(g["intrinsics"]), t(g["extrinsics"]), t(g["depth_min"]), t(g["depth_max"]),
                               noise=t(g["noise"]))
    rel = np.abs(n(depth) - g["depth"]) / np.abs(g["depth"])
    assert np.quantile(rel, 0.999) < 1e-3, float(np.quantile(rel, 0.999))
    assert rel.max() < 1e-4, float(rel.max())  # measured 1.6e-6


@pytest.mark.parametrize("stage", [3, 2])
def test_evaluation_forward_reference_signature(stage):
    """Evaluation.forward called exactly as the reference's PatchMatch.forward calls it (models/patchmatch.py:499-511): NCHW
    features, projection matrices, the reference's own materialised ``grid`` [B,K*h,w,2] and ``weight`` [B,D,K,h,w] -- against
    the reference's outputs at that boundary (tests/golden/evaluation_io.npz, generated by make_golden.py)."""
    P = _gpu()
    g, params, kw = GU.load_case("default")
    e = GU.load_npz("evaluation_io.npz")
    model = _model(P, params, kw)
    pm = getattr(model, f"patchmatch_{stage}")
    feats, proj, _, _ = GU.stage_inputs(g, kw, stage)
    vw_in = e[f"s{stage}_it1_view_weights_in"]
    with torch.no_grad():
        depth, score, vw = pm.evaluation(
            t(feats[0]), [t(f) for f in feats[1:]], t(proj[:, 0]), [t(proj[:, i]) for i in range(1, proj.shape[1])],
            t(e[f"s{stage}_it1_depth_sample"]), t(e[f"s{stage}_it1_grid"]), t(e[f"s{stage}_it1_weight"]),
            t(vw_in) if vw_in.size else torch.empty(0, device=DEV), False)
    assert tuple(depth.shape) == e[f"s{stage}_it1_depth"].shape and tuple(score.shape) == e[f"s{stage}_it1_score"].shape
    assert GU.abs_err(n(vw), e[f"s{stage}_it1_view_weights"]) < 1e-5
    assert GU.abs_err(n(score), e[f"s{stage}_it1_score"]) < 2e-4
    assert GU.rel_err(n(depth), e[f"s{stage}_it1_depth"]) < 2e-5
    with pytest.raises(AssertionError):  # the reference's argument checks (models/patchmatch.py:183-189)
        pm.evaluation(t(feats[0]), [t(f) for f in feats[1:]], t(proj[:, 0]), [t(proj[:, 1])], t(e[f"s{stage}_it1_depth_sample"]),
                      t(e[f"s{stage}_it1_grid"]), t(e[f"s{stage}_it1_weight"]), torch.empty(0, device=DEV), False)


def test_view_weight_upsampling_shift_matches_materialised():
    """vw_shift / depth_shift (nearest up-sampling folded into the consumer) == explicit F.interpolate."""
    P = _gpu()
    g, params, kw = GU.load_case("default")
    model = _model(P, params, kw)
    feats, proj, depth, vw = GU.stage_inputs(g, kw, 2)
    pm = model.patchmatch_2
    args = dict(ref_feature=t(feats[0]), src_features=[t(f) for f in feats[1:]], ref_proj=t(proj[:, 0]),
                src_projs=[t(proj[:, i]) for i in range(1, proj.shape[1])], depth_min=t(g["depth_min"]),
                depth_max=t(g["depth_max"]))
    with torch.no_grad():
        a = pm(depth=t(depth), view_weights=t(vw), **args)
        half_d = t(g["s3_it2_depth_out"])
        half_vw = t(g["s3_it2_view_weights"])
        b = pm(depth=half_d, view_weights=half_vw, depth_shift=1, vw_shift=1, **args)
    for x, y in zip(a[0], b[0]):
        assert torch.equal(x, y)
    assert torch.equal(a[1], b[1])


@pytest.mark.parametrize("cin,cout,K,stride,pad,dil,in_nchw,out_nchw,up", [
    (3, 8, 3, 1, 1, 1, True, False, False), (1, 8, 3, 1, 1, 1, True, False, False), (8, 8, 3, 1, 1, 1, False, False, False),
    (8, 16, 5, 2, 2, 1, False, False, False), (16, 16, 3, 1, 1, 1, False, False, False),
    (16, 32, 5, 2, 2, 1, False, False, False), (32, 32, 3, 1, 1, 1, False, False, False),
    (32, 64, 5, 2, 2, 1, False, False, False), (64, 64, 3, 1, 1, 1, False, False, False),
    (64, 64, 1, 1, 0, 1, False, False, False), (32, 64, 1, 1, 0, 1, False, False, True),
    (16, 64, 1, 1, 0, 1, False, False, True), (64, 32, 1, 1, 0, 1, False, False, False),
    (64, 16, 1, 1, 0, 1, False, False, False), (64, 32, 3, 1, 2, 2, False, True, False),
    (32, 18, 3, 1, 4, 4, False, True, False), (16, 18, 3, 1, 6, 6, False, True, False), (16, 8, 3, 1, 1, 1, False, False, False)])
def test_conv2d_against_torch(cin, cout, K, stride, pad, dil, in_nchw, out_nchw, up):
    """pmn_conv2d (direct fp32 conv, folded BatchNorm, fused ReLU / FPN add) vs F.conv2d + BatchNorm + ReLU on the device,
    on odd sizes (ragged tiles, borders)."""
    P = _gpu()
    from patchmatchnet_amd import params as PP
    gen = torch.Generator().manual_seed(cin * 100 + cout)
    N, H, W = 2, 38 if stride == 1 else 44, 54 if stride == 1 else 60
    x = torch.randn(N, cin, H, W, generator=gen).to(DEV)
    wt = (0.2 * torch.randn(cout, cin, K, K, generator=gen)).to(DEV)
    use_bn = not out_nchw and not up
    bias = None if use_bn else (0.1 * torch.randn(cout, generator=gen)).to(DEV)
    bn = None
    if use_bn:
        bn = ((0.5 + torch.rand(cout, generator=gen)).to(DEV), (0.1 * torch.randn(cout, generator=gen)).to(DEV),
              (0.1 * torch.randn(cout, generator=gen)).to(DEV), (0.5 + torch.rand(cout, generator=gen)).to(DEV))
    ref = torch.nn.functional.conv2d(x, wt, bias, stride, pad, dil)
    if use_bn:
        ref = torch.nn.functional.batch_norm(ref, bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
    upt = None
    if up:
        upt = torch.randn(N, ref.shape[2] // 2, ref.shape[3] // 2, cout, generator=gen).to(DEV)
        ref = torch.nn.functional.interpolate(upt.permute(0, 3, 1, 2), scale_factor=2.0, mode="bilinear",
                                              align_corners=False) + ref
    relu = use_bn
    if relu:
        ref = torch.relu(ref)
    w, s = PP.pack_conv(wt, bn=bn, bias=bias)
    xin = x if in_nchw else x.permute(0, 2, 3, 1).contiguous()
    out = P.ops.conv2d(xin, torch.from_numpy(w).to(DEV), torch.from_numpy(s).to(DEV), cout, K, stride, pad, dil, relu=relu,
                       up=upt, in_nchw=in_nchw, out_nchw=out_nchw)
    got = out if out_nchw else out.permute(0, 3, 1, 2)
    assert got.shape == ref.shape
    err = float((got - ref).abs().max() / ref.abs().max().clamp_min(1.0))
    assert err < 2e-5, err


@pytest.mark.parametrize("cin,cout,K,stride", [(64, 64, 3, 1), (32, 32, 3, 1), (32, 64, 5, 2), (16, 32, 5, 2)])
@pytest.mark.parametrize("H,W", [(38, 54), (16, 16), (75, 100)])
def test_conv2d_mfma_against_torch(cin, cout, K, stride, H, W):
    """pmn_conv2d_mfma (fp32 implicit GEMM on the matrix cores) vs F.conv2d + BatchNorm + ReLU in float64, and vs the VALU
    kernel pmn_conv2d on the same input; ragged tiles and borders."""
    _research()
    P = _gpu()
    from patchmatchnet_amd import params as PP
    gen = torch.Generator().manual_seed(cin * 100 + cout + H)
    N, pad = 2, K // 2
    H, W = H * stride, W * stride
    x = torch.randn(N, cin, H, W, generator=gen)
    wt = 0.2 * torch.randn(cout, cin, K, K, generator=gen)
    bn = (0.5 + torch.rand(cout, generator=gen), 0.1 * torch.randn(cout, generator=gen),
          0.1 * torch.randn(cout, generator=gen), 0.5 + torch.rand(cout, generator=gen))
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), None, stride, pad)
    ref = torch.relu(torch.nn.functional.batch_norm(ref, bn[2].double(), bn[3].double(), bn[0].double(), bn[1].double(),
                                                    False, 0.0, 1e-5))
    xin = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    w, sh = PP.pack_conv_mfma(wt, bn=bn)
    got = P.ops.conv2d_mfma(xin, torch.from_numpy(w).to(DEV), torch.from_numpy(sh).to(DEV), K, stride, pad, relu=True)
    assert tuple(got.shape) == (N, ref.shape[2], ref.shape[3], cout)
    err = float((got.permute(0, 3, 1, 2).double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err
    w2, s2 = PP.pack_conv(wt, bn=bn)
    valu = P.ops.conv2d(xin, torch.from_numpy(w2).to(DEV), torch.from_numpy(s2).to(DEV), cout, K, stride, pad, relu=True)
    assert float((got - valu).abs().max() / ref.abs().max()) < 1e-5


@pytest.mark.parametrize("cin,dil,n_p,n_e", [(64, 2, 32, 18), (64, 2, 0, 18), (32, 4, 16, 18), (32, 4, 16, 34), (16, 6, 0, 18),
                                             (16, 6, 8, 18)])
@pytest.mark.parametrize("H,W,N", [(37, 50, 2), (16, 16, 1)])
def test_offset_heads_mfma_against_torch(cin, dil, n_p, n_e, H, W, N):
    """pmn_conv2d_mfma planar form: propa_conv + eval_conv of a stage as one dilated 3x3 convolution on the matrix cores vs
    F.conv2d (float64) per head, ragged tiles, batch > 1, padded channel blocks."""
    _research()
    P = _gpu()
    from patchmatchnet_amd import params as PP
    gen = torch.Generator().manual_seed(cin + dil + n_p + H)
    x = torch.randn(N, cin, H, W, generator=gen)
    heads = [(0.1 * torch.randn(c, cin, 3, 3, generator=gen), 0.1 * torch.randn(c, generator=gen)) for c in (n_p, n_e) if c]
    wcat, bcat = torch.cat([w for w, _ in heads], 0), torch.cat([b for _, b in heads], 0)
    w, sh = PP.pack_conv_mfma(wcat, bias=bcat)
    xin = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    ca = n_p if n_p else n_e
    a, b = P.ops.offset_heads_mfma(xin, torch.from_numpy(w).to(DEV), torch.from_numpy(sh).to(DEV), n_p + n_e, ca, dil)
    outs = [a] if b is None else [a, b]
    assert len(outs) == len(heads)
    for got, (wt, bias) in zip(outs, heads):
        ref = torch.nn.functional.conv2d(x.double(), wt.double(), bias.double(), 1, dil, dil)
        assert tuple(got.shape) == tuple(ref.shape) and got.is_contiguous()
        assert float((got.double().cpu() - ref).abs().max() / ref.abs().max()) < 1e-5


@pytest.mark.parametrize("cin,dil,n_p,n_e", [(64, 2, 32, 18), (64, 2, 0, 18), (32, 4, 16, 18), (32, 4, 8, 34), (16, 6, 0, 18),
                                             (16, 6, 8, 18), (16, 6, 16, 34)])
@pytest.mark.parametrize("H,W,N", [(37, 50, 2), (16, 16, 1), (40, 52, 1)])
def test_offset_heads_f16_split_against_torch(cin, dil, n_p, n_e, H, W, N):
    """pmn_offset_heads_f16s: propa_conv + eval_conv of a stage as one dilated 3x3 convolution with bias on the fp16 matrix cores
    (split operands) vs F.conv2d (float64) per head: ragged tiles, widths with and without the float4 store path, batch > 1,
    zero-padded output rows (18 / 34 / 50 channels)."""
    P = _gpu()
    from patchmatchnet_amd import params as PP
    gen = torch.Generator().manual_seed(cin + dil + n_p + H)
    x = torch.randn(N, cin, H, W, generator=gen)
    heads = [(0.1 * torch.randn(c, cin, 3, 3, generator=gen), 0.1 * torch.randn(c, generator=gen)) for c in (n_p, n_e) if c]
    wcat, bcat = torch.cat([w for w, _ in heads], 0), torch.cat([b for _, b in heads], 0)
    w, sh = PP.pack_offset_heads_f16s(wcat, bcat)
    xin = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    ca = n_p if n_p else n_e
    a, b = P.ops.offset_heads_f16s(xin, torch.from_numpy(w).to(DEV), torch.from_numpy(sh).to(DEV), n_p + n_e, ca, dil)
    outs = [a] if b is None else [a, b]
    assert len(outs) == len(heads)
    for got, (wt, bias) in zip(outs, heads):
        ref = torch.nn.functional.conv2d(x.double(), wt.double(), bias.double(), 1, dil, dil)
        assert tuple(got.shape) == tuple(ref.shape) and got.is_contiguous()
        assert float((got.double().cpu() - ref).abs().max() / ref.abs().max()) < 1e-6


@pytest.mark.parametrize("C", [16, 32, 64])
@pytest.mark.parametrize("H,W", [(37, 51), (8, 16), (150, 200)])
def test_conv3x3_winograd_against_torch(C, H, W):
    """pmn_conv3x3_wino (Winograd F(2x2,3x3) on the matrix cores) vs F.conv2d + BatchNorm + ReLU in float64 and vs the direct
    kernel pmn_conv2d; odd sizes (partial tiles, borders), batch 2."""
    _research()
    P = _gpu()
    from patchmatchnet_amd import params as PP
    gen = torch.Generator().manual_seed(C + H)
    x = torch.randn(2, C, H, W, generator=gen)
    wt = 0.2 * torch.randn(C, C, 3, 3, generator=gen)
    bn = (0.5 + torch.rand(C, generator=gen), 0.1 * torch.randn(C, generator=gen), 0.1 * torch.randn(C, generator=gen),
          0.5 + torch.rand(C, generator=gen))
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), None, 1, 1)
    ref = torch.relu(torch.nn.functional.batch_norm(ref, bn[2].double(), bn[3].double(), bn[0].double(), bn[1].double(),
                                                    False, 0.0, 1e-5))
    xin = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    w, sh = PP.pack_conv_wino(wt, bn=bn)
    got = P.ops.conv3x3_wino(xin, torch.from_numpy(w).to(DEV), torch.from_numpy(sh).to(DEV), relu=True)
    assert tuple(got.shape) == (2, H, W, C)
    err = float((got.permute(0, 3, 1, 2).double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err
    w2, s2 = PP.pack_conv(wt, bn=bn)
    direct = P.ops.conv2d(xin, torch.from_numpy(w2).to(DEV), torch.from_numpy(s2).to(DEV), C, 3, 1, 1, relu=True)
    assert float((got - direct).abs().max() / ref.abs().max()) < 1e-5


@pytest.mark.parametrize("cin,cout", [(8, 16), (16, 32), (32, 64)])
@pytest.mark.parametrize("H,W", [(37, 51), (8, 64), (150, 200)])
def test_conv5x5s2_winograd_against_torch(cin, cout, H, W):
    """pmn_conv5x5s2_wino (four parity sub-convolutions in Winograd form on the matrix cores) vs F.conv2d(stride 2, padding 2) +
    BatchNorm + ReLU in float64 and vs the direct kernels; odd sizes (partial tiles, borders), batch 2."""
    _research()
    P = _gpu()
    from patchmatchnet_amd import params as PP
    gen = torch.Generator().manual_seed(cin + H)
    x = torch.randn(2, cin, H, W, generator=gen)
    wt = 0.2 * torch.randn(cout, cin, 5, 5, generator=gen)
    bn = (0.5 + torch.rand(cout, generator=gen), 0.1 * torch.randn(cout, generator=gen), 0.1 * torch.randn(cout, generator=gen),
          0.5 + torch.rand(cout, generator=gen))
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), None, 2, 2)
    ref = torch.relu(torch.nn.functional.batch_norm(ref, bn[2].double(), bn[3].double(), bn[0].double(), bn[1].double(),
                                                    False, 0.0, 1e-5))
    xin = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    w, sh = PP.pack_conv5x5s2_wino(wt, bn=bn)
    got = P.ops.conv5x5s2_wino(xin, torch.from_numpy(w).to(DEV), torch.from_numpy(sh).to(DEV), relu=True)
    assert tuple(got.shape) == (2, ref.shape[2], ref.shape[3], cout)
    err = float((got.permute(0, 3, 1, 2).double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err
    w2, s2 = PP.pack_conv(wt, bn=bn)
    direct = P.ops.conv2d(xin, torch.from_numpy(w2).to(DEV), torch.from_numpy(s2).to(DEV), cout, 5, 2, 2, relu=True)
    assert float((got - direct).abs().max() / ref.abs().max()) < 1e-5


@pytest.mark.parametrize("k,stride,cin,cout", [(3, 1, 16, 16), (3, 1, 32, 32), (3, 1, 64, 64), (5, 2, 8, 16), (5, 2, 16, 32), (5, 2, 32, 64)])
@pytest.mark.parametrize("H,W", [(37, 51), (8, 64), (150, 200)])
def test_conv_f16_split_against_torch(k, stride, cin, cout, H, W):
    """pmn_conv2d_f16s (fp16 matrix cores, hi + lo/2048 split operands, three MFMAs per k-step) vs F.conv2d + BatchNorm + ReLU in
    float64: fp32-convolution accuracy (<= 1e-6 of the output scale; an fp32 direct convolution: 2-4e-7), incl. inputs with a wide
    dynamic range (tiny values next to large ones) and odd sizes (partial tiles, borders), batch 2.  The numpy emulation of the
    same kernel is tests/test_f16s_emulation.py."""
    P = _gpu()
    from patchmatchnet_amd import params as PP
    gen = torch.Generator().manual_seed(cin + H + k)
    x = torch.randn(2, cin, H, W, generator=gen) * (4.0 * torch.rand(1, cin, 1, 1, generator=gen))
    x[:, :, 1:4, 2:6] *= 1e-5
    wt = 0.2 * torch.randn(cout, cin, k, k, generator=gen)
    bn = (0.5 + torch.rand(cout, generator=gen), 0.1 * torch.randn(cout, generator=gen), 0.1 * torch.randn(cout, generator=gen),
          0.5 + torch.rand(cout, generator=gen))
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), None, stride, k // 2)
    ref = torch.relu(torch.nn.functional.batch_norm(ref, bn[2].double(), bn[3].double(), bn[0].double(), bn[1].double(),
                                                    False, 0.0, 1e-5))
    xin = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    w, sh = PP.pack_conv_f16s(wt, bn=bn)
    got = P.ops.conv2d_f16s(xin, torch.from_numpy(w).to(DEV), torch.from_numpy(sh).to(DEV), k, stride, relu=True)
    assert tuple(got.shape) == (2, ref.shape[2], ref.shape[3], cout)
    err = float((got.permute(0, 3, 1, 2).double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 1e-6, err


@pytest.mark.parametrize("H,W", [(37, 51), (16, 16), (150, 200), (50, 68)])
def test_stem_kernels_against_torch(H, W):
    """pmn_stem (conv0 + conv1 on the fp32 VALU) and pmn_stem_f16s (conv1 on the fp16 matrix cores, split operands) vs
    conv + BatchNorm + ReLU twice in float64 (reference models/net.py:17-19, 51); partial tiles, image borders (conv1 pads conv0's
    OUTPUT with zeros, not the image), batch 2."""
    P = _gpu()
    from patchmatchnet_amd import params as PP
    gen = torch.Generator().manual_seed(H)
    x = torch.rand(2, 3, H, W, generator=gen)
    w0, w1 = 0.3 * torch.randn(8, 3, 3, 3, generator=gen), 0.2 * torch.randn(8, 8, 3, 3, generator=gen)
    bns = [(0.5 + torch.rand(8, generator=gen), 0.1 * torch.randn(8, generator=gen), 0.1 * torch.randn(8, generator=gen),
            0.5 + torch.rand(8, generator=gen)) for _ in range(2)]
    t = x.double()
    for w, bn in ((w0, bns[0]), (w1, bns[1])):
        t = torch.nn.functional.conv2d(t, w.double(), None, 1, 1)
        t = torch.relu(torch.nn.functional.batch_norm(t, bn[2].double(), bn[3].double(), bn[0].double(), bn[1].double(), False, 0.0, 1e-5))
    ref = t.permute(0, 2, 3, 1)
    p0 = [torch.from_numpy(a).to(DEV) for a in PP.pack_conv(w0, bn=bns[0])]
    p1 = [torch.from_numpy(a).to(DEV) for a in PP.pack_conv(w1, bn=bns[1])]
    p1h = [torch.from_numpy(a).to(DEV) for a in PP.pack_stem_conv1_f16s(w1, bn=bns[1])]
    got = P.ops.stem(x.to(DEV), *p0, *p1)
    got_h = P.ops.stem_f16s(x.to(DEV), *p0, *p1h)
    for name, g in (("stem", got), ("stem_f16s", got_h)):
        assert tuple(g.shape) == (2, H, W, 8)
        err = float((g.double().cpu() - ref).abs().max() / ref.abs().max())
        assert err < 1e-6, (name, err)
    # pmn_stem_f16s stages the image with aligned float4 loads when W % 4 == 0 and the base is 16-byte aligned, one float at a time
    # otherwise: the same image at a base that is off by one float must give the same bits
    shifted = torch.empty(x.numel() + 1, device=DEV)[1:].view(x.shape).copy_(x.to(DEV))
    assert shifted.data_ptr() % 16 != 0 and shifted.is_contiguous()
    assert torch.equal(P.ops.stem_f16s(shifted, *p0, *p1h), got_h)


def test_fpn_level8_matrix_core_form_matches_valu_form():
    """The 1/8-resolution level of the folded FPN head: pmn_conv2d_mfma's split 1x1 form vs pmn_fpn_level (VALU) vs float64."""
    P = _gpu()
    from patchmatchnet_amd import params as PP
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(3, 19, 27, 64, generator=gen)
    wt = 0.2 * torch.randn(112, 64, generator=gen)
    bias = 0.1 * torch.randn(112, generator=gen)
    ref = torch.einsum("nhwc,dc->nhwd", x.double(), wt.double()) + bias.double()
    w, sh = PP.pack_conv_mfma(wt[:, :, None, None], bias=bias)
    a, b = P.ops.pointwise_split_mfma(x.to(DEV), torch.from_numpy(w).to(DEV), torch.from_numpy(sh).to(DEV), cout=112, ca=64)
    a2, b2 = P.ops.fpn_level(x.to(DEV), None, wt.t().contiguous().to(DEV), bias.to(DEV), ca=64)
    assert tuple(a.shape) == (3, 19, 27, 64) and tuple(b.shape) == (3, 19, 27, 48)
    for got, want in ((a, ref[..., :64]), (b, ref[..., 64:]), (a2, ref[..., :64]), (b2, ref[..., 64:])):
        assert float((got.double().cpu() - want).abs().max() / want.abs().max()) < 1e-5


@pytest.mark.parametrize("mode", ["default", "fp32", "fp32_unfolded_fpn", "research_fp32"])
def test_featurenet_hip_matches_miopen(mode):
    """FeatureNet through the HIP kernels vs the same module on PyTorch-ROCm (MIOpen), all three pyramid levels: the default
    (fp16-split matrix-core convolutions, composed FPN head), the fp32 kernels (f16_split = False: pmn_stem + pmn_conv2d), those with
    the FPN head layer by layer in the reference's order, and -- research build only -- rounds 1-2's Winograd / fp32 MFMA kernels."""
    P = _gpu()
    g, params, kw = GU.load_case("default")
    model = _model(P, params, kw)
    model.feature.f16_split = mode == "default"
    model.feature.fold_fpn = mode != "fp32_unfolded_fpn"
    if mode == "research_fp32":
        _research()
        from patchmatchnet_amd import research
        research.install(model, winograd=True, winograd5=True, mfma_convs=True)
        model.feature.fpn8_valu = True
    x = torch.cat([t(g[f"image_{v}"]) for v in range(int(g["n_views"]))], 0)
    with torch.no_grad():
        ref = model.feature(x)
        got = model.feature.forward_hip(x)
    for s in (1, 2, 3):
        a, b = got[s].permute(0, 3, 1, 2), ref[s]
        assert a.shape == b.shape
        assert float((a - b).abs().max() / b.abs().max()) < 5e-5, s
    # and against the reference's own CPU FeatureNet outputs stored in the fixture
    for v in range(int(g["n_views"])):
        for s in (1, 2, 3):
            gold = g[f"feature_{v}_s{s}"]
            assert GU.abs_err(n(got[s][v:v + 1].permute(0, 3, 1, 2)), gold) / np.abs(gold).max() < 1e-4


def test_stage_projections_kernel():
    """pmn_stage_projections vs the reference's op sequence (net.py:225-231 + module.py:148) in float64."""
    P = _gpu()
    intr, extr = synth.synthetic_cameras(6, 1200, 1600)
    intr = np.repeat(intr, 2, 0)
    extr = np.repeat(extr, 2, 0)
    extr[1, :, :3, 3] += 5.0
    rel = n(P.ops.stage_projections(t(intr), t(extr), 3, 0.125))
    assert rel.shape == (3, 2, 5, 4, 4)
    for s, scale in enumerate((0.125, 0.25, 0.5)):
        proj = synth.stage_projections(intr, extr, scale).astype(np.float64)
        for b in range(2):
            for v in range(1, 6):
                want = proj[b, v] @ np.linalg.inv(proj[b, 0])
                assert np.abs(rel[s, b, v - 1] - want).max() / np.abs(want).max() < 2e-6


@pytest.mark.parametrize("mode", ["one_kernel", "two_kernels", "layers"])
@pytest.mark.parametrize("B,H,W", [(1, 96, 128), (2, 50, 70), (1, 38, 52)])
def test_refinement_hip_matches_miopen(mode, B, H, W):
    """Refinement through pmn_refine_fused (one launch, conv3 on the fp16 matrix cores with split operands: the default) or
    pmn_refine_front / pmn_refine_tail (f16_split = False: two fp32 launches) or one pmn_conv2d / pmn_deconv3x3s2 launch per layer
    in the reference's order (the verification switch research["layers"]) vs the same module on PyTorch-ROCm (MIOpen); ragged tiles,
    W % 4 != 0 (the scalar staging path) and batch > 1 included.  The one-launch form must agree with the two-launch form far
    inside the tolerance: they differ by conv3's split-fp16 rounding only."""
    P = _gpu()
    g, params, kw = GU.load_case("default")
    model = _model(P, params, kw)
    model.upsample_net.f16_split = mode == "one_kernel"
    model.upsample_net.research["layers"] = mode == "layers"
    gen = torch.Generator().manual_seed(5 + H)
    img = torch.rand(B, 3, H, W, generator=gen).to(DEV)
    d0 = (425.0 + 510.0 * torch.rand(B, 1, H // 2, W // 2, generator=gen)).to(DEV)
    dmin, dmax = t(np.full(B, 425.0, np.float32)), t(np.linspace(935.0, 1000.0, B).astype(np.float32))
    with torch.no_grad():
        ref = model.upsample_net(img, d0, dmin, dmax)
        got = model.upsample_net.forward_hip(img, d0, dmin, dmax)
        if mode == "one_kernel":
            model.upsample_net.f16_split = False
            two = model.upsample_net.forward_hip(img, d0, dmin, dmax)
            assert float(((got - two).abs() / two.abs()).max()) < 2e-6
    assert got.shape == ref.shape
    assert float(((got - ref).abs() / ref.abs()).max()) < 1e-5


def _rand_sample(n_views, H, W, B=1, seed=0):
    imgs = [im.repeat(B, 1, 1, 1).to(DEV) for im in synth.synthetic_images(n_views, H, W)]
    if B > 1:
        imgs = [torch.cat([im[:1], torch.flip(im[1:], dims=[3])], 0).contiguous() for im in imgs]
    intr, extr = synth.synthetic_cameras(n_views, H, W)
    return imgs, t(np.repeat(intr, B, 0)), t(np.repeat(extr, B, 0)), t(np.full(B, 425.0, np.float32)), \
        t(np.full(B, 935.0, np.float32))


def test_batch_of_two_equals_two_single_samples():
    """B=2 through the whole HIP forward (FeatureNet, PatchMatch, refinement) == the two samples run one by one."""
    P = _gpu()
    _, params, kw = GU.load_case("default")
    model = _model(P, params, kw)
    imgs, K, E, dmin, dmax = _rand_sample(4, 96, 128, B=2)
    noise = torch.rand(2, 48, 12, 16, generator=torch.Generator().manual_seed(9)).to(DEV)
    with torch.no_grad():
        d2, c2, _ = model([i.clone() for i in imgs], K.clone(), E, dmin, dmax, noise=noise)
        for b in range(2):
            d1, c1, _ = model([i[b:b + 1].clone() for i in imgs], K[b:b + 1].clone(), E[b:b + 1], dmin[b:b + 1],
                              dmax[b:b + 1], noise=noise[b:b + 1])
            assert torch.equal(d1[0], d2[b]) and torch.equal(c1[0], c2[b])


def test_config0_one_iteration_per_stage_against_oracle():
    """BASELINE configs[0]'s shape and schedule (160x128, 2 source views, ONE PatchMatch iteration per stage) through the HIP
    cascade, against the oracle on the same features and noise (the oracle itself is pinned to the reference at this size in
    tests/test_oracle_vs_reference.py).  With one iteration at stage 1 the reference skips propagation there and regresses in
    inverse depth (patchmatch.py:465, :482)."""
    P = _gpu()
    _, params, kw = GU.load_case("default")
    kw = dict(kw, patchmatch_iteration=[1, 1, 1])
    model = _model(P, params, kw)
    H, W = 128, 160
    imgs, K, E, dmin, dmax = _rand_sample(3, H, W)
    noise = torch.rand(1, 48, H // 8, W // 8, generator=torch.Generator().manual_seed(21))
    with torch.no_grad():
        feats = model.extract_features(imgs)
        depth, conf, dpm = model([i.clone() for i in imgs], K.clone(), E, dmin, dmax, noise=noise.to(DEV), features=feats)
    feats_np = [{s: np.ascontiguousarray(n(f[s])) for s in (1, 2, 3)} for f in feats]
    d1, score, out = O.cascade(params, feats_np, n(K), n(E), n(dmin), n(dmax), noise.numpy(), configs=_configs(kw))
    for stage in (3, 2, 1):
        assert len(dpm[stage]) == 1
        assert GU.rel_err(n(dpm[stage][0]), out[stage][0]) < 1e-3, stage
    assert GU.rel_err(n(dpm[1][0]), d1) < 1e-3
    c, _ = O.confidence(score, (H, W))
    assert float((np.abs(n(conf) - c) > 1e-3).mean()) < 5e-3
    assert depth.shape == (1, 1, H, W) and bool(torch.isfinite(depth).all())


def test_single_source_view_and_argument_errors():
    """Smallest problem the reference accepts (one source view) against the oracle, and the reference's own argument checks
    (models/net.py:196-197: one intrinsic / extrinsic matrix per image)."""
    P = _gpu()
    _, params, kw = GU.load_case("default")
    model = _model(P, params, kw)
    H, W = 64, 96
    imgs, K, E, dmin, dmax = _rand_sample(2, H, W)
    noise = torch.rand(1, 48, H // 8, W // 8, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        feats = model.extract_features(imgs)
        depth, conf, dpm = model([i.clone() for i in imgs], K.clone(), E, dmin, dmax, noise=noise.to(DEV), features=feats)
    feats_np = [{s: np.ascontiguousarray(n(f[s])) for s in (1, 2, 3)} for f in feats]
    d1, _, out = O.cascade(params, feats_np, n(K), n(E), n(dmin), n(dmax), noise.numpy(), configs=_configs(kw))
    assert GU.rel_err(n(dpm[3][0]), out[3][0]) < 1e-3 and GU.rel_err(n(dpm[1][-1]), d1) < 1e-3
    assert conf.shape == (1, H, W) and bool(torch.isfinite(depth).all())
    with pytest.raises(AssertionError, match="Different number of images and intrinsic matrices"):
        model(imgs, K[:, :1], E, dmin, dmax)
    with pytest.raises(AssertionError, match="Different number of images and extrinsic matrices"):
        model(imgs, K, E[:, :1], dmin, dmax)


def test_sizes_not_multiple_of_8_are_resized_like_the_reference():
    """adjust_image_dims (reference net.py:304-318): inputs are stretched to multiples of 8, intrinsics rescaled IN
    PLACE, outputs come back at the original size."""
    P = _gpu()
    _, params, kw = GU.load_case("default")
    model = _model(P, params, kw)
    imgs, K, E, dmin, dmax = _rand_sample(3, 100, 130)
    K0 = K.clone()
    with torch.no_grad():
        depth, conf, dpm = model(imgs, K, E, dmin, dmax)
    assert depth.shape == (1, 1, 100, 130) and conf.shape == (1, 100, 130)
    # 100/8 = 12.5 rounds to 12 (Python's round-half-even, as in the reference) -> 96; 130/8 = 16.25 -> 16 -> 128
    assert imgs[0].shape[-2:] == (96, 128)  # the caller's list now holds the resized images, as with the reference
    assert torch.allclose(K[:, :, 0], K0[:, :, 0] * (128 / 130)) and torch.allclose(K[:, :, 1], K0[:, :, 1] * (96 / 100))
    assert bool(torch.isfinite(depth).all()) and dpm[1][-1].shape == (1, 1, 48, 64)


@pytest.mark.parametrize("H,W,n_src", [(1056, 1920, 7), (2048, 3072, 10)])
def test_large_configs_run_and_are_sane(H, W, n_src):
    """BASELINE configs[2] (Tanks&Temples 1920x1056, N=7) and configs[4] (ETH3D 3072x2048, N=10) through the full forward:
    finite outputs inside the depth range, probabilities behind the confidence in [0,1], bit-identical reruns."""
    P = _gpu()
    _, params, kw = GU.load_case("default")
    model = _model(P, params, kw)
    imgs, K, E, dmin, dmax = _rand_sample(n_src + 1, H, W)
    noise = torch.rand(1, 48, H // 8, W // 8, generator=torch.Generator().manual_seed(3)).to(DEV)
    with torch.no_grad():
        d1, c1, dpm = model(list(imgs), K.clone(), E, dmin, dmax, noise=noise)
        d2, c2, _ = model(list(imgs), K.clone(), E, dmin, dmax, noise=noise)
    torch.cuda.synchronize()
    assert d1.shape == (1, 1, H, W) and c1.shape == (1, H, W)
    assert torch.equal(d1, d2) and torch.equal(c1, c2)
    assert bool(torch.isfinite(d1).all()) and bool(torch.isfinite(c1).all())
    for s in (3, 2, 1):
        for d in dpm[s]:
            assert float(d.min()) >= 425.0 * (1 - 1e-5) and float(d.max()) <= 935.0 * (1 + 1e-5)
    assert float(c1.min()) >= 0.0 and float(c1.max()) <= 1.0 + 1e-5


# ---- BASELINE-size checks -------------------------------------------------------------------------------------------

def _fullsize_stage(P, stage, n_src, H, W, params, kw, seed=0):
    scale = {3: 8, 2: 4, 1: 2}[stage]
    C = {3: 64, 2: 32, 1: 16}[stage]
    h, w = H // scale, W // scale
    feats = synth.synthetic_features(n_src + 1, C, h, w, seed=seed)
    intr, extr = synth.synthetic_cameras(n_src + 1, H, W)
    proj = synth.stage_projections(intr, extr, 1.0 / scale)
    return feats, proj, h, w


@pytest.mark.parametrize("stage,n_src,H,W", [(3, 5, 1200, 1600), (1, 5, 1200, 1600), (2, 7, 1056, 1920)])
def test_fullsize_stage_against_oracle(stage, n_src, H, W):
    """One PatchMatch stage at BASELINE config sizes (cfg-2: 1600x1200 N=5; cfg-3: 1920x1056 N=7) vs the CPU oracle on
    identical inputs (same conv offsets), plus size-independent properties."""
    P = _gpu()
    _, params, kw = GU.load_case("default")
    cfg = _configs(kw)[stage]
    model = _model(P, params, kw)
    pm = getattr(model, f"patchmatch_{stage}")
    feats, proj, h, w = _fullsize_stage(P, stage, n_src, H, W, params, kw)
    dmin, dmax = np.array([425.0], np.float32), np.array([935.0], np.float32)
    gen = torch.Generator().manual_seed(1234)
    noise = torch.rand(1, 48, h, w, generator=gen)
    if stage == 3:
        depth, vw = None, None
    else:
        depth = (425.0 + 510.0 * torch.rand(1, 1, h, w, generator=gen)).numpy()
        vw = torch.rand(1, n_src, h, w, generator=gen).numpy()
    dbg = []
    with torch.no_grad():
        out = pm(ref_feature=feats[0].to(DEV), src_features=[f.to(DEV) for f in feats[1:]], ref_proj=t(proj[:, 0]),
                 src_projs=[t(proj[:, i]) for i in range(1, proj.shape[1])], depth_min=t(dmin), depth_max=t(dmax),
                 depth=torch.empty(0, device=DEV) if depth is None else t(depth),
                 view_weights=torch.empty(0, device=DEV) if vw is None else t(vw), noise=noise.to(DEV), debug=dbg)
    torch.cuda.synchronize()
    # --- properties that hold at any size
    for rec in dbg:
        prob = rec["score"]
        assert torch.allclose(prob.sum(1), torch.ones_like(prob.sum(1)), atol=1e-5)
        hyp = rec["depth_sample"]
        assert float(hyp.min()) >= 425.0 * (1 - 1e-6) and float(hyp.max()) <= 935.0 * (1 + 1e-6)
        if rec["propa_offsets"] is not None and not (stage == 1):
            assert bool((hyp[:, 1:] >= hyp[:, :-1]).all()), "propagated hypotheses must be sorted ascending"
        d = rec["depth"]
        assert float(d.min()) >= 425.0 * (1 - 1e-5) and float(d.max()) <= 935.0 * (1 + 1e-5)
        assert float(rec["view_weights"].min()) >= 0.0 and float(rec["view_weights"].max()) <= 1.0
    # determinism: a second run is bit-identical
    with torch.no_grad():
        out2 = pm(ref_feature=feats[0].to(DEV), src_features=[f.to(DEV) for f in feats[1:]], ref_proj=t(proj[:, 0]),
                  src_projs=[t(proj[:, i]) for i in range(1, proj.shape[1])], depth_min=t(dmin), depth_max=t(dmax),
                  depth=torch.empty(0, device=DEV) if depth is None else t(depth),
                  view_weights=torch.empty(0, device=DEV) if vw is None else t(vw), noise=noise.to(DEV))
    assert all(torch.equal(a, b) for a, b in zip(out[0], out2[0])) and torch.equal(out[1], out2[1])
    # --- oracle on the same inputs (offsets taken from the GPU conv so both sides sample the same neighbours)
    otr = []
    O.patchmatch_stage(cfg, params, feats[0].numpy(), [f.numpy() for f in feats[1:]], proj[:, 0],
                       [proj[:, i] for i in range(1, proj.shape[1])], dmin, dmax, depth, vw, noise=noise.numpy(),
                       propa_offsets=None if dbg[0]["propa_offsets"] is None else n(dbg[0]["propa_offsets"]),
                       eval_offsets=n(dbg[0]["eval_offsets"]), trace=otr)
    for it, (rec, orec) in enumerate(zip(dbg, otr)):
        if it == 0:
            assert GU.rel_err(n(rec["depth_sample"]), orec["depth_sample"]) < 2e-6
            assert GU.abs_err(n(rec["similarity"]), orec["similarity"]) < 1e-4
            if stage == 3:
                # arg-max over D of the PixelwiseNet response: identical to the oracle's except at fp32 near-ties
                # (two hypotheses whose responses agree to ~1 ulp), where either index yields the same weight
                bad = n(rec["view_weight_argmax"]) != orec["view_weight_argmax"]
                assert float(bad.mean()) < 1e-4, float(bad.mean())
                if bad.any():  # a flipped index is legitimate only where the two responses are (nearly) tied
                    assert GU.abs_err(n(rec["view_weights"])[bad], orec["view_weights"][bad]) < 1e-4
                assert GU.abs_err(n(rec["view_weights"]), orec["view_weights"]) < 1e-4
        rel = np.abs(n(rec["depth"]) - orec["depth"]) / orec["depth"]
        assert rel.max() < 1e-3, (it, float(rel.max()))


@pytest.mark.parametrize("stage", [3, 2])
def test_source_views_of_different_sizes_against_oracle(stage):
    """A sample whose source images differ in size from the reference image and from each other -- legal in the reference
    (models/module.py:130-181 warps every view at its own size; models/net.py:304-318 rounds every image to multiples of 8 on its
    own) and refused by rounds 1-2.  The source maps sit zero-padded in one buffer and the projection rows carry the per-view scale
    (ops.stack_sources_padded / rescale_projection_rows); checked against the oracle, which walks every view at its own size."""
    P = _gpu()
    _, params, kw = GU.load_case("default")
    m = P.PatchmatchNet(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.to(DEV).eval()
    cfg = O.default_stage_configs(kw["patchmatch_interval_scale"], kw["propagation_range"], kw["patchmatch_iteration"],
                                  kw["patchmatch_num_sample"], kw["propagate_neighbors"], kw["evaluate_neighbors"])[stage]
    pm = getattr(m, f"patchmatch_{stage}")
    C = {3: 64, 2: 32}[stage]
    h, w = 40, 56
    sizes = [(40, 56), (32, 48), (48, 72)]
    gen = torch.Generator().manual_seed(stage)
    base = torch.nn.functional.avg_pool2d(0.5 * torch.randn(1, C, 64, 96, generator=gen), 5, 1, 2) * 3.0
    ref = base[:, :, 8:8 + h, 16:16 + w].contiguous()
    srcs = [torch.nn.functional.interpolate(base[:, :, 8:8 + h, 16 + 2 * i:16 + 2 * i + w], size=sz, mode="bilinear",
                                            align_corners=True).contiguous() for i, sz in enumerate(sizes)]
    # cameras: the synthetic rig at the REFERENCE map's resolution; a source map of another size has its intrinsics scaled to it
    intr, extr = synth.synthetic_cameras(4, h, w)
    proj = synth.stage_projections(intr, extr, 1.0)  # [1,4,4,4]
    src_projs = []
    for v, (hv, wv) in enumerate(sizes):
        K = intr[0, v + 1].astype(np.float32).copy()
        K[0] *= wv / w
        K[1] *= hv / h
        pr = extr[0, v + 1].astype(np.float32).copy()
        pr[:3, :4] = K @ extr[0, v + 1, :3, :4].astype(np.float32)
        src_projs.append(pr[None])
    dmin, dmax = np.array([425.0], np.float32), np.array([935.0], np.float32)
    noise = torch.rand(1, 48, h, w, generator=gen)
    if stage == 3:
        depth, vw = None, None
    else:
        yy, xx = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
        depth = (680.0 + 100.0 * torch.sin(xx / 11.0) * torch.cos(yy / 9.0)).clamp(425.0, 935.0)[None, None].numpy()
        vw = torch.rand(1, 3, h, w, generator=gen).numpy()
    dbg = []
    with torch.no_grad():
        pm(ref_feature=ref.to(DEV), src_features=[f.to(DEV) for f in srcs], ref_proj=t(proj[:, 0]), src_projs=[t(p_) for p_ in src_projs],
           depth_min=t(dmin), depth_max=t(dmax), depth=torch.empty(0, device=DEV) if depth is None else t(depth),
           view_weights=torch.empty(0, device=DEV) if vw is None else t(vw), noise=noise.to(DEV), debug=dbg)
    torch.cuda.synchronize()
    import copy
    one = copy.copy(cfg)
    one.iterations = 1
    otr = []
    O.patchmatch_stage(one, params, ref.numpy(), [f.numpy() for f in srcs], proj[:, 0], src_projs, dmin, dmax, depth, vw,
                       noise=noise.numpy(), propa_offsets=None if dbg[0]["propa_offsets"] is None else n(dbg[0]["propa_offsets"]),
                       eval_offsets=n(dbg[0]["eval_offsets"]), trace=otr)
    assert GU.abs_err(n(dbg[0]["similarity"]), otr[0]["similarity"]) < 1e-4
    if stage == 3:
        assert GU.abs_err(n(dbg[0]["view_weights"]), otr[0]["view_weights"]) < 1e-4
    rel = np.abs(n(dbg[0]["depth"]) - otr[0]["depth"]) / otr[0]["depth"]
    assert rel.max() < 1e-3, float(rel.max())


def test_forward_with_images_of_different_sizes_against_the_reference():
    """PatchmatchNet.forward on a sample whose source images are smaller / larger than the reference image, against the REFERENCE's
    own output on the same sample (tests/golden/cascade_mixed_sizes.npz, make_golden.py --only mixed; reference models/net.py:176-301
    with per-view FeatureNet passes and per-view warps)."""
    P = _gpu()
    g, params, kw = GU.load_npz("cascade_mixed_sizes.npz"), GU.load_npz("params_000007.npz"), GU.CASES["default"][2]
    # the inputs regenerate from seeds (same generator as the golden script, restated here: no reference import on the GPU box)
    sizes = [(96, 128), (80, 112), (96, 144)]
    intr, extr = synth.synthetic_cameras(3, 96, 128)
    intr = intr.copy()
    imgs = []
    for v, (H, W) in enumerate(sizes):
        imgs.append(synth.synthetic_images(3, H, W)[v].to(DEV))
        intr[0, v, 0] *= W / 128
        intr[0, v, 1] *= H / 96
    noise = torch.rand(1, 48, 12, 16, generator=torch.Generator().manual_seed(77)).to(DEV)
    m = P.PatchmatchNet(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.to(DEV).eval()
    with torch.no_grad():
        depth, conf, dpm = m(imgs, t(intr), t(extr), t(np.array([425.0], np.float32)), t(np.array([935.0], np.float32)), noise=noise)
    assert tuple(depth.shape) == (1, 1, 96, 128)
    for st in (3, 2, 1):
        for it, d in enumerate(dpm[st]):
            rel = np.abs(n(d) - g[f"s{st}_it{it + 1}_depth_out"]) / g[f"s{st}_it{it + 1}_depth_out"]
            assert np.quantile(rel, 0.999) < 1e-3 and rel.max() < 2e-2, (st, it, float(rel.max()))
    rel = np.abs(n(depth) - g["depth"]) / g["depth"]
    assert float(np.quantile(rel, 0.999)) < 1e-3, float(np.quantile(rel, 0.999))
    assert float((np.abs(n(conf) - g["confidence"]) > 1e-3).mean()) < 1e-2


# ---- round 4: the fp16-split kernels' accepted range, and the configurations they do not cover ------------------------------

def test_f16_split_domain():
    """include/pmn_hip.h, "fp16-split entry points": what pmn_conv2d_f16s does at both ends of float16's range.
      * activations of 6e4 (just inside): still the error of an fp32 convolution;
      * activations of 1e-7 (below fp16's normal range) next to weights of normal size: ABSOLUTE error far below the output's
        own scale (the subnormal hi keeps ~3e-8 absolute precision per factor);
      * activations of 7e4 (outside): hi = inf -- the output is non-finite where the fp32 kernel is finite.  Documented, not
        guarded in the kernel; weights outside the range are refused when they are packed (next test)."""
    P = _gpu()
    from patchmatchnet_amd import params as PP
    gen = torch.Generator().manual_seed(3)
    cin = cout = 16
    wt = 0.2 * torch.randn(cout, cin, 3, 3, generator=gen)
    w, sh = PP.pack_conv_f16s(wt)
    w2, s2 = PP.pack_conv(wt)
    wd, sd, w2d, s2d = (torch.from_numpy(a).to(DEV) for a in (w, sh, w2, s2))
    base = torch.randn(1, 20, 24, cin, generator=gen)
    for scale, expect_finite in ((6.0e4 / 4.5, True), (1.0e-7, True), (7.0e4, False)):
        x = (base * scale if expect_finite else torch.full_like(base, scale)).to(DEV)
        got = P.ops.conv2d_f16s(x, wd, sd, 3, 1, relu=False)
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), wt.double(), None, 1, 1).permute(0, 2, 3, 1)
        fp32 = P.ops.conv2d(x, w2d, s2d, cout, 3, 1, 1, relu=False)
        assert bool(torch.isfinite(fp32).all())
        if expect_finite:
            assert bool(torch.isfinite(got).all()), scale
            err = float((got.double().cpu() - ref).abs().max() / ref.abs().max())
            assert err < (2e-6 if scale > 1 else 2e-3), (scale, err)  # 1e-7: relative to an output of ~1e-7 the 3e-8-level
            #                                                            absolute error of the subnormal hi parts shows
            assert float((got.double().cpu() - ref).abs().max()) < max(1e-6 * float(ref.abs().max()), 1e-9)
        else:
            assert not bool(torch.isfinite(got).all())  # hi = fp16(7e4) = inf


@pytest.mark.parametrize("N,H,W", [(2, 37, 53), (1, 28, 14), (3, 14, 45), (1, 600, 800)])
def test_conv2d_f16s_pair_is_two_single_layers_bit_for_bit(N, H, W):
    """pmn_conv2d_f16s_pair (FeatureNet conv3 + conv4 in one launch, the intermediate map in LDS; round 6) against two
    pmn_conv2d_f16s launches: the same bits at ragged sizes (tiles of 14 x 14 cut by the image border: the second layer pads the first
    layer's OUTPUT with zeros, not its input), several images, and the benchmark's half resolution."""
    P = _gpu()
    from patchmatchnet_amd import params as PP
    gen = torch.Generator().manual_seed(11)
    packs = []
    for _ in range(2):
        wt = 0.15 * torch.randn(16, 16, 3, 3, generator=gen)
        bn = (torch.rand(16, generator=gen) + 0.5, 0.1 * torch.randn(16, generator=gen), 0.1 * torch.randn(16, generator=gen),
              torch.rand(16, generator=gen) + 0.5)
        w, sh = PP.pack_conv_f16s(wt, bn=bn)
        packs.append((torch.from_numpy(w).to(DEV), torch.from_numpy(sh).to(DEV)))
    x = torch.randn(N, H, W, 16, generator=gen).to(DEV)
    want = P.ops.conv2d_f16s(P.ops.conv2d_f16s(x, *packs[0], 3, 1, relu=True), *packs[1], 3, 1, relu=True)
    got = P.ops.conv2d_f16s_pair(x, *packs[0], *packs[1], relu=True)
    assert got.shape == want.shape and torch.equal(got, want), float((got - want).abs().max())
    want = P.ops.conv2d_f16s(P.ops.conv2d_f16s(x, *packs[0], 3, 1, relu=False), *packs[1], 3, 1, relu=False)
    got = P.ops.conv2d_f16s_pair(x, *packs[0], *packs[1], relu=False)
    assert torch.equal(got, want)


def test_featurenet_fused_conv34_is_the_unfused_network():
    P = _gpu()
    g, params, kw = GU.load_case("default")
    model = _model(P, params, kw)
    x = torch.cat([t(g[f"image_{v}"]) for v in range(int(g["n_views"]))], 0)
    with torch.no_grad():
        assert model.feature.fuse_conv34
        a = model.feature.forward_hip(x)
        model.feature.fuse_conv34 = False
        b = model.feature.forward_hip(x)
    for s in (1, 2, 3):
        assert torch.equal(a[s], b[s]), s


def test_f16_domain_check_is_opt_in_and_catches_a_scaled_model():
    """PMN_CHECK_F16_DOMAIN=1: every input of an fp16-split entry point is scanned by pmn_check_f16_domain and
    ops.f16_domain_check() raises when one was not finite or >= 65504 in magnitude -- here a checkpoint whose first layer is scaled
    so that FeatureNet's activations leave float16's range (every weight still packs: the weights' own check cannot see it).  Off by
    default: the product forward launches no check kernel."""
    P = _gpu()
    g, params, kw = GU.load_case("default")
    model = _model(P, params, kw)
    x = torch.cat([t(g[f"image_{v}"]) for v in range(int(g["n_views"]))], 0)
    ops = P.ops
    assert ops.F16_DOMAIN_CHECK is False  # the suite runs without the variable
    ops.F16_DOMAIN_CHECK = True
    try:
        with torch.no_grad():
            model.feature.forward_hip(x)
            ops.f16_domain_check()  # the released checkpoint on images in [0, 1]: inside the domain
            model.feature.forward_hip(x * 3.0e6)  # activations ~1e6 x the usual few units
            with pytest.raises(P.PmnError, match="65504"):
                ops.f16_domain_check()
            ops.f16_domain_check()  # the flag was reset
            bad = torch.full((1, 8, 8, 16), float("nan"), device=DEV)
            ops._f16_domain_probe(bad)
            with pytest.raises(P.PmnError, match="65504"):
                ops.f16_domain_check()
    finally:
        ops.F16_DOMAIN_CHECK = False


def test_f16_split_weights_outside_the_range_fall_back_to_fp32():
    """A BatchNorm running_var small enough to fold conv3's weights beyond 65504: params refuses to pack them, FeatureNet says so
    (RuntimeWarning + f16_domain_error) and runs the fp32 kernels -- finite output equal to the MIOpen module's."""
    P = _gpu()
    from patchmatchnet_amd import params as PP
    with pytest.raises(PP.F16DomainError):
        PP.split_f16(np.array([1.0, 7.0e4]))
    g, params, kw = GU.load_case("default")
    model = _model(P, params, kw)
    with torch.no_grad():
        model.feature.conv3.bn.running_var.fill_(1e-12)  # scale = gamma / sqrt(var + eps) with eps 1e-5 ... and a huge gamma
        model.feature.conv3.bn.weight.fill_(3.0e5)
        model.feature.conv3.bn.bias.zero_()
    x = t(g["image_0"])
    with pytest.warns(RuntimeWarning, match="fp32 kernels"):
        got = model.feature.forward_hip(x)
    assert model.feature.f16_domain_error is not None and "conv3" in model.feature.f16_domain_error
    with torch.no_grad():
        ref = model.feature(x)
    for s_ in (1, 2, 3):
        a, b = got[s_].permute(0, 3, 1, 2), ref[s_]
        assert bool(torch.isfinite(a).all())
        assert float((a - b).abs().max() / b.abs().max()) < 1e-4, s_


@pytest.mark.parametrize("n_p,n_e", [(16, 17), (4, 9)])
def test_offset_heads_outside_the_f16_split_instantiations(n_p, n_e):
    """pmn_offset_heads_f16s instantiates padded row counts 32, 48 and 64.  16 propagation + 17 evaluation neighbours give 66 rows
    (80 padded): PatchMatch must not pack / call it for that stage and take pmn_conv2d instead (round 3 packed anything up to 64 rows
    and would have handed a 16-row padding to the kernel: PMN_ERR_SHAPE); 4 + 9 = 26 rows pads to 32 and stays on the matrix cores.
    Either way the whole forward must agree with the MIOpen heads."""
    P = _gpu()
    kw = dict(patchmatch_interval_scale=[0.005, 0.0125, 0.025], propagation_range=[6, 4, 2], patchmatch_iteration=[1, 2, 2],
              patchmatch_num_sample=[8, 8, 16], propagate_neighbors=[0, n_p, n_p], evaluate_neighbors=[n_e, n_e, n_e])
    torch.manual_seed(0)
    model = P.PatchmatchNet(**kw).to(DEV).eval()
    with torch.no_grad():
        for pmx in (model.patchmatch_1, model.patchmatch_2, model.patchmatch_3):  # non-trivial heads (the reference initialises them to 0)
            for m in (pmx.propa_conv, pmx.eval_conv):
                m.weight.normal_(0, 0.02)
                m.bias.normal_(0, 0.1)
    imgs, K, E, dmin, dmax = _rand_sample(3, 64, 96)
    noise = torch.rand(1, 48, 8, 12, generator=torch.Generator().manual_seed(1)).to(DEV)
    with torch.no_grad():
        d1, c1, _ = model(list(imgs), K.clone(), E, dmin, dmax, noise=noise)
        for pmx in (model.patchmatch_2, model.patchmatch_3):
            pk = pmx._packed_heads()
            assert ("f16s_both" in pk) == (2 * (n_p + n_e) <= 64), sorted(pk)
        for pmx in (model.patchmatch_1, model.patchmatch_2, model.patchmatch_3):
            pmx.hip_offset_heads = False
        d2, c2, _ = model(list(imgs), K.clone(), E, dmin, dmax, noise=noise)
    assert bool(torch.isfinite(d1).all())
    rel = ((d1 - d2).abs() / d2.abs()).flatten()
    assert float(rel.median()) < 1e-5 and float((rel > 1e-3).float().mean()) < 2e-2


def test_integration_stub_warps_like_the_op():
    """INTEGRATION.md's ctypes stub of differentiable_warping, executed as written, against the golden known answer."""
    P = _gpu()
    import test_abi
    ns = {}
    exec(compile(test_abi._integration_stub(), "INTEGRATION.md", "exec"), ns)
    g = GU.load_npz("ops_small.npz")
    got = ns["differentiable_warping"](t(g["A_src"]), t(g["A_src_proj"]), t(g["A_ref_proj"]), t(g["A_depth"]))
    torch.cuda.synchronize()
    assert GU.abs_err(n(got), g["A_warped"]) < 5e-5


def test_warp_correlate_views_reads_the_sources_in_place():
    """pmn_warp_correlate_views (source views through a device table of per-view addresses) == pmn_warp_correlate on the stacked
    tensor, bit for bit, known weights and PixelwiseNet, batch of two."""
    P = _gpu()
    from patchmatchnet_amd import _lib
    gen = torch.Generator().manual_seed(4)
    B, N, C, G, D, h, w = 2, 3, 32, 8, 16, 21, 36
    intr, extr = synth.synthetic_cameras(N + 1, h * 8, w * 8)
    proj = synth.stage_projections(intr, extr, 0.125)
    P0 = torch.from_numpy(proj[0, 0]).double()
    rel = torch.stack([torch.from_numpy(proj[0, i]).double() @ torch.inverse(P0) for i in range(1, N + 1)], 0).float()
    rel = rel[None].repeat(B, 1, 1, 1).contiguous().to(DEV)
    ref = torch.randn(B, h, w, C, generator=gen).to(DEV)
    views = [torch.randn(B, h, w, C, generator=gen).to(DEV) for _ in range(N)]  # separately allocated, as a feature cache holds them
    stacked = torch.stack(views, 0).contiguous()
    lo, hi = 1 / 935.0, 1 / 425.0
    depth = (1.0 / (lo + torch.rand(B, D, h, w, generator=gen) * (hi - lo))).sort(dim=1)[0].contiguous().to(DEV)
    mlp = lambda s: (0.4 * torch.randn(_lib.MLP_FLOATS, generator=torch.Generator().manual_seed(s))).to(DEV)
    table = P.ops.SourceTable(torch.tensor(P.ops.SourceTable.addresses(views), dtype=torch.int64, device=DEV), stacked.shape)
    for vw in (torch.rand(B, N, h, w, generator=gen).to(DEV), None):
        a = P.ops.warp_correlate(ref, stacked, rel, depth, vw, 0, mlp(1), None if vw is not None else mlp(2), G, want_similarity=True)
        b = P.ops.warp_correlate(ref, table, rel, depth, vw, 0, mlp(1), None if vw is not None else mlp(2), G, want_similarity=True)
        torch.cuda.synchronize()
        assert torch.equal(a[0], b[0]) and torch.equal(a[3], b[3]) and torch.equal(a[1], b[1])


def test_confidence_2x_path_equals_the_general_kernel():
    """pmn_confidence at H = 2h, W = 2w (one thread per source pixel, 2 x 2 outputs each) against the general nearest-resize kernel,
    which the same call takes when the output is not 8-byte aligned; and the general kernel at a non-2x size stays reachable."""
    P = _gpu()
    from patchmatchnet_amd import _lib, ops
    g = torch.Generator().manual_seed(8)
    for (B, D, h, w) in ((1, 8, 37, 50), (2, 8, 16, 24), (1, 16, 9, 11)):
        score = torch.softmax(4 * torch.randn(B, D, h, w, generator=g), 1).to(DEV).contiguous()
        fast, idx_fast = ops.confidence(score, 2 * h, 2 * w, want_index=True)
        buf = torch.empty(B * 4 * h * w + 1, device=DEV)
        slow = buf[1:].view(B, 2 * h, 2 * w)
        assert slow.data_ptr() % 8 == 4
        idx_slow = torch.empty(B, h, w, dtype=torch.int32, device=DEV)
        _lib.check(_lib.lib().pmn_confidence(score.data_ptr(), B, D, h, w, 2 * h, 2 * w, slow.data_ptr(), idx_slow.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream), "pmn_confidence")
        torch.cuda.synchronize()
        assert torch.equal(fast, slow) and torch.equal(idx_fast, idx_slow)
        assert bool((fast[:, ::2, ::2] == fast[:, 1::2, 1::2]).all())
        odd, _ = ops.confidence(score, 2 * h + 3, 2 * w - 1)
        assert odd.shape == (B, 2 * h + 3, 2 * w - 1) and bool(torch.isfinite(odd).all())


def test_forward_on_a_size_the_reference_adjusts_against_the_reference():
    """100 x 130 images (not multiples of 8) against the REFERENCE's own output on the same sample (tests/golden/
    cascade_resized_100x130.npz, make_golden.py --only resized): adjust_image_dims stretches to 96 x 128 and rescales the caller's
    intrinsics in place (reference models/net.py:304-318), the final depth comes back bilinearly at 100 x 130 and the confidence by the
    GENERAL nearest-resize kernel (the cascade's 2x fast path does not apply)."""
    P = _gpu()
    g, params, kw = GU.load_npz("cascade_resized_100x130.npz"), GU.load_npz("params_000007.npz"), GU.CASES["default"][2]
    H, W = 100, 130
    imgs = [im.to(DEV) for im in synth.synthetic_images(3, H, W)]
    intr, extr = synth.synthetic_cameras(3, H, W)
    noise = torch.rand(1, 48, 12, 16, generator=torch.Generator().manual_seed(55)).to(DEV)
    m = P.PatchmatchNet(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.to(DEV).eval()
    K = t(intr.copy())
    with torch.no_grad():
        depth, conf, dpm = m(imgs, K, t(extr), t(np.array([425.0], np.float32)), t(np.array([935.0], np.float32)), noise=noise)
    assert tuple(depth.shape) == (1, 1, H, W) and tuple(conf.shape) == (1, H, W)
    np.testing.assert_array_equal(n(K), g["intrinsics_after"])  # the caller's matrices, mutated exactly as the reference mutates them
    for st in (3, 2, 1):
        for it, d in enumerate(dpm[st]):
            rel = np.abs(n(d) - g[f"s{st}_it{it + 1}_depth_out"]) / g[f"s{st}_it{it + 1}_depth_out"]
            assert np.quantile(rel, 0.999) < 1e-3 and rel.max() < 2e-2, (st, it, float(rel.max()))
    rel = np.abs(n(depth) - g["depth"]) / g["depth"]
    assert float(np.quantile(rel, 0.999)) < 1e-3, float(np.quantile(rel, 0.999))
    assert float((np.abs(n(conf) - g["confidence"]) > 1e-3).mean()) < 1e-2
