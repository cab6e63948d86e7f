"""Parity on the configurations the numbers are quoted on (BASELINE.json configs 2, 3 and 5), all through the C ABI.

* cfg-2 (1600x1200, N=5, iters 1,2,2): the WHOLE cascade chained from the HIP FeatureNet's features on bench.py's sample
  vs the CPU oracle on the same features and noise -- every stage / iteration depth <= 1e-3 relative (north_star),
  view-weight arg-max indices and the integer confidence index equal off fp32 ties (reference models/net.py:221-301,
  :288-299; models/patchmatch.py:695-702).
* cfg-3 (1920x1056, N=7): all three stages; cfg-5 (3072x2048, N=10): stages 3 and 1 -- each vs the oracle on identical
  inputs (reference models/patchmatch.py:428-529).
* FeatureNet through the HIP convolutions vs the same module on MIOpen at 6x1600x1200 and 1x3072x2048.
Measured maxima are appended to gpurun_out/parity_report.jsonl (copied into profiles/ as evidence).
"""
import json
import os

import numpy as np
import pytest
import torch

import goldenutil as GU
import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpu():
    assert torch.cuda.is_available(), "GPU tests selected but no ROCm device is visible"
    import patchmatchnet_amd as P
    P.lib()
    return P


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def n(x):
    return x.detach().cpu().numpy()


def _report(**kw):
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "parity_report.jsonl"), "a") as f:
        f.write(json.dumps(kw) + "\n")


def _model(P):
    _, params, kw = GU.load_case("default")
    m = P.PatchmatchNet(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return m.to(DEV).eval(), params, kw


def _configs(kw):
    return O.default_stage_configs(kw["patchmatch_interval_scale"], kw["propagation_range"], kw["patchmatch_iteration"],
                                   kw["patchmatch_num_sample"], kw["propagate_neighbors"], kw["evaluate_neighbors"])


def _argmax_check(got_idx, got_vw, want_idx, want_vw, tie=1e-4, want_resp=None, report=None):
    """Arg-max over D of the PixelwiseNet response: exact except at fp32 near-ties, where either index gives the same weight.
    ``want_resp`` [B,N,D,h,w] = the oracle's own per-hypothesis responses: at every mismatch the gap between the oracle's response at
    ITS index and at the index this engine chose is then measured in ulps of the response (the north star's "bit-exact view_weights
    indices" can only fail where two hypotheses tie at fp32 resolution) and put into ``report``."""
    bad = got_idx != want_idx
    frac = float(bad.mean())
    assert frac < 5e-4, frac  # measured 1.1e-4 of (pixel, view) pairs at cfg-2 (150 K pairs x 64 hypotheses)
    if want_resp is not None and bad.any():
        b_, v_, y_, x_ = np.nonzero(bad)
        r_own = want_resp[b_, v_, want_idx[bad].astype(np.int64), y_, x_].astype(np.float32)
        r_eng = want_resp[b_, v_, got_idx[bad].astype(np.int64), y_, x_].astype(np.float32)
        ulp = np.spacing(np.abs(r_own))
        gap = (r_own.astype(np.float64) - r_eng.astype(np.float64)) / ulp
        assert (gap >= 0).all()
        hist = {str(k): int((np.round(gap) == k).sum()) for k in range(0, 9)}
        hist[">8"] = int((np.round(gap) > 8).sum())
        if report is not None:
            report["view_weight_argmax_mismatches"] = int(bad.sum())
            report["view_weight_argmax_pairs"] = int(bad.size)
            report["argmax_mismatch_gap_in_ulps_of_the_oracle_response_max"] = float(gap.max())
            report["argmax_mismatch_gap_in_ulps_histogram"] = hist
        # a mismatch is admitted only where the oracle's two responses are closer to each other than the two IMPLEMENTATIONS are on
        # the pairs they agree about (the noise between two correct fp32 evaluations of the same net: fma contraction inside the
        # oracle's C, OCML's expf vs glibc's): then either ordering is a correct answer
        noise = float(np.abs(got_vw.astype(np.float64) - want_vw)[~bad].max()) if (~bad).any() else 0.0
        gap_abs = r_own.astype(np.float64) - r_eng.astype(np.float64)
        if report is not None:
            report["argmax_mismatch_gap_abs_max"] = float(gap_abs.max())
            report["view_weight_noise_between_implementations_abs_max"] = noise
        assert float(gap_abs.max()) <= 2.0 * noise + 4.0 * float(ulp.max()), (float(gap_abs.max()), noise)
    if bad.any():  # ... every one of them a near-tie: the two hypotheses' responses differ by less than the fp32 noise of the
        # response itself (measured: weights differ by <= 3.4e-6 at cfg-2, <= 3.2e-5 at cfg-5's 10 views -- inside the 1e-4 the
        # weights themselves are held to), so either index yields the same weight
        assert GU.abs_err(got_vw[bad], want_vw[bad]) < tie
    return frac


def _index_check(got_idx, got_score, want_idx):
    """Integer confidence index trunc(sum_d d * p_d) (reference models/net.py:294-297): equal wherever the expectation is not
    within fp32 noise of an integer boundary (where trunc() legitimately flips)."""
    D = got_score.shape[1]
    e = (got_score.astype(np.float64) * np.arange(D, dtype=np.float64).reshape(1, D, 1, 1)).sum(1)
    bad = got_idx.astype(np.int64) != want_idx.astype(np.int64)
    frac = float(bad.mean())
    if bad.any():
        dist = np.abs(e[bad] - np.round(e[bad]))
        assert float(dist.max()) < 2e-3, float(dist.max())  # every mismatch sits on an integer boundary of the expectation
        assert int(np.abs(got_idx.astype(np.int64) - want_idx.astype(np.int64)).max()) <= 1
    assert frac < 2e-3, frac
    return frac


@pytest.mark.parametrize("scene,H,W,nsrc", [("surface", 1200, 1600, 5), ("rolled", 1200, 1600, 5), ("surface", 1056, 1920, 7),
                                            ("surface", 2048, 3072, 10)])
def test_chained_cascade_from_hip_featurenet(scene, H, W, nsrc):
    """bench.py's samples (cfg-2 on both scenes; cfg-3 = 1920x1056, N=7; since round 4 cfg-5 = 3072x2048, N=10: the whole forward's
    VALUES at the largest BASELINE configuration, not only their ranges): HIP FeatureNet features -> the whole HIP cascade vs
    oracle.cascade-style chaining on the same features."""
    P = _gpu()
    import bench
    model, params, kw = _model(P)
    cfgs = _configs(kw)
    s = bench.make_samples(1, nsrc + 1, H, W, torch.device(DEV), 0, scene)[0]
    noise = torch.rand(1, 48, H // 8, W // 8, generator=torch.Generator().manual_seed(1234)).to(DEV)
    dbg = {}
    with torch.no_grad():
        feats = model.extract_features(list(s["images"]))
        depth, conf, dpm = model(list(s["images"]), s["intrinsics"].clone(), s["extrinsics"], s["depth_min"], s["depth_max"],
                                 noise=noise, features=feats, debug=dbg)
    torch.cuda.synchronize()
    fnp = [{st: n(f[st].contiguous()) for st in (1, 2, 3)} for f in feats]
    intr, extr = n(s["intrinsics"]), n(s["extrinsics"])
    dmin, dmax = n(s["depth_min"]), n(s["depth_max"])
    O.set_num_threads(min(os.cpu_count() or 1, 48))  # 256 OpenMP threads next to torch's own pool: 7x SLOWER than 8 (measured)
    # Two oracle runs per (stage, iteration):
    #  * "forced": the oracle consumes the HIP cascade's own previous depth / view weights, i.e. IDENTICAL inputs at every
    #    Evaluation call of the real cascade -> the north star's <= 1e-3 relative, strictly, on every pixel;
    #  * "free": the oracle chains its own outputs.  Differences then compound through the hypothesis generation (on this
    #    random-texture scene the matching cost is multi-modal, a 1e-5 change of the previous depth can move the soft arg-max
    #    to another mode), so that comparison is statistical and its maxima are only recorded.
    import copy
    worst = {}
    oscore = None
    free_depth, free_vw = None, None
    prev_depth, prev_vw = None, None  # HIP outputs handed to the forced oracle
    scale = 0.125
    for stage in (3, 2, 1):
        proj = O.stage_projections(intr, extr, scale)
        scale *= 2.0
        rec0 = dbg[stage][0]
        offs = dict(propa_offsets=None if rec0["propa_offsets"] is None else n(rec0["propa_offsets"]),
                    eval_offsets=n(rec0["eval_offsets"]))
        args = (params, fnp[0][stage], [f[stage] for f in fnp[1:]], proj[:, 0], [proj[:, i] for i in range(1, proj.shape[1])],
                dmin, dmax)
        one = copy.copy(cfgs[stage])
        one.iterations = 1  # one Evaluation call at a time (propagation / inverse-regression guards are unchanged for it)
        for it, rec in enumerate(dbg[stage]):
            last = stage == 1 and it == cfgs[stage].iterations - 1
            assert last == (stage == 1)  # default config: stage 1 has a single iteration
            if it == 0:
                d_in = None if prev_depth is None else O.nearest_up2(prev_depth)
                prev_vw = None if prev_vw is None else O.nearest_up2(prev_vw)
            else:
                d_in = prev_depth
            vw_in = prev_vw
            otr = []
            _, oscore, _ = O.patchmatch_stage(one, *args, d_in, vw_in, noise=n(noise) if stage == 3 else None, trace=otr, **offs)
            orec = otr[0]
            rel = np.abs(n(rec["depth"]) - orec["depth"]) / orec["depth"]
            worst[f"s{stage}_it{it + 1}_forced_depth_rel_max"] = float(rel.max())
            assert rel.max() < 1e-3, (stage, it, float(rel.max()))
            if stage == 3 and it == 0:
                worst["view_weight_argmax_mismatch_frac"] = _argmax_check(
                    n(rec["view_weight_argmax"]), n(rec["view_weights"]), orec["view_weight_argmax"], orec["view_weights"],
                    want_resp=orec.get("view_weight_responses"), report=worst)
                assert GU.abs_err(n(rec["view_weights"]), orec["view_weights"]) < 1e-4
            prev_depth = n(rec["depth"])[:, None]
            if stage == 3 and it == 0:
                prev_vw = n(rec["view_weights"])  # computed once; later calls hand the stage-3 map through (read with a shift)
        # free-running oracle chain of this stage
        otr = []
        fdepths, _, free_vw = O.patchmatch_stage(cfgs[stage], *args, free_depth, free_vw, noise=n(noise) if stage == 3 else None,
                                                 trace=otr, **offs)
        for it, (rec, orec) in enumerate(zip(dbg[stage], otr)):
            rel = np.abs(n(rec["depth"]) - orec["depth"]) / orec["depth"]
            worst[f"s{stage}_it{it + 1}_free_depth_rel_p999"] = float(np.quantile(rel, 0.999))
            worst[f"s{stage}_it{it + 1}_free_depth_frac_over_1e-3"] = float((rel > 1e-3).mean())
            worst[f"s{stage}_it{it + 1}_free_depth_rel_max"] = float(rel.max())
        free_depth = fdepths[-1]
        if stage > 1:
            free_depth, free_vw = O.nearest_up2(free_depth), O.nearest_up2(free_vw)
    # (no pass / fail on the free-running chain: measured 3.5 % of the stage-1 pixels beyond 1e-3, p99.9 1.6e-2, while every
    #  Evaluation call on identical inputs agrees to <= 1e-5 -- the amplification is the cascade's, not the kernels')
    # integer confidence index on the stage-1 probabilities (both sides computed from their own cascade)
    score_hip = dbg[1][-1]["score"]
    _, idx_hip = P.ops.confidence(score_hip.contiguous(), H, W, want_index=True)
    _, idx_or = O.confidence(oscore, (H, W))
    worst["depth_index_mismatch_frac"] = _index_check(n(idx_hip), n(score_hip), idx_or)
    _report(test="chained_cascade", scene=scene, H=H, W=W, n_src=nsrc, **worst)


def _scene_against_reference(P, fixture, gates):
    """Free-running forward from the images against the reference's own CPU forward stored in ``fixture`` (made by
    tests/golden/make_golden.py --only scene3 / scene5; maps kept at every ``stride``-th pixel)."""
    model, params, kw = _model(P)
    g = GU.load_npz(fixture)
    H, W, nv, st = int(g["H"]), int(g["W"]), int(g["n_views"]), int(g["stride"])
    # rendered on the device (float64; measured byte-identical to the CPU rendering the fixtures were made from, and seconds instead of
    # minutes at these sizes); the signature check below admits isolated one-level differences should a host differ in its last bit
    step = float(g["camera_step"]) if "camera_step" in g else 0.08
    imgs, intr, extr, gt = synth.render_scene(nv, H, W, int(g["scene_seed"]), device=DEV, cameras=synth.synthetic_cameras(nv, H, W, step))
    imgs, gt = [im.cpu() for im in imgs], gt.cpu()
    exact = synth.scene_digest(imgs) == str(g["scene_digest"])
    ok, note = (True, "byte-identical") if exact else synth.scene_matches(imgs, g["scene_thumb"], g["scene_sums"])
    assert ok, "this host renders a different scene than the golden was made on: " + note
    noise = torch.rand(1, 48, H // 8, W // 8, generator=torch.Generator().manual_seed(int(g["noise_seed"]))).to(DEV)
    dbg = {}
    with torch.no_grad():
        depth, conf, dpm = model([im.to(DEV) for im in imgs], t(intr), t(extr), torch.tensor([425.0], device=DEV),
                                 torch.tensor([935.0], device=DEV), noise=noise, debug=dbg)
    torch.cuda.synchronize()
    sub = lambda a: a[..., ::st, ::st]  # noqa: E731
    rep = {"fixture": fixture, "H": H, "W": W, "n_src": nv - 1, "stride": st, "scene_vs_the_fixture": note}

    def stats(name, got, want):
        rel = np.abs(sub(got).astype(np.float64) - want) / np.abs(want)
        rep[name] = {"max": float(rel.max()), "p999": float(np.quantile(rel, 0.999)), "p99": float(np.quantile(rel, 0.99)),
                     "frac_over_1e-3": float((rel > 1e-3).mean()), "frac_over_1e-4": float((rel > 1e-4).mean())}
        return rep[name]

    for s_ in (3, 2, 1):
        for it, d in enumerate(dpm[s_]):
            stats(f"s{s_}_it{it + 1}", n(d), g[f"s{s_}_it{it + 1}_depth_out"])
    fin = stats("final", n(depth), g["depth"])
    err_gt = np.abs(n(depth)[0, 0] - gt.numpy())
    rep["vs_ground_truth_mm"] = {"median": float(np.median(err_gt)), "p90": float(np.quantile(err_gt, 0.9))}
    rep["view_weights_abs_max"] = GU.abs_err(sub(n(dbg[3][0]["view_weights"])), g["view_weights"])
    _, idx_hip = P.ops.confidence(dbg[1][-1]["score"].contiguous(), H, W, want_index=True)
    rep["depth_index_mismatch_frac"] = float((sub(n(idx_hip)).astype(np.int64) != g["depth_index"].astype(np.int64)).mean())
    rep["confidence_frac_over_1e-3"] = float((np.abs(sub(n(conf)) - g["confidence"]) > 1e-3).mean())
    _report(test=fixture.replace(".npz", "_vs_reference"), **rep)
    # the stage-3 first iteration has no history to amplify: strict
    assert rep["s3_it1"]["max"] < 1e-4, rep["s3_it1"]
    assert rep["view_weights_abs_max"] < 1e-4, rep["view_weights_abs_max"]
    for k in ("s3_it2", "s2_it1", "s2_it2", "s1_it1", "final"):
        assert rep[k]["p99"] < 1e-5, (k, rep[k])
        assert rep[k]["frac_over_1e-3"] < gates["frac"], (k, rep[k])
        assert rep[k]["max"] < gates["max"], (k, rep[k])
    assert fin["p999"] < gates["p999"], fin
    assert rep["vs_ground_truth_mm"]["median"] < 1.0, rep["vs_ground_truth_mm"]
    return rep


def test_cfg3_scene_end_to_end_against_the_reference_itself():
    """BASELINE configs[2] (Tanks & Temples shape: 1920x1056, N=7), FREE-RUNNING from the images, against the reference's own CPU
    forward (tests/golden/cfg3_scene.npz, every second pixel).  Round 5 gated free-running parity against the reference's CPU path at
    configs[1] only.  Gates: the bulk at the north star (p99 <= 1e-5); the soft-arg-max flips (the reference's
    own CPU<->ROCm distance is 5.4e-4 of the pixels, profiles/r05_rocm_parity.md) at 1.5x what this configuration measures."""
    P = _gpu()
    # measured (profiles/r06_parity_report.jsonl): 2.6e-4 of the final-depth pixels beyond 1e-3, max 7.2e-3, p99 2.0e-6
    _scene_against_reference(P, "cfg3_scene.npz", dict(frac=3.9e-4, max=1.1e-2, p999=5e-4))


def test_cfg5_scene_end_to_end_against_the_reference_itself():
    """BASELINE configs[4] (ETH3D shape: 3072x2048, N=10, one GPU's share), the same way (tests/golden/cfg5_scene.npz, every fourth
    pixel in both directions)."""
    P = _gpu()
    # measured (profiles/r06_parity_report.jsonl): 7.6e-6 of the final-depth pixels beyond 1e-3 (4.1e-5 at stage 2, iteration 2), max
    # 2.1e-3, p99.9 5.1e-6 -- on a rig of 0.04 rad per view (see make_golden.py: the 0.08 rig does not render identically on two hosts)
    _scene_against_reference(P, "cfg5_scene.npz", dict(frac=7e-5, max=3.2e-3, p999=1e-5))


def test_cfg2_scene_end_to_end_against_the_reference_itself():
    """BASELINE configs[1] (1600x1200, N=5, iters 1,2,2), FREE-RUNNING from the images, against the REFERENCE's own output
    (tests/golden/cfg2_scene.npz: the imported reference run by tests/golden/make_golden.py on the photo-consistent scene of
    tests/synth.render_scene; reference models/net.py:176-301).  Nothing is teacher-forced: HIP FeatureNet -> HIP cascade -> HIP
    refinement vs the reference's CPU forward.

    Gates.  The reference does not reproduce ITSELF to 1e-3 on every pixel: profiles/r03_noise_floor.json (scripts/noise_floor.py)
    has, on this scene, 4.6e-5 of the final-depth pixels beyond 1e-3 between 8 and 1 threads and 2.3e-4 between oneDNN and native
    convolutions (max 4.6e-3 / 9.1e-3; p99 2e-7), because a rounding-level change of a feature occasionally moves the soft arg-max
    of one pixel to a neighbouring hypothesis and later stages keep it.  Round 4 attributed this engine's excess over that floor
    (6.5e-4 of the pixels, max 1.9e-2 in round 3) to the warp's approximate perspective division and removed it
    (profiles/r04_ieee_attribution.md): measured now 7.3e-5 of the final-depth pixels beyond 1e-3, max 5.5e-3, depth_index mismatch
    1.5e-4 -- INSIDE the reference's own floor.  The gates sit at 1.5x those measurements (the computation is deterministic: same
    kernels, same inputs, same bits on every box), the bulk far inside the north star's 1e-3 (p99 <= 1e-5)."""
    P = _gpu()
    model, params, kw = _model(P)
    g = GU.load_npz("cfg2_scene.npz")
    H, W, nv = int(g["H"]), int(g["W"]), int(g["n_views"])
    imgs, intr, extr, gt = synth.render_scene(nv, H, W, int(g["scene_seed"]))
    assert synth.scene_digest(imgs) == str(g["scene_digest"]), "this host renders a different scene than the golden was made on"
    noise = torch.rand(1, 48, H // 8, W // 8, generator=torch.Generator().manual_seed(int(g["noise_seed"]))).to(DEV)
    dbg = {}
    with torch.no_grad():
        depth, conf, dpm = model([im.to(DEV) for im in imgs], t(intr), t(extr), torch.tensor([425.0], device=DEV),
                                 torch.tensor([935.0], device=DEV), noise=noise, debug=dbg)
    torch.cuda.synchronize()
    rep = {}

    def stats(name, got, want):
        rel = np.abs(got.astype(np.float64) - want) / np.abs(want)
        rep[name] = {"max": float(rel.max()), "p999": float(np.quantile(rel, 0.999)), "p99": float(np.quantile(rel, 0.99)),
                     "frac_over_1e-3": float((rel > 1e-3).mean()), "frac_over_1e-4": float((rel > 1e-4).mean())}
        return rep[name]

    for st in (3, 2, 1):
        for it, d in enumerate(dpm[st]):
            stats(f"s{st}_it{it + 1}", n(d), g[f"s{st}_it{it + 1}_depth_out"])
    fin = stats("final", n(depth), g["depth"])
    err_gt = np.abs(n(depth)[0, 0] - gt.numpy())
    rep["vs_ground_truth_mm"] = {"median": float(np.median(err_gt)), "p90": float(np.quantile(err_gt, 0.9))}
    # stage-3 view weights (PixelwiseNet max over D): the first free-running product of the cascade
    rep["view_weights_abs_max"] = GU.abs_err(n(dbg[3][0]["view_weights"]), g["view_weights"])
    # confidence: reads the stage-1 probabilities at the integer index trunc(sum_d d p_d)
    cdiff = np.abs(n(conf) - g["confidence"])
    rep["confidence_frac_over_1e-3"] = float((cdiff > 1e-3).mean())
    score_hip = dbg[1][-1]["score"]
    _, idx_hip = P.ops.confidence(score_hip.contiguous(), H, W, want_index=True)
    idx_bad = n(idx_hip).astype(np.int64) != g["depth_index"].astype(np.int64)
    rep["depth_index_mismatch_frac"] = float(idx_bad.mean())
    _report(test="cfg2_scene_vs_reference", **rep)
    # the stage-3 first iteration has no history to amplify: strict
    assert rep["s3_it1"]["max"] < 1e-4, rep["s3_it1"]
    assert rep["view_weights_abs_max"] < 1e-4, rep["view_weights_abs_max"]
    for k in ("s3_it2", "s2_it1", "s2_it2", "s1_it1", "final"):
        assert rep[k]["p99"] < 1e-5, (k, rep[k])
        assert rep[k]["frac_over_1e-3"] < 1.15e-4, (k, rep[k])  # measured 0 / 0 / 5.0e-5 / 7.5e-5 / 7.3e-5 (round 3: up to 6.5e-4)
        assert rep[k]["max"] < 8.5e-3, (k, rep[k])              # measured <= 5.5e-3 (round 3: 2.2e-2)
    assert fin["p999"] < 1.5e-4, fin                            # measured 9.7e-5 (round 3: 5.5e-4)
    assert rep["vs_ground_truth_mm"]["median"] < 1.0, rep["vs_ground_truth_mm"]  # the reference itself: 0.63 mm
    assert rep["depth_index_mismatch_frac"] < 2.2e-4 and rep["confidence_frac_over_1e-3"] < 2.7e-3, rep  # measured 1.5e-4 / 1.8e-3


@pytest.mark.parametrize("stage,n_src,H,W", [(3, 7, 1056, 1920), (2, 7, 1056, 1920), (1, 7, 1056, 1920),
                                             (2, 5, 1200, 1600), (3, 10, 2048, 3072), (2, 10, 2048, 3072), (1, 10, 2048, 3072)])
def test_fullsize_stage_against_oracle_cfg3_cfg5(stage, n_src, H, W):
    """One PatchMatch stage at cfg-3 / cfg-5 sizes (and the cfg-2 stage the round-1 suite skipped) vs the CPU oracle on identical
    inputs (same conv offsets)."""
    P = _gpu()
    model, params, kw = _model(P)
    cfg = _configs(kw)[stage]
    pm = getattr(model, f"patchmatch_{stage}")
    scale = {3: 8, 2: 4, 1: 2}[stage]
    C = {3: 64, 2: 32, 1: 16}[stage]
    h, w = H // scale, W // scale
    feats = synth.synthetic_features(n_src + 1, C, h, w, seed=stage)
    intr, extr = synth.synthetic_cameras(n_src + 1, H, W)
    proj = synth.stage_projections(intr, extr, 1.0 / scale)
    dmin, dmax = np.array([425.0], np.float32), np.array([935.0], np.float32)
    gen = torch.Generator().manual_seed(4321)
    noise = torch.rand(1, 48, h, w, generator=gen)
    if stage == 3:
        depth, vw = None, None
    else:  # a smooth previous estimate + noise, view weights in [0,1]
        yy, xx = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
        depth = (680.0 + 200.0 * torch.sin(xx / 37.0) * torch.cos(yy / 29.0) + 3.0 * torch.randn(h, w, generator=gen))
        depth = depth.clamp(425.0, 935.0)[None, None].numpy()
        vw = torch.rand(1, n_src, h, w, generator=gen).numpy()
    dbg = []
    with torch.no_grad():
        pm(ref_feature=feats[0].to(DEV), src_features=[f.to(DEV) for f in feats[1:]], ref_proj=t(proj[:, 0]),
           src_projs=[t(proj[:, i]) for i in range(1, proj.shape[1])], depth_min=t(dmin), depth_max=t(dmax),
           depth=torch.empty(0, device=DEV) if depth is None else t(depth),
           view_weights=torch.empty(0, device=DEV) if vw is None else t(vw), noise=noise.to(DEV), debug=dbg)
    torch.cuda.synchronize()
    O.set_num_threads(min(os.cpu_count() or 1, 48))  # 256 OpenMP threads next to torch's own pool: 7x SLOWER than 8 (measured)
    # every Evaluation call on identical inputs: iteration k > 1 of the oracle starts from the HIP path's iteration k-1 depth
    # (what the stage does internally; see test_cfg2_chained_cascade_from_hip_featurenet for why the free-running chain is not
    # a pass / fail criterion)
    import copy
    one = copy.copy(cfg)
    one.iterations = 1
    otr = []
    d_in, vw_in = depth, vw
    for it, rec in enumerate(dbg):
        tr1 = []
        O.patchmatch_stage(one, params, feats[0].numpy(), [f.numpy() for f in feats[1:]], proj[:, 0],
                           [proj[:, i] for i in range(1, proj.shape[1])], dmin, dmax, d_in, vw_in, noise=noise.numpy(),
                           propa_offsets=None if dbg[0]["propa_offsets"] is None else n(dbg[0]["propa_offsets"]),
                           eval_offsets=n(dbg[0]["eval_offsets"]), trace=tr1)
        otr.append(tr1[0])
        d_in = n(rec["depth"])[:, None]
        if vw_in is None:
            vw_in = n(rec["view_weights"])
    worst = {}
    for it, (rec, orec) in enumerate(zip(dbg, otr)):
        if it == 0:
            assert GU.rel_err(n(rec["depth_sample"]), orec["depth_sample"]) < 2e-6
            worst["similarity_abs_max"] = GU.abs_err(n(rec["similarity"]), orec["similarity"])
            assert worst["similarity_abs_max"] < 1e-4
            if stage == 3:
                # (the weights first: _argmax_check's tie criterion is stated in terms of them)
                worst["view_weights_abs_max"] = GU.abs_err(n(rec["view_weights"]), orec["view_weights"])
                # cfg-5's synthetic rig reaches 0.8 rad between reference and source view: grazing projections with steep (random)
                # feature gradients -- measured 1.0e-4 there since the warp uses the reference's IEEE chain (round 3: 2.3e-4 with
                # the approximate perspective division, tolerance 5e-4), <= 2e-5 elsewhere
                assert worst["view_weights_abs_max"] < (1e-4 if n_src <= 7 else 2e-4)
                worst["view_weight_argmax_mismatch_frac"] = _argmax_check(
                    n(rec["view_weight_argmax"]), n(rec["view_weights"]), orec["view_weight_argmax"], orec["view_weights"],
                    tie=1e-4 if n_src <= 7 else 2e-4, want_resp=orec.get("view_weight_responses"), report=worst)
        rel = np.abs(n(rec["depth"]) - orec["depth"]) / orec["depth"]
        worst[f"it{it + 1}_depth_rel_max"] = float(rel.max())
        assert rel.max() < 1e-3, (it, float(rel.max()))
    _report(test="fullsize_stage", stage=stage, n_src=n_src, H=H, W=W, **worst)


@pytest.mark.parametrize("nimg,H,W", [(6, 1200, 1600), (1, 2048, 3072)])
def test_featurenet_hip_matches_miopen_fullsize(nimg, H, W):
    """forward_hip (stem / Winograd / MFMA convolutions / folded FPN head) vs the same module on PyTorch-ROCm (MIOpen) at the
    benchmark's size: tile edges, XCD tile order and > 2^31-element offsets do not show up at 96x128."""
    P = _gpu()
    model, _, _ = _model(P)
    g = torch.Generator().manual_seed(7)
    x = torch.rand(nimg, 3, H, W, generator=g).to(DEV)
    with torch.no_grad():
        got = model.feature.forward_hip(x)
        worst = {}
        for i in range(nimg):  # MIOpen one image at a time keeps its workspace small
            ref = model.feature(x[i:i + 1])
            for s in (1, 2, 3):
                a, b = got[s][i:i + 1].permute(0, 3, 1, 2), ref[s]
                assert a.shape == b.shape
                e = float((a - b).abs().max() / b.abs().max())
                worst[f"s{s}"] = max(worst.get(f"s{s}", 0.0), e)
                assert e < 5e-5, (i, s, e)
            del ref
    _report(test="featurenet_fullsize", nimg=nimg, H=H, W=W, **worst)


@pytest.mark.parametrize("hip_feature_net", [True, False])
def test_end_to_end_from_images_both_featurenets(hip_feature_net):
    """Raw images -> FeatureNet -> cascade -> refinement vs the reference's CPU result (golden).  hip_feature_net=True is the
    product path (HIP FeatureNet / Refinement), False runs the same modules on MIOpen.  FeatureNet rounding differs from the
    CPU backend's and amplifies down the cascade, so the criterion is statistical; the measured maxima are pinned here:
    a regression past them fails."""
    P = _gpu()
    g, params, kw = GU.load_case("default")
    m = P.PatchmatchNet(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.to(DEV).eval()
    m.hip_feature_net = hip_feature_net
    nv = int(g["n_views"])
    imgs = [t(g[f"image_{v}"]) for v in range(nv)]
    with torch.no_grad():
        depth, conf, _ = m(imgs, t(g["intrinsics"]), t(g["extrinsics"]), t(g["depth_min"]), t(g["depth_max"]), noise=t(g["noise"]))
    rel = np.abs(n(depth) - g["depth"]) / np.abs(g["depth"])
    q999, mx = float(np.quantile(rel, 0.999)), float(rel.max())
    _report(test="end_to_end_from_images", hip_feature_net=hip_feature_net, rel_p999=q999, rel_max=mx)
    assert q999 < 1e-3, q999
    assert mx < 1e-4, mx  # measured 1.6e-6 (both FeatureNet paths, profiles/r02_parity_report.jsonl); 2e-2 was allowed in round 1


def _identical_inputs_against_rocm(ref, b_feats, ref_image, Kd, Ed, dmin, dmax, d_depth, d_dpm, vw_engine, stats, cpu_ref=None,
                                   attribution=None):
    """The reference's stage modules on ROCm (``ref`` = the pinned-draw archive on the device) are handed THIS ENGINE's previous-stage
    depth and view weights (nearest x2, models/net.py:272-275), the reference's own features ``b_feats`` and torch-on-ROCm projections
    (models/net.py:225-231); iteration 1 of every stage and the refinement are then one call of each side on the same tensors -- the
    north star's "identical inputs" clause against the GPU path; iteration 2 has one free step inside the stage.  ``d_depth`` /
    ``d_dpm`` / ``vw_engine``: this engine's outputs on the same features.  Returns {stage-iteration or "final": stats}.

    ``cpu_ref`` (the same archive loaded on the CPU) + ``attribution`` (a dict): for every first iteration with a pixel beyond 1e-3
    of the reference-on-ROCm, the reference's OWN CPU path is run on the very same tensors and, per such pixel, attribution receives
    {engine_vs_rocm, engine_vs_cpu, rocm_vs_cpu}: whose number is the outlier."""
    Fn = torch.nn.functional
    depth_in, vw_in = torch.empty(0, device=DEV), torch.empty(0, device=DEV)
    forced, scale = {}, 0.125
    with torch.no_grad():
        for stage in (3, 2, 1):
            Ks = Kd.clone()
            Ks[:, :, :2] *= scale
            proj = Ed.clone()
            proj[:, :, :3, :4] = torch.matmul(Ks, Ed[:, :, :3, :4])
            pl = torch.unbind(proj, 1)
            scale *= 2.0
            depths, _, _ = getattr(ref, f"patchmatch_{stage}")(
                ref_feature=b_feats[0][stage], src_features=[f[stage] for f in b_feats[1:]], ref_proj=pl[0], src_projs=list(pl[1:]),
                depth_min=dmin, depth_max=dmax, depth=depth_in, view_weights=vw_in)
            for it, d in enumerate(depths):
                forced[f"s{stage}_it{it + 1}"] = stats(n(d_dpm[stage][it]), n(d))
            rel1 = np.abs(n(d_dpm[stage][0]).astype(np.float64) - n(depths[0])) / np.abs(n(depths[0]))
            if cpu_ref is not None and attribution is not None and (rel1 > 1e-3).any():
                c = lambda x: x.detach().cpu()  # noqa: E731
                cdepths, _, _ = getattr(cpu_ref, f"patchmatch_{stage}")(
                    ref_feature=c(b_feats[0][stage]), src_features=[c(f[stage]) for f in b_feats[1:]], ref_proj=c(pl[0]),
                    src_projs=[c(x) for x in pl[1:]], depth_min=c(dmin), depth_max=c(dmax), depth=c(depth_in), view_weights=c(vw_in))
                cpu1 = cdepths[0].numpy().astype(np.float64)
                eng1, roc1 = n(d_dpm[stage][0]).astype(np.float64), n(depths[0]).astype(np.float64)
                where = np.argwhere(rel1 > 1e-3)
                attribution[f"s{stage}_it1"] = [
                    {"pixel": [int(v) for v in ix], "engine_vs_rocm": float(rel1[tuple(ix)]),
                     "engine_vs_cpu": float(abs(eng1[tuple(ix)] - cpu1[tuple(ix)]) / abs(cpu1[tuple(ix)])),
                     "rocm_vs_cpu": float(abs(roc1[tuple(ix)] - cpu1[tuple(ix)]) / abs(cpu1[tuple(ix)]))} for ix in where[:64]]
                attribution[f"s{stage}_it1_engine_vs_cpu_all_pixels_max"] = float((np.abs(eng1 - cpu1) / np.abs(cpu1)).max())
                attribution[f"s{stage}_it1_rocm_vs_cpu_all_pixels_max"] = float((np.abs(roc1 - cpu1) / np.abs(cpu1)).max())
            if stage > 1:
                depth_in = Fn.interpolate(d_dpm[stage][-1].detach(), scale_factor=2.0, mode="nearest")
                vw_in = Fn.interpolate(vw_engine if stage == 3 else vw_in, scale_factor=2.0, mode="nearest")
        forced["final"] = stats(n(d_depth), n(ref.upsample_net(ref_image, d_dpm[1][-1].detach(), dmin, dmax)))
    torch.cuda.synchronize()
    return forced


def _rel_stats(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    rel = np.abs(got - want) / np.abs(want)
    return {"p50": float(np.median(rel)), "p99": float(np.quantile(rel, 0.99)), "p999": float(np.quantile(rel, 0.999)),
            "frac_over_1e-3": float((rel > 1e-3).mean()), "frac_over_1e-4": float((rel > 1e-4).mean()), "max": float(rel.max())}


@pytest.mark.parametrize("H,W,nsrc,rig", [(1056, 1920, 7, "general"), (2048, 3072, 10, "orbit")])
def test_identical_inputs_against_the_reference_on_rocm_cfg3_cfg5(H, W, nsrc, rig):
    """BASELINE configs[2] and configs[4] against the reference's GPU path (PyTorch-ROCm, pinned-draw archive) ON IDENTICAL INPUTS at
    every stage boundary -- the first iteration of every stage and the refinement within the north star's 1e-3 on EVERY pixel --, at
    configs[2] on a rig WITHOUT the y-axis symmetry of the other full-size tests (tests/synth.general_cameras: roll, pitch, off-orbit
    translations, fx != fy: every entry of the relative projections is exercised, epipolar lines are not horizontal).  Also records
    the free-running difference with the reference's own features (no golden of the CPU reference at these sizes: no gate on it)."""
    P = _gpu()
    pinned = os.path.join(ROOT, "oracle", "_ref", "patchmatchnet_reference_pinned.pt")
    if not os.path.isfile(pinned):
        pytest.skip("oracle/_ref/patchmatchnet_reference_pinned.pt was never built (python oracle/make_ref.py needs the reference checkout)")
    model, params, kw = _model(P)
    cams = synth.general_cameras(nsrc + 1, H, W) if rig == "general" else None
    imgs, intr, extr, _ = synth.render_scene(nsrc + 1, H, W, seed=5, device=DEV, cameras=cams)
    dimgs = [im.to(DEV).contiguous() for im in imgs]
    noise = torch.rand(1, 48, H // 8, W // 8, generator=torch.Generator().manual_seed(4242)).to(DEV)
    dmin, dmax = torch.tensor([425.0], device=DEV), torch.tensor([935.0], device=DEV)
    torch.backends.cudnn.benchmark = False
    ref = torch.jit.load(pinned, map_location=DEV).eval()
    ref.patchmatch_3.depth_initialization.noise = noise
    with torch.no_grad():
        for _ in range(2):  # the TorchScript executor profiles on its first call and specialises on its second
            b_depth, _, b_dpm = ref([im.clone() for im in dimgs], t(intr), t(extr), dmin, dmax)
        b_feats = [ref.feature(im) for im in dimgs]
        d_dbg = {}
        d_depth, _, d_dpm = model([im.clone() for im in dimgs], t(intr), t(extr), dmin, dmax, noise=noise,
                                  features=[{s: f[s].contiguous() for s in (1, 2, 3)} for f in b_feats], debug=d_dbg)
    torch.cuda.synchronize()
    cpu_ref = torch.jit.load(pinned, map_location="cpu").eval()
    attribution = {}
    forced = _identical_inputs_against_rocm(ref, b_feats, dimgs[0], t(intr), t(extr), dmin, dmax, d_depth, d_dpm,
                                            d_dbg[3][0]["view_weights"], _rel_stats, cpu_ref=cpu_ref, attribution=attribution)
    free = {f"s{st}_it{it + 1}": _rel_stats(n(d), n(b_dpm[st][it])) for st in (3, 2, 1) for it, d in enumerate(d_dpm[st])}
    free["final"] = _rel_stats(n(d_depth), n(b_depth))
    _report(test="identical_inputs_vs_reference_on_rocm", H=H, W=W, n_src=nsrc, rig=rig, forced=forced, free_running_on_the_reference_features=free,
            pixels_beyond_tolerance_attributed=attribution)
    # MEASURED (profiles/r05_rocm_parity.md): configs[2], general rig: max 7.1e-6 / 2.4e-4 / 4.1e-6 (stages 3 / 2 / 1), refinement 2.0e-7 --
    # every pixel inside the north star's 1e-3.  configs[4] (N=10, 6.3 M pixels): 1.8e-5 / 1.6e-3 / 1.1e-5, refinement 2.2e-7: ONE pixel
    # of the 393 K of stage 2 sits at 1.6e-3 (the reference's GPU kernels are not this engine's arithmetic operation for operation --
    # ATen takes divisions by host scalars as reciprocal multiplies, contracts differently -- and with ten views one call is enough for
    # one soft arg-max to tip); against the CPU oracle, whose chain the kernels restate, the same calls hold <= 4e-5 on every pixel
    # (test_chained_cascade_from_hip_featurenet).  So: strict at configs[2]; at configs[4] at most 1e-5 of the pixels beyond 1e-3 and
    # none beyond 5e-3, the bulk (p99) inside 1e-5 for both.
    for k in ("s3_it1", "s2_it1", "s1_it1", "final"):
        assert forced[k]["p99"] < 1e-5, (k, forced[k])
        if nsrc <= 7:
            assert forced[k]["max"] < 1e-3, (k, forced[k])
        elif forced[k]["max"] >= 1e-3:
            # configs[4]: a pixel beyond 1e-3 of the reference-on-ROCm is admitted ONLY where the reference's own CPU path, run on the
            # very same tensors, agrees with THIS ENGINE (<= 1e-4) and disagrees with its own GPU path by the same amount: then the
            # outlier is the reference's ROCm arithmetic at that pixel (ATen's reciprocal-multiply divisions, another fma order: one
            # soft arg-max tipping among ten views), not this engine's (round 5 allowed "<= 1e-5 of the pixels, none beyond 5e-3"
            # without looking at them: VERDICT r05 weak 2a)
            assert k in attribution and attribution[k], (k, forced[k], "a pixel beyond 1e-3 that was not attributed")
            for px in attribution[k]:
                assert px["engine_vs_cpu"] < 1e-4 and px["rocm_vs_cpu"] > 0.5 * px["engine_vs_rocm"], (k, px)
            assert forced[k]["frac_over_1e-3"] <= 1e-5, (k, forced[k])
    assert free["final"]["p99"] < (1e-4 if nsrc <= 7 else 2e-3), free["final"]  # the bulk; the tail is the cascade's amplification


def test_cfg2_scene_against_the_reference_on_rocm():
    """The north star's parity clause names the reference's GPU path: the UNMODIFIED reference network on THIS MI355X through
    PyTorch-ROCm (reference eval.py:37-41: ``torch.jit.load`` + ``.cuda()``; models/net.py:203-208 FeatureNet per image), here the
    pinned-draw archive of oracle/make_ref.py (bit for bit the reference, stage-3 draw handed in: tests/test_reference_archive.py) so
    that the reference-on-ROCm, the reference-on-CPU (tests/golden/cfg2_scene.npz) and this engine all see ONE draw.

    Legs (all on BASELINE configs[1], 1600x1200, N=5, iters 1,2,2, the golden's scene):
      A  reference on the CPU                       = tests/golden/cfg2_scene.npz
      B  reference on ROCm (MIOpen FeatureNet + ATen-HIP cascade)
      C  this engine, free-running from the images  (HIP FeatureNet + HIP cascade)
      D  this engine fed B's OWN FeatureNet outputs (``features=``: MIOpen FeatureNet + HIP cascade)
    D vs B isolates this engine's cascade from the FeatureNet backends and is GATED at the level C holds against A (the CPU golden):
    the cascade is as close to the reference on the GPU as it is to the reference on the CPU.  B vs A is the reference's own
    CPU<->GPU floor; C vs B (what bench.py's ``reference_rocm.parity_vs_this_engine`` reports) is recorded beside it and must not
    exceed that floor by more than the cascade's own share.  Everything lands in gpurun_out/rocm_parity.json -> profiles/r05_rocm_parity.md."""
    P = _gpu()
    pinned = os.path.join(ROOT, "oracle", "_ref", "patchmatchnet_reference_pinned.pt")
    if not os.path.isfile(pinned):
        pytest.skip("oracle/_ref/patchmatchnet_reference_pinned.pt was never built (python oracle/make_ref.py needs the reference checkout)")
    model, params, kw = _model(P)
    g = GU.load_npz("cfg2_scene.npz")
    H, W, nv = int(g["H"]), int(g["W"]), int(g["n_views"])
    imgs, intr, extr, gt = synth.render_scene(nv, H, W, int(g["scene_seed"]))
    assert synth.scene_digest(imgs) == str(g["scene_digest"]), "this host renders a different scene than the golden was made on"
    noise = torch.rand(1, 48, H // 8, W // 8, generator=torch.Generator().manual_seed(int(g["noise_seed"]))).to(DEV)
    dimgs = [im.to(DEV) for im in imgs]
    dmin, dmax = torch.tensor([425.0], device=DEV), torch.tensor([935.0], device=DEV)
    torch.backends.cudnn.benchmark = False  # (reference eval.py:301 turns MIOpen's find mode on: algorithm choice by timing; off here)

    ref = torch.jit.load(pinned, map_location=DEV).eval()
    ref.patchmatch_3.depth_initialization.noise = noise
    with torch.no_grad():
        for _ in range(2):  # the TorchScript executor profiles on its first call and specialises on its second
            b_depth, b_conf, b_dpm = ref([im.clone() for im in dimgs], t(intr), t(extr), dmin, dmax)
        b_feats = [ref.feature(im) for im in dimgs]  # B's own FeatureNet outputs (MIOpen), {3: [1,64,h/8,w/8], 2: ..., 1: ...}
        torch.cuda.synchronize()
        c_dbg, d_dbg = {}, {}
        c_depth, c_conf, c_dpm = model([im.clone() for im in dimgs], t(intr), t(extr), dmin, dmax, noise=noise, debug=c_dbg)
        own = model.extract_features(dimgs)
        d_depth, d_conf, d_dpm = model([im.clone() for im in dimgs], t(intr), t(extr), dmin, dmax, noise=noise,
                                       features=[{s: f[s].contiguous() for s in (1, 2, 3)} for f in b_feats], debug=d_dbg)
    torch.cuda.synchronize()

    def stats(got, want):
        got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
        rel = np.abs(got - want) / np.abs(want)
        return {"p50": float(np.median(rel)), "p99": float(np.quantile(rel, 0.99)), "p999": float(np.quantile(rel, 0.999)),
                "frac_over_1e-3": float((rel > 1e-3).mean()), "frac_over_1e-4": float((rel > 1e-4).mean()), "max": float(rel.max())}

    def maps(depth, dpm):
        out = {f"s{st}_it{it + 1}": n(d) for st in (3, 2, 1) for it, d in enumerate(dpm[st])}
        out["final"] = n(depth)
        return out

    A = {k: g[f"{k}_depth_out"] for k in ("s3_it1", "s3_it2", "s2_it1", "s2_it2", "s1_it1")}
    A["final"] = g["depth"]
    B, C, D = maps(b_depth, b_dpm), maps(c_depth, c_dpm), maps(d_depth, d_dpm)
    rep = {"what": "relative depth differences at 1600x1200 N=5 on the cfg2 golden scene, one stage-3 draw for all legs",
           "legs": {"A": "reference on the CPU (tests/golden/cfg2_scene.npz)", "B": "reference on ROCm (pinned-draw archive, MIOpen + ATen-HIP)",
                    "C": "this engine, free-running from the images", "D": "this engine fed B's own FeatureNet outputs"},
           "B_vs_A_reference_cpu_gpu_floor": {k: stats(B[k], A[k]) for k in A},
           "C_vs_A": {k: stats(C[k], A[k]) for k in A}, "C_vs_B": {k: stats(C[k], B[k]) for k in A},
           "D_vs_B_cascade_only": {k: stats(D[k], B[k]) for k in A}}
    # ---- F: IDENTICAL INPUTS at every stage boundary (see _identical_inputs_against_rocm)
    forced = _identical_inputs_against_rocm(ref, b_feats, dimgs[0], t(intr), t(extr), dmin, dmax, d_depth, d_dpm, d_dbg[3][0]["view_weights"], stats)
    rep["F_identical_inputs_at_every_stage_boundary"] = forced
    # FeatureNet backends against each other (relative to each map's scale): MIOpen (B) vs this engine's HIP convolutions
    rep["featurenet_hip_vs_miopen_rel_to_scale"] = {
        f"s{s}": float(max(((own[v][s] - b_feats[v][s]).abs().max() / b_feats[v][s].abs().max()).item() for v in range(nv)))
        for s in (3, 2, 1)}
    # view weights of the first Evaluation (no history): D has B's features, so its view weights answer for the kernels alone
    rep["view_weights_abs_max_C_vs_A"] = GU.abs_err(n(c_dbg[3][0]["view_weights"]), g["view_weights"])
    conf_bad = np.abs(n(d_conf) - n(b_conf)) > 1e-3
    rep["confidence_frac_over_1e-3_D_vs_B"] = float(conf_bad.mean())
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "rocm_parity.json"), "w") as f:
        json.dump(rep, f, indent=1)
    _report(test="cfg2_scene_vs_reference_on_rocm", **{k: v for k, v in rep.items() if k not in ("what", "legs")})

    # identical inputs: the north star's 1e-3 on EVERY pixel, against the reference's GPU path
    for k in ("s3_it1", "s2_it1", "s1_it1", "final"):
        assert forced[k]["max"] < 1e-3, (k, forced[k])
    dvb, floor_all = rep["D_vs_B_cascade_only"], rep["B_vs_A_reference_cpu_gpu_floor"]
    # the first Evaluation has no history to amplify: strict, as against the CPU golden (measured 4.7e-6; the reference's own
    # CPU<->GPU difference there: 6.0e-6)
    assert dvb["s3_it1"]["max"] < 1e-4, dvb["s3_it1"]
    for k in ("s3_it2", "s2_it1", "s2_it2", "s1_it1", "final"):
        assert dvb[k]["p99"] < 1e-5, (k, dvb[k])                  # the bulk: measured <= 5.7e-7
        # MEASURED (profiles/r05_rocm_parity.md): with the reference's own ROCm features this engine's final depth has 4.9e-4 of its
        # pixels beyond 1e-3 of the reference-on-ROCm's (max 1.7e-2) -- and the reference-on-ROCm has 5.6e-4 of ITS pixels beyond 1e-3 of
        # the reference-on-CPU's (max 1.8e-2): the reference's two backends differ from each other by as much, through the same
        # mechanism (a rounding-level change flips the soft arg-max of a pixel, later stages keep it), while this engine holds 7.3e-5
        # against the CPU output it was aligned with in round 4.  Gates at 1.5x measured, and never beyond the reference's own floor.
        assert dvb[k]["frac_over_1e-3"] < 7.5e-4, (k, dvb[k])
        assert dvb[k]["max"] < 4e-2, (k, dvb[k])
        assert dvb[k]["frac_over_1e-3"] <= 1.25 * floor_all[k]["frac_over_1e-3"] + 1e-5, (k, dvb[k], floor_all[k])
    # the free-running difference to the reference on ROCm is no larger than the reference's own CPU<->GPU floor plus this engine's
    # share against the CPU output (C vs A)
    floor, cva, cvb = floor_all["final"], rep["C_vs_A"]["final"], rep["C_vs_B"]["final"]
    assert cvb["frac_over_1e-3"] <= 1.5 * (floor["frac_over_1e-3"] + cva["frac_over_1e-3"]) + 1e-5, (cvb, floor, cva)
