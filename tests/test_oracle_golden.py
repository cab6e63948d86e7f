"""Pins the CPU oracle (oracle/) against golden vectors produced by the reference itself (tests/golden/)."""
import numpy as np
import pytest

import goldenutil as GU
from oracle import oracle as O


def _configs(kw):
    return O.default_stage_configs(kw["patchmatch_interval_scale"], kw["propagation_range"],
                                   kw["patchmatch_iteration"], kw["patchmatch_num_sample"],
                                   kw["propagate_neighbors"], kw["evaluate_neighbors"])


@pytest.mark.parametrize("name", ["A", "B"])
def test_differentiable_warping_known_answers(name):
    g = GU.load_npz("ops_small.npz")
    out = O.differentiable_warping(g[f"{name}_src"], g[f"{name}_src_proj"], g[f"{name}_ref_proj"], g[f"{name}_depth"])
    ref = g[f"{name}_warped"]
    assert out.shape == ref.shape
    assert GU.abs_err(out, ref) < 2e-5
    # the negative-depth sentinel must produce exact zeros in the same places (case A has them)
    if name == "A":
        assert (ref == 0).any()
    np.testing.assert_array_equal(out == 0, ref == 0)


@pytest.mark.parametrize("case", ["default", "variant", "counts", "dilations", "rig"])
@pytest.mark.parametrize("stage", [3, 2, 1])
def test_stage_against_golden(case, stage):
    """Each PatchMatch stage fed the reference's own inputs reproduces every per-iteration intermediate."""
    g, params, kw = GU.load_case(case)
    cfg = _configs(kw)[stage]
    feats, proj, depth, vw = GU.stage_inputs(g, kw, stage)
    tr = []
    depths, score, vw_out = O.patchmatch_stage(
        cfg, params, feats[0], feats[1:], proj[:, 0], [proj[:, i] for i in range(1, proj.shape[1])],
        g["depth_min"], g["depth_max"], depth, vw, noise=g["noise"] if stage == 3 else None,
        propa_offsets=g.get(f"s{stage}_propa_offsets"), eval_offsets=g[f"s{stage}_eval_offsets"], trace=tr)
    assert len(depths) == cfg.iterations
    assert GU.abs_err(tr[0]["feature_weight"], g[f"s{stage}_feature_weight"]) < 2e-5
    for it in range(1, cfg.iterations + 1):
        rec = tr[it - 1]
        key = f"s{stage}_it{it}_"
        assert GU.rel_err(rec["depth_sample"], g[key + "depth_sample"]) < 1e-5
        assert GU.abs_err(rec["similarity"], g[key + "similarity"]) < 5e-5
        assert GU.abs_err(rec["score"], g[key + "score"]) < 2e-4
        assert GU.abs_err(rec["view_weights"], g[key + "view_weights"]) < 1e-5
        assert GU.rel_err(rec["depth"], g[key + "depth"]) < 1e-4
        assert GU.rel_err(depths[it - 1], g[key + "depth_out"]) < 1e-4


@pytest.mark.parametrize("case", ["default", "variant", "counts", "dilations", "rig"])
def test_numpy_offset_heads(case):
    g, params, kw = GU.load_case(case)
    for stage in (3, 2, 1):
        dil = kw["propagation_range"][stage - 1]
        ref = g[f"feature_0_s{stage}"]
        ev = O.dilated_conv3x3(ref, params[f"patchmatch_{stage}.eval_conv.weight"],
                               params[f"patchmatch_{stage}.eval_conv.bias"], dil)
        assert GU.abs_err(ev, g[f"s{stage}_eval_offsets"]) < 1e-4
        if f"s{stage}_propa_offsets" in g:
            pr = O.dilated_conv3x3(ref, params[f"patchmatch_{stage}.propa_conv.weight"],
                                   params[f"patchmatch_{stage}.propa_conv.bias"], dil)
            assert GU.abs_err(pr, g[f"s{stage}_propa_offsets"]) < 1e-4


@pytest.mark.parametrize("case", ["default", "variant", "counts", "dilations", "rig"])
def test_cascade_and_confidence(case):
    """Whole cascade chained from the FeatureNet outputs (errors compound down the stages, tolerance 1e-3 as in
    BASELINE.json's north_star) and the confidence epilogue on the golden stage-1 probabilities."""
    g, params, kw = GU.load_case(case)
    nv = int(g["n_views"])
    feats = [{s: g[f"feature_{v}_s{s}"] for s in (1, 2, 3)} for v in range(nv)]
    d1, score, out = O.cascade(params, feats, g["intrinsics"], g["extrinsics"], g["depth_min"], g["depth_max"],
                               g["noise"], configs=_configs(kw))
    last = kw["patchmatch_iteration"][0]
    assert GU.rel_err(d1, g[f"s1_it{last}_depth_out"]) < 1e-3
    H, W = g["confidence"].shape[1:]
    conf, idx = O.confidence(g[f"s1_it{last}_score"], (H, W))
    mism = float((np.abs(conf - g["confidence"]) > 1e-4).mean())
    assert mism < 2e-3, mism
