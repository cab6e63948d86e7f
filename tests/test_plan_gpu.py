"""GPU tests of the launch-plan replay (include/pmn_hip.h pmn_plan_*, patchmatchnet_amd/graph.py PlannedForward): a recorded forward,
replayed from C with plain launches, hands out the eager forward's bits -- also with several samples in flight on their own streams on
the runtime's DEFAULT hardware queues, which is exactly where rounds 2-5's overlapped forwards did not (DESIGN_LESSONS.md lessons 45-46:
two kernels of the library computed wrong values beside co-running fp16 MFMA kernels; fixed in round 6).
Nothing here sets GPU_MAX_HW_QUEUES; tests/conftest.py no longer does either."""
import os
import subprocess
import sys

import pytest
import torch

import goldenutil as GU
import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    import patchmatchnet_amd as P
    _, params, kw = GU.load_case("default")
    model = P.PatchmatchNet(**kw)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    return model.cuda().eval()


def _sample(n_src, H, W, seed):
    imgs, intr, extr, _ = synth.render_scene(n_src + 1, H, W, seed=seed, device="cuda")
    return dict(images=[im.cuda().contiguous() for im in imgs], intrinsics=torch.as_tensor(intr).cuda(),
                extrinsics=torch.as_tensor(extr).cuda(), depth_min=torch.tensor([425.0]).cuda(), depth_max=torch.tensor([935.0]).cuda())


def _call(f, s):
    out = f([im for im in s["images"]], s["intrinsics"].clone(), s["extrinsics"], s["depth_min"], s["depth_max"])
    return out[0], out[1]


def test_suite_runs_on_the_default_hardware_queues():
    assert os.environ.get("GPU_MAX_HW_QUEUES") is None


def test_a_plan_recorded_on_one_sample_replays_any_other():
    """Record on sample A, replay on B, C, ...: the outputs must be B's, C's eager outputs bit for bit.  (An operator that had run
    during the recording instead of being part of the plan would leave A's values behind.)"""
    assert torch.cuda.is_available(), "GPU tests selected but no ROCm device is visible"
    from patchmatchnet_amd.graph import PlannedForward
    model = _model()
    for in_place in (False, True):
        slot = PlannedForward(model, inputs_in_place=in_place)
        with torch.no_grad():
            for k in range(4):
                s = _sample(3, 96, 128, seed=20 + k)
                torch.manual_seed(100 + k)
                want = _call(model, s)
                torch.manual_seed(100 + k)
                got = _call(slot, s)
                torch.cuda.synchronize()
                assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), (in_place, k)
        assert slot.captures == 1 and slot.replays == 4
        handle = next(iter(slot.cache.values()))[0]
        names = handle.kernel_names()
        assert handle.count == len(names) >= 35, handle.count
        # the whole forward is in the plan: FeatureNet's stem, the five warp+correlate launches, Refinement, the confidence epilogue
        assert any("stem_f16s_kernel" in n for n in names) and any("refine_fused_kernel" in n for n in names)
        assert sum("gather_corr_kernel" in n or "pixelwise_wave_kernel" in n for n in names) == 5 + 3  # + FeatureWeightNet per stage
        assert "normalize_depth_kernel" in " ".join(names) and "confidence" in names[-1]


def test_injected_features_replay():
    """eval.py's encode-once path: the pyramids come from a FeatureNet pass outside the plan and are found through device tables of
    addresses that are rewritten per sample."""
    assert torch.cuda.is_available(), "GPU tests selected but no ROCm device is visible"
    from patchmatchnet_amd.graph import PlannedForward
    model = _model()
    slot = PlannedForward(model)
    with torch.no_grad():
        for k in range(3):
            s = _sample(2, 96, 128, seed=40 + k)
            f = model.feature.forward_hip(s["images"])
            feats = [{st: t[j:j + 1].permute(0, 3, 1, 2) for st, t in f.items()} for j in range(3)]
            args = ([s["images"][0]] * 3, s["intrinsics"].clone(), s["extrinsics"], s["depth_min"], s["depth_max"])
            torch.manual_seed(7)
            want = model(*args, features=feats)
            torch.manual_seed(7)
            got = slot(*args, features=feats)
            torch.cuda.synchronize()
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), k
    assert slot.captures == 1


@pytest.mark.parametrize("H,W,n_src,steps", [(480, 640, 4, 60), (1200, 1600, 5, 48)])
def test_three_samples_in_flight_are_three_eager_forwards(H, W, n_src, steps):
    """bench.py's timed mode: three replay slots on three streams, the launch thread running ahead, inputs read in place -- every
    step's maps against the same step launched eagerly under the same seed.  Before lesson 46's fix 87-89 of 96 such steps differed at
    1600x1200, whether replayed as HIP graphs (round 5) or as plain launches (round 6's first session)."""
    assert torch.cuda.is_available(), "GPU tests selected but no ROCm device is visible"
    from patchmatchnet_amd.graph import PlannedForward
    model = _model()
    samples = [_sample(n_src, H, W, seed=60 + k) for k in range(4)]
    S = 3
    streams = [torch.cuda.Stream() for _ in range(S)]
    slots = [PlannedForward(model, inputs_in_place=True) for _ in range(S)]
    main = torch.cuda.current_stream()
    kept = []
    with torch.no_grad():
        for st in streams:
            st.wait_stream(main)
        for i in range(steps):
            torch.manual_seed(900 + i)
            with torch.cuda.stream(streams[i % S]):
                d, c = _call(slots[i % S], samples[i % len(samples)])
                kept.append((d.clone(), c.clone()))
        torch.cuda.synchronize()
        bad = []
        for i in range(steps):
            torch.manual_seed(900 + i)
            d, c = _call(model, samples[i % len(samples)])
            if not (torch.equal(d, kept[i][0]) and torch.equal(c, kept[i][1])):
                bad.append(i)
    assert not bad, f"{len(bad)} of {steps} steps differ from the eager forward: {bad[:10]}"


def test_a_forward_with_foreign_launches_is_refused():
    """MIOpen's FeatureNet (hip_feature_net = False) launches kernels this library does not own: such a forward cannot be a plan, and
    the recording pass says so instead of producing a plan that silently skips them."""
    assert torch.cuda.is_available(), "GPU tests selected but no ROCm device is visible"
    from patchmatchnet_amd import PmnError
    from patchmatchnet_amd.graph import PlannedForward
    model = _model()
    model.hip_feature_net = False
    s = _sample(2, 96, 128, seed=3)
    with torch.no_grad(), pytest.raises(PmnError, match="outside"):
        _call(PlannedForward(model), s)


def test_graph_replay_is_the_eager_forward_too():
    """Rounds 2-5's replay form (HIP graphs; bench.py --launch graph, eval.py --hip_graph 2), three slots in flight on the default
    hardware queues: round 5 measured 87 of 96 such steps wrong and blamed the graphs; with lesson 46's kernel fix they are right."""
    assert torch.cuda.is_available(), "GPU tests selected but no ROCm device is visible"
    from patchmatchnet_amd.graph import GraphedForward
    model = _model()
    samples = [_sample(4, 480, 640, seed=80 + k) for k in range(3)]
    S = 3
    streams = [torch.cuda.Stream() for _ in range(S)]
    slots = [GraphedForward(model, inputs_in_place=True) for _ in range(S)]
    kept = []
    with torch.no_grad():
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        for i in range(45):
            torch.manual_seed(300 + i)
            with torch.cuda.stream(streams[i % S]):
                d, c = _call(slots[i % S], samples[i % len(samples)])
                kept.append((d.clone(), c.clone()))
        torch.cuda.synchronize()
        bad = []
        for i in range(45):
            torch.manual_seed(300 + i)
            d, c = _call(model, samples[i % len(samples)])
            if not (torch.equal(d, kept[i][0]) and torch.equal(c, kept[i][1])):
                bad.append(i)
    assert not bad, f"{len(bad)} of 45 graph-replayed steps differ from the eager forward: {bad[:10]}"
