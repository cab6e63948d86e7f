"""Live check of the oracle against the imported reference (authoring container only; skipped where
/root/reference is absent, e.g. the GPU box -- the committed golden fixtures cover that case)."""
import numpy as np
import pytest
import torch

import goldenutil as GU
import refutil
from oracle import oracle as O

pytestmark = pytest.mark.skipif(not refutil.have_reference(), reason="reference checkout not present")


def test_cascade_160x128_n2_matches_reference():
    """BASELINE config 0 shape (160x128, 2 source views; default iterations) end to end through the hot path."""
    torch.set_num_threads(4)
    H, W, NV = 128, 160, 3
    model = refutil.build_reference_model()
    imgs = refutil.synthetic_images(NV, H, W)
    intr, extr = refutil.synthetic_cameras(NV, H, W)
    dmin, dmax = np.array([425.0], np.float32), np.array([935.0], np.float32)
    noise = torch.rand(1, 48, H // 8, W // 8, generator=torch.Generator().manual_seed(5))
    depth, conf, dpm, tr = refutil.trace_reference_forward(
        model, [i.clone() for i in imgs], torch.from_numpy(intr).clone(), torch.from_numpy(extr).clone(),
        torch.from_numpy(dmin), torch.from_numpy(dmax), noise)
    params = refutil.state_dict_numpy(model)
    feats = [{k: v.numpy() for k, v in f.items()} for f in tr["features"]]
    d1, score, out = O.cascade(params, feats, intr, extr, dmin, dmax, noise.numpy())
    assert GU.rel_err(d1, dpm[1][-1].numpy()) < 1e-4
    c, _ = O.confidence(score, (H, W))
    assert float((np.abs(c - conf.numpy()) > 1e-3).mean()) < 5e-3


def test_golden_fixtures_are_current():
    """The committed weights fixture equals the reference checkpoint bit for bit."""
    sd = refutil.load_reference_state_dict()
    g = GU.load_npz("params_000007.npz")
    assert set(g) == set(sd)
    for k, v in sd.items():
        np.testing.assert_array_equal(g[k], v.numpy())
