"""Helpers that import the read-only reference checkout (``/root/reference``) to produce oracle-pinning data.

Only usable in the authoring container: the GPU box has no ``/root/reference``.  Nothing here is imported by
``-m gpu`` tests, ``smoke()`` or ``bench.py`` -- those consume the committed fixtures in ``tests/golden/``.
"""
from __future__ import annotations

import os
import sys
from contextlib import contextmanager
from typing import Dict, List

import numpy as np
import torch

REFERENCE_ROOT = os.environ.get("PMN_REFERENCE_ROOT", "/root/reference")
CHECKPOINT = os.path.join(REFERENCE_ROOT, "checkpoints", "params_000007.ckpt")


def have_reference() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "patchmatch.py")) and os.path.isfile(CHECKPOINT)


def import_reference():
    """Returns the reference's ``models`` package (imported, never copied)."""
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import models.module as ref_module  # noqa: F401
    import models.net as ref_net
    import models.patchmatch as ref_pm
    return ref_net, ref_pm, ref_module


def import_reference_eval():
    """The reference's eval.py (filter_depth and friends), imported by path with its absent third-party imports stubbed:
    cv2 (remap = the oracle's restatement of cv2.remap INTER_LINEAR), plyfile, torchvision, tensorboard.  Everything else that
    runs is the reference's own numpy code."""
    import importlib.util
    import types
    from oracle import fusion_oracle as FO

    saved = {k: sys.modules.get(k) for k in ("cv2", "plyfile", "torchvision", "torchvision.utils", "torch.utils.tensorboard",
                                             "utils", "datasets", "datasets.data_io", "datasets.mvs")}

    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    stub("cv2", INTER_LINEAR=1, remap=lambda src, mx, my, interpolation=None: FO.remap_linear_cv2(src, mx, my))
    stub("plyfile", PlyData=object, PlyElement=object)
    tv = stub("torchvision")
    tv.utils = stub("torchvision.utils")
    stub("torch.utils.tensorboard", SummaryWriter=object)
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        spec = importlib.util.spec_from_file_location("pmn_reference_eval", os.path.join(REFERENCE_ROOT, "eval.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.path.remove(REFERENCE_ROOT)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def load_reference_state_dict() -> Dict[str, torch.Tensor]:
    sd = torch.load(CHECKPOINT, map_location="cpu")["model"]
    return {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}


def build_reference_model(**overrides):
    ref_net, _, _ = import_reference()
    kw = dict(patchmatch_interval_scale=[0.005, 0.0125, 0.025], propagation_range=[6, 4, 2],
              patchmatch_iteration=[1, 2, 2], patchmatch_num_sample=[8, 8, 16], propagate_neighbors=[0, 8, 16],
              evaluate_neighbors=[9, 9, 9])
    kw.update(overrides)
    model = ref_net.PatchmatchNet(**kw)
    missing, unexpected = model.load_state_dict(load_reference_state_dict(), strict=True)
    assert not missing and not unexpected
    return model.eval()


from synth import synthetic_cameras, synthetic_images  # noqa: E402,F401  (shared seeded generators)


@contextmanager
def injected_rand(noise: torch.Tensor):
    """Pin the stage-3 ``torch.rand`` draw (patchmatch.py:61-62) to ``noise``."""
    orig = torch.rand

    def fake(*args, **kwargs):
        size = kwargs.get("size", args[0] if args else None)
        if size is not None and tuple(size) == tuple(noise.shape):
            return noise.clone().to(kwargs.get("device", "cpu"))
        return orig(*args, **kwargs)

    torch.rand = fake
    try:
        yield
    finally:
        torch.rand = orig


def trace_reference_forward(model, images, intrinsics, extrinsics, depth_min, depth_max, noise):
    """Runs the reference forward and records the hot-path intermediates via hooks.

    Returns (depth, confidence, depth_patchmatch, trace) with trace[stage] = list of per-iteration dicts plus
    trace['features'] = list (per view) of {stage: tensor}."""
    trace: Dict = {"features": []}
    handles = []

    def feat_hook(_m, _inp, out):
        trace["features"].append({k: v.detach().clone() for k, v in out.items()})

    handles.append(model.feature.register_forward_hook(feat_hook))
    for stage in (1, 2, 3):
        pm = getattr(model, f"patchmatch_{stage}")
        recs: List[Dict] = []
        trace[stage] = recs
        per_stage: Dict = {}
        trace[f"stage{stage}"] = per_stage

        def conv_hook(name, per_stage=per_stage):
            def h(_m, _inp, out):
                per_stage[name] = out.detach().clone()
            return h

        handles.append(pm.propa_conv.register_forward_hook(conv_hook("propa_offsets")))
        handles.append(pm.eval_conv.register_forward_hook(conv_hook("eval_offsets")))
        handles.append(pm.feature_weight_net.register_forward_hook(conv_hook("feature_weight")))

        def eval_pre(_m, args, kwargs, recs=recs):
            recs.append({"depth_sample": kwargs["depth_sample"].detach().clone(),
                         "weight": kwargs["weight"].detach().clone(),
                         "grid": kwargs["grid"].detach().clone(),
                         "view_weights_in": kwargs["view_weights"].detach().clone(),
                         "pixelwise_in": []})

        def eval_post(_m, args, kwargs, out, recs=recs):
            recs[-1]["depth"] = out[0].detach().clone()
            recs[-1]["score"] = out[1].detach().clone()
            recs[-1]["view_weights"] = out[2].detach().clone()

        def sim_pre(_m, args, recs=recs):
            recs[-1]["similarity"] = args[0].detach().clone()

        def sim_post(_m, args, out, recs=recs):
            recs[-1]["score_pre_softmax"] = out.detach().clone()

        def pix_pre(_m, args, recs=recs):
            recs[-1]["pixelwise_in"].append(args[0].detach().clone())

        handles.append(pm.evaluation.register_forward_pre_hook(eval_pre, with_kwargs=True))
        handles.append(pm.evaluation.register_forward_hook(eval_post, with_kwargs=True))
        handles.append(pm.evaluation.similarity_net.register_forward_pre_hook(sim_pre))
        handles.append(pm.evaluation.similarity_net.register_forward_hook(sim_post))
        handles.append(pm.evaluation.pixel_wise_net.register_forward_pre_hook(pix_pre))
    try:
        with torch.no_grad(), injected_rand(noise):
            depth, conf, dpm = model(images, intrinsics, extrinsics, depth_min, depth_max)
    finally:
        for h in handles:
            h.remove()
    return depth, conf, dpm, trace


def state_dict_numpy(model) -> Dict[str, np.ndarray]:
    return {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
