#!/usr/bin/env python
"""Attribution probe behind tests/test_fullsize_parity.py::test_cfg2_scene_against_the_reference_on_rocm: the reference on ROCm
(pinned-draw archive) vs this engine fed the archive's own FeatureNet outputs, with pieces of this engine's projection arithmetic
swapped for what the reference's GPU path runs:
    default                 pmn_stage_projections (fp64 inverse) + the CPU reference's mul/add order for rot @ [x y 1]^T
    --torch-projections     src_proj @ inverse(ref_proj) through torch on ROCm (rocSOLVER / rocBLAS), as the reference does there
    --lib <POSE_FMA build>  rot @ [x y 1]^T as a k-ordered fma chain (a GPU BLAS's order)
One process per library.  Prints one JSON line."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--tag", default="default")
a = ap.parse_args()
import torch  # noqa: E402
if a.lib:
    from patchmatchnet_amd import _lib
    _lib.LIB_PATH = os.path.abspath(a.lib)
import goldenutil as GU  # noqa: E402
import synth  # noqa: E402
import patchmatchnet_amd as P  # noqa: E402

DEV = "cuda:0"
_, params, kw = GU.load_case("default")
model = P.PatchmatchNet(**kw)
model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
model = model.to(DEV).eval()
g = GU.load_npz("cfg2_scene.npz")
H, W, nv = int(g["H"]), int(g["W"]), int(g["n_views"])
imgs, intr, extr, gt = synth.render_scene(nv, H, W, int(g["scene_seed"]))
noise = torch.rand(1, 48, H // 8, W // 8, generator=torch.Generator().manual_seed(int(g["noise_seed"]))).to(DEV)
dimgs = [im.to(DEV) for im in imgs]
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)  # noqa: E731
dmin, dmax = torch.tensor([425.0], device=DEV), torch.tensor([935.0], device=DEV)
ref = torch.jit.load(os.path.join(ROOT, "oracle", "_ref", "patchmatchnet_reference_pinned.pt"), map_location=DEV).eval()
ref.patchmatch_3.depth_initialization.noise = noise


def stats(got, want):
    got, want = got.double().flatten(), want.double().flatten()
    rel = (got - want).abs() / want.abs()
    return {"frac_over_1e-3": float((rel > 1e-3).double().mean()), "frac_over_1e-4": float((rel > 1e-4).double().mean()),
            "max": float(rel.max())}


out = {"tag": a.tag}
with torch.no_grad():
    for _ in range(2):
        b_depth, _, b_dpm = ref([im.clone() for im in dimgs], t(intr), t(extr), dmin, dmax)
    feats = [{s: f.contiguous() for s, f in ref.feature(im).items()} for im in dimgs]
    for name, torch_proj in (("engine_projections", False), ("torch_rocm_projections", True)):
        model.hip_projections = not torch_proj
        d_depth, _, d_dpm = model([im.clone() for im in dimgs], t(intr), t(extr), dmin, dmax, noise=noise, features=feats)
        r = {f"s{st}_it{it + 1}": stats(d, b_dpm[st][it]) for st in (3, 2, 1) for it, d in enumerate(d_dpm[st])}
        r["final"] = stats(d_depth, b_depth)
        out[name] = r
    # the relative projections themselves: this engine's (fp64 inverse) vs torch's on ROCm vs torch's on the CPU
    from patchmatchnet_amd import ops
    E, K = t(extr), t(intr)
    rel_engine = ops.stage_projections(K, E, 3, 0.125)[0]
    Ks = K.clone(); Ks[:, :, :2] *= 0.125
    proj = E.clone(); proj[:, :, :3, :4] = torch.matmul(Ks, E[:, :, :3, :4])
    rel_rocm = torch.matmul(proj[:, 1:], torch.inverse(proj[:, 0]).unsqueeze(1))
    pc = proj.cpu()
    rel_cpu = torch.matmul(pc[:, 1:], torch.inverse(pc[:, 0]).unsqueeze(1))
    sc = rel_cpu.abs().amax(dim=(-1, -2), keepdim=True)
    out["rel_proj_max_diff_over_scale"] = {"engine_vs_torch_cpu": float(((rel_engine.cpu() - rel_cpu).abs() / sc).max()),
                                           "torch_rocm_vs_torch_cpu": float(((rel_rocm.cpu() - rel_cpu).abs() / sc).max()),
                                           "engine_vs_torch_rocm": float(((rel_engine.cpu() - rel_rocm.cpu()).abs() / sc).max())}
print(json.dumps(out))
