#!/usr/bin/env python
"""TEST INFRASTRUCTURE (the checker / the timed baseline, never the product): builds oracle/_ref/ from the reference where it lies.

The reference (FangjinhuaWang/PatchmatchNet) is pure Python, so there is nothing to compile -- but its model is TorchScript-able
(the reference's own eval.py loads such an archive with ``--input_type module``, eval.py:37-41, and ships one as
checkpoints/module_000007.pt).  This recipe imports the UNMODIFIED reference ``models/net.py`` read-only from /root/reference,
builds ``PatchmatchNet`` with the reference's default evaluation arguments (eval.py:300-330) and ``checkpoints/params_000007.ckpt``,
scripts it with ``torch.jit.script`` and writes

    oracle/_ref/patchmatchnet_reference.pt        the reference's own op graph + weights (git-ignored, travels with gpurun)
    oracle/_ref/patchmatchnet_reference.json      how it was made + a known-answer (input seed -> output digest) for the loader

No reference source is copied into the repository: the archive is a build output, like a compiled .so, and it is listed in
.gitignore.  ``bench.py``'s cpu_baseline leg loads it on the bench box (where /root/reference does not exist) to time the REAL
reference on the host cores (kind "reference") and on the MI355X through PyTorch-ROCm (the north star's 4x denominator);
tests/test_reference_archive.py checks it against tests/golden (outputs of the imported reference) wherever it exists.

    python oracle/make_ref.py [--reference /root/reference] [--force]
"""
import argparse
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT_DIR = os.path.join(ROOT, "oracle", "_ref")
ARCHIVE = os.path.join(OUT_DIR, "patchmatchnet_reference.pt")
PINNED = os.path.join(OUT_DIR, "patchmatchnet_reference_pinned.pt")  # same network, stage-3 draw read from a buffer (see pinned_draw)
META = os.path.join(OUT_DIR, "patchmatchnet_reference.json")

DEFAULT_KW = dict(patchmatch_interval_scale=[0.005, 0.0125, 0.025], propagation_range=[6, 4, 2],
                  patchmatch_iteration=[1, 2, 2], patchmatch_num_sample=[8, 8, 16], propagate_neighbors=[0, 8, 16],
                  evaluate_neighbors=[9, 9, 9])


def known_answer_inputs(n_views=3, H=64, W=80):
    """Small seeded inputs (cfg-1 class): images in [0,1], DTU-like cameras.  Used for the archive's known answer."""
    import numpy as np
    import torch
    g = torch.Generator().manual_seed(20260926)
    imgs = [torch.rand(1, 3, H, W, generator=g) for _ in range(n_views)]
    intr = np.zeros((1, n_views, 3, 3), np.float32)
    extr = np.zeros((1, n_views, 4, 4), np.float32)
    for v in range(n_views):
        intr[0, v] = [[0.9 * W, 0, W / 2], [0, 0.9 * W, H / 2], [0, 0, 1]]
        extr[0, v] = np.eye(4)
        extr[0, v, 0, 3] = -40.0 * v
    return imgs, torch.from_numpy(intr), torch.from_numpy(extr), torch.tensor([425.0]), torch.tensor([935.0])


def pinned_draw(inner):
    """Wraps the reference's DepthInitialization module so that the stage-3 random draw can be HANDED IN: the one line of the
    reference that keeps two runs on different devices from being comparable is ``torch.rand(..., device=device)``
    (models/patchmatch.py:61-62) -- the CPU and the ROCm generator produce different streams under the same seed.  The wrapper reads
    the uniform draw from its ``noise`` attribute ([B,48,h,w], assigned by the caller on the loaded archive) and maps it to the 48
    inverse-depth bins as models/patchmatch.py:63-71 does; every other call goes to the unmodified reference module.  Pinned to the
    unmodified archive bit for bit in tests/test_reference_archive.py (same CPU draw handed in == the seed's own draw)."""
    import torch

    class PinnedDraw(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner = inner
            self.register_buffer("noise", torch.zeros(1, 48, 1, 1), persistent=False)

        def forward(self, min_depth: torch.Tensor, max_depth: torch.Tensor, height: int, width: int,
                    depth_interval_scale: float, device: torch.device, depth: torch.Tensor) -> torch.Tensor:
            if depth.numel() == 0:
                nb = min_depth.size()[0]
                bins = self.noise + torch.arange(start=0, end=48, step=1, device=device).view(1, 48, 1, 1)
                far = (1.0 / max_depth).view(nb, 1, 1, 1)
                near = (1.0 / min_depth).view(nb, 1, 1, 1)
                return 1.0 / (far + bins / 48 * (near - far))
            return self.inner(min_depth, max_depth, height, width, depth_interval_scale, device, depth)

    return PinnedDraw(inner)


def build(reference="/root/reference", force=False, verbose=True):
    """Returns the archive path, or None when the reference checkout is not there (the GPU box: the prebuilt file is used)."""
    if not os.path.isdir(os.path.join(reference, "models")):
        return ARCHIVE if os.path.isfile(ARCHIVE) else None
    if os.path.isfile(ARCHIVE) and os.path.isfile(META) and os.path.isfile(PINNED) and not force:
        return ARCHIVE
    import torch
    sys.path.insert(0, reference)
    try:
        from models.net import PatchmatchNet  # the reference's own class, imported where it lies
    finally:
        sys.path.remove(reference)
    model = PatchmatchNet(**DEFAULT_KW)
    ckpt = os.path.join(reference, "checkpoints", "params_000007.ckpt")
    sd = torch.load(ckpt, map_location="cpu")["model"]
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
    missing = model.load_state_dict(sd, strict=True)
    model.eval()
    scripted = torch.jit.script(model)
    os.makedirs(OUT_DIR, exist_ok=True)
    scripted.save(ARCHIVE)
    # the same network with the stage-3 draw handed in (CPU <-> ROCm comparisons of the reference with itself need one draw)
    model.patchmatch_3.depth_initialization = pinned_draw(model.patchmatch_3.depth_initialization)
    torch.jit.script(model).save(PINNED)
    # known answer on the CPU backend (seeded stage-3 draw), for the loader on the other side
    imgs, intr, extr, dmin, dmax = known_answer_inputs()
    torch.manual_seed(1234)
    with torch.no_grad():
        depth, conf, _ = torch.jit.load(ARCHIVE, map_location="cpu")(imgs, intr, extr, dmin, dmax)
    meta = dict(
        made_by="oracle/make_ref.py", reference=reference, checkpoint="checkpoints/params_000007.ckpt", kwargs=DEFAULT_KW,
        torch=torch.__version__, state_dict_load=str(missing),
        archive_sha256=hashlib.sha256(open(ARCHIVE, "rb").read()).hexdigest(),
        known_answer=dict(seed=1234, depth_mean=float(depth.double().mean()), depth_std=float(depth.double().std()),
                          confidence_mean=float(conf.double().mean()), shape=list(depth.shape)))
    with open(META, "w") as f:
        json.dump(meta, f, indent=1)
    if verbose:
        print(f"wrote {ARCHIVE} ({os.path.getsize(ARCHIVE)} bytes) and {META}")
    return ARCHIVE


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--force", action="store_true")
    a = ap.parse_args()
    p = build(a.reference, a.force)
    print(p if p else "no reference checkout and no prebuilt archive")
