"""Seeded synthetic inputs shared by tests, smoke() and bench.py; no reference imports."""
from __future__ import annotations

import math
from typing import List

import numpy as np
import torch


def synthetic_cameras(n_views: int, H: int, W: int):
    """DTU-like pinhole cameras orbiting the point (0,0,650): returns (intrinsics [1,N,3,3], extrinsics [1,N,4,4])."""
    f = 2892.33 * W / 1600.0
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], np.float64)
    intr = np.stack([K] * n_views).astype(np.float32)
    extr = []
    P = np.array([0.0, 0.0, 650.0])
    for i in range(n_views):
        a = 0.0 if i == 0 else 0.08 * i * (1.0 if i % 2 else -1.0)
        R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
        E = np.eye(4)
        E[:3, :3] = R
        E[:3, 3] = P - R @ P
        extr.append(E)
    return intr[None], np.stack(extr).astype(np.float32)[None]


def synthetic_images(n_views: int, H: int, W: int, smooth: bool = True) -> List[torch.Tensor]:
    """Seeded images in [0,1].  ``smooth`` low-pass filters the noise and shifts it per view so the matching cost
    has structure; smooth=False is i.i.d. noise (the adversarial case for rounding parity)."""
    imgs = []
    for i in range(n_views):
        g = torch.Generator().manual_seed(i if not smooth else 0)
        img = torch.rand(1, 3, H, W, generator=g)
        if smooth:
            k = 9
            img = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(img, (k // 2,) * 4, mode="reflect"), k, 1)
            img = torch.roll(img, shifts=3 * i, dims=3)
            img = (img - img.min()) / (img.max() - img.min())
        imgs.append(img.contiguous())
    return imgs


def synthetic_features(n_views: int, C: int, h: int, w: int, seed: int = 0) -> List[torch.Tensor]:
    """Feature-map stand-ins [1,C,h,w]: smooth random fields, view i = view 0 shifted by 2i px plus 10% noise."""
    g = torch.Generator().manual_seed(seed)
    base = 0.5 * torch.randn(1, C, h + 8, w + 8, generator=g)
    base = torch.nn.functional.avg_pool2d(base, 5, 1, 2) * 3.0
    out = []
    for i in range(n_views):
        f = torch.roll(base, shifts=2 * i, dims=3)[:, :, 4:4 + h, 4:4 + w]
        f = f + 0.05 * torch.randn(f.shape, generator=g)
        out.append(f.contiguous())
    return out


def stage_projections(intr: np.ndarray, extr: np.ndarray, scale: float) -> np.ndarray:
    """models/net.py:225-229 in numpy: proj[:, :, :3, :4] = (K with rows 0,1 scaled) @ E[:3,:4]."""
    K = intr.astype(np.float32).copy()
    K[:, :, :2] *= np.float32(scale)
    proj = extr.astype(np.float32).copy()
    proj[:, :, :3, :4] = np.matmul(K, extr[:, :, :3, :4].astype(np.float32))
    return proj


def write_scan(root: str, scan: str, n_views: int, H: int, W: int, n_src: int = 2) -> str:
    """Writes a DTU-layout scan (images/*.jpg, cams/*_cam.txt, pair.txt) with the synthetic cameras above."""
    import os
    from PIL import Image
    d = os.path.join(root, scan)
    os.makedirs(os.path.join(d, "images"), exist_ok=True)
    os.makedirs(os.path.join(d, "cams"), exist_ok=True)
    intr, extr = synthetic_cameras(n_views, H, W)
    imgs = synthetic_images(n_views, H, W)
    for v in range(n_views):
        arr = (imgs[v][0].permute(1, 2, 0).numpy() * 255).astype(np.uint8)
        Image.fromarray(arr).save(os.path.join(d, "images", "{:0>8}.jpg".format(v)), quality=95)
        with open(os.path.join(d, "cams", "{:0>8}_cam.txt".format(v)), "w") as f:
            f.write("extrinsic\n")
            for r in extr[0, v]:
                f.write(" ".join("%.8f" % x for x in r) + "\n")
            f.write("\nintrinsic\n")
            for r in intr[0, v]:
                f.write(" ".join("%.8f" % x for x in r) + "\n")
            f.write("\n425.0 935.0\n")
    with open(os.path.join(d, "pair.txt"), "w") as f:
        f.write("%d\n" % n_views)
        for v in range(n_views):
            others = [u for u in range(n_views) if u != v][:max(n_src, 1)]
            f.write("%d\n%d " % (v, len(others)) + " ".join("%d %.2f" % (u, 100.0 - u) for u in others) + "\n")
    return d
