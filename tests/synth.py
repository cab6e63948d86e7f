"""Seeded synthetic inputs shared by tests, smoke() and bench.py; no reference imports."""
from __future__ import annotations

import math
from typing import List

import numpy as np
import torch


def synthetic_cameras(n_views: int, H: int, W: int, step: float = 0.08):
    """DTU-like pinhole cameras orbiting the point (0,0,650): returns (intrinsics [1,N,3,3], extrinsics [1,N,4,4]).  ``step`` = the
    angle between neighbouring views in radians (view i sits at +- step * i)."""
    f = 2892.33 * W / 1600.0
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], np.float64)
    intr = np.stack([K] * n_views).astype(np.float32)
    extr = []
    P = np.array([0.0, 0.0, 650.0])
    for i in range(n_views):
        a = 0.0 if i == 0 else step * i * (1.0 if i % 2 else -1.0)
        R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
        E = np.eye(4)
        E[:3, :3] = R
        E[:3, 3] = P - R @ P
        extr.append(E)
    return intr[None], np.stack(extr).astype(np.float32)[None]


def general_cameras(n_views: int, H: int, W: int):
    """A rig WITHOUT the symmetries of ``synthetic_cameras`` (rotations about the y axis only: there the relative projections have
    exact zeros where a general pose has none -- the y coefficients of x and z -- and every epipolar line is horizontal): each source
    camera is rotated about its own tilted axis (so it also rolls and pitches), sits off the reference's baseline in x, y AND z, and
    has its own focal lengths (fx != fy) and principal point.  Same contract: (intrinsics [1,N,3,3], extrinsics [1,N,4,4]); the
    cameras look at the point (0,0,650)."""
    intr, extr = [], []
    P = np.array([0.0, 0.0, 650.0])
    for i in range(n_views):
        s = 1.0 + 0.04 * i * (1 if i % 2 else -1)
        fx, fy = 2892.33 * W / 1600.0 * s, 2892.33 * W / 1600.0 * (2.0 - s) * 1.01
        K = np.array([[fx, 0, W / 2.0 + 3.0 * i], [0, fy, H / 2.0 - 2.0 * i], [0, 0, 1]], np.float64)
        if i == 0:
            R = np.eye(3)
            C = np.zeros(3)
        else:
            axis = np.array([0.35 * (1 if i % 2 else -1), 1.0, 0.25 * (-1) ** (i // 2)])
            axis /= np.linalg.norm(axis)
            a = 0.07 * i * (1.0 if i % 2 else -1.0)
            Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
            R = np.eye(3) + math.sin(a) * Kx + (1 - math.cos(a)) * (Kx @ Kx)  # Rodrigues
            roll = 0.05 * i
            Rz = np.array([[math.cos(roll), -math.sin(roll), 0], [math.sin(roll), math.cos(roll), 0], [0, 0, 1]])
            R = Rz @ R
            C = np.array([0.0, 12.0 * i * (-1) ** i, -8.0 * i])  # off the orbit: up / down and towards / away from the scene
        E = np.eye(4)
        E[:3, :3] = R
        E[:3, 3] = (P - R @ P) - R @ C
        intr.append(K)
        extr.append(E)
    return np.stack(intr).astype(np.float32)[None], np.stack(extr).astype(np.float32)[None]


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


def arc_cameras(n_views: int, H: int, W: int, step: float = 0.015):
    """A DTU-like capture: n_views cameras on an arc around the point (0,0,650), ``step`` rad apart about the y axis with a small
    nod about x, all looking at the surface of ``render_scene`` (neighbouring views overlap almost completely)."""
    f = 2892.33 * W / 1600.0
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], np.float64)
    P = np.array([0.0, 0.0, 650.0])
    extr = []
    for i in range(n_views):
        a, b = step * (i - n_views // 2), 0.03 * math.sin(0.7 * i)
        Ry = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
        Rx = np.array([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
        R = Rx @ Ry
        E = np.eye(4)
        E[:3, :3] = R
        E[:3, 3] = P - R @ P
        extr.append(E)
    return np.stack([K] * n_views).astype(np.float32)[None], np.stack(extr).astype(np.float32)[None]


def write_scene_scan(root: str, scan: str, n_views: int, H: int, W: int, n_src: int = 10, seed: int = 0, device="cpu",
                     quality: int = 95) -> str:
    """A DTU-layout scan of the photo-consistent scene: ``n_views`` JPEGs rendered from ``arc_cameras``, camera files, and a
    pair.txt that lists every view's ``n_src`` nearest neighbours on the arc (nearest first, like a DTU pair file)."""
    import os
    from PIL import Image
    d = os.path.join(root, scan)
    os.makedirs(os.path.join(d, "images"), exist_ok=True)
    os.makedirs(os.path.join(d, "cams"), exist_ok=True)
    cams = arc_cameras(n_views, H, W)
    imgs, intr, extr, _ = render_scene(n_views, H, W, seed=seed, device=device, cameras=cams)
    for v in range(n_views):
        arr = (imgs[v][0].permute(1, 2, 0).cpu().numpy() * 255).round().astype(np.uint8)
        Image.fromarray(arr).save(os.path.join(d, "images", "{:0>8}.jpg".format(v)), quality=quality)
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
            others = sorted((u for u in range(n_views) if u != v), key=lambda u: (abs(u - v), u))[:max(n_src, 1)]
            f.write("%d\n%d " % (v, len(others)) + " ".join("%d %.2f" % (u, 100.0 - abs(u - v)) for u in others) + "\n")
    return d


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


# ---- photo-consistent scene ---------------------------------------------------------------
# An analytic height field z = f(X, Y) in the reference camera's frame (= world: the reference extrinsic is the identity),
# textured by a band-limited procedural pattern defined ON THE SURFACE, and rendered into every camera by inverse warping
# (per-pixel ray / surface intersection, Newton).  Every view therefore sees the same surface colours: the matching cost along
# a pixel's ray has ONE mode at the true depth, unlike the rolled-noise images of ``synthetic_images`` whose content is
# unrelated to the cameras.  Images are quantised to k/255 like decoded JPEG / PNG data, which also makes them reproducible
# across hosts (libm / SIMD sin() differences of an ulp cannot move a rounding except with probability ~1e-13 per pixel);
# ``scene_digest`` is stored in the golden so that a host that renders something else fails loudly.

SCENE_Z0, SCENE_AMP, SCENE_LX, SCENE_LY, SCENE_TILT = 650.0, 25.0, 260.0, 210.0, 0.10


def scene_height(X, Y):
    """Height field and its gradient: z = z0 + A sin(2 pi X / Lx) cos(2 pi Y / Ly) + tilt * X."""
    kx, ky = 2.0 * math.pi / SCENE_LX, 2.0 * math.pi / SCENE_LY
    sx, cx, sy, cy = torch.sin(kx * X), torch.cos(kx * X), torch.sin(ky * Y), torch.cos(ky * Y)
    z = SCENE_Z0 + SCENE_AMP * sx * cy + SCENE_TILT * X
    return z, SCENE_AMP * kx * cx * cy + SCENE_TILT, -SCENE_AMP * ky * sx * sy


def scene_texture(X, Y, seed: int, n_waves: int = 40):
    """[3, ...] colours in [0,1]: per channel a sum of plane waves on the surface coordinates (wavelengths 1.5 - 80 mm,
    log-uniform, random directions / phases); band-limited, so sampling it at 0.22 mm per pixel is alias-free."""
    g = torch.Generator().manual_seed(10007 * seed + 17)
    lam = 1.5 * (80.0 / 1.5) ** torch.rand(n_waves, generator=g, dtype=torch.float64)
    ang = 2.0 * math.pi * torch.rand(n_waves, generator=g, dtype=torch.float64)
    phase = 2.0 * math.pi * torch.rand(3, n_waves, generator=g, dtype=torch.float64)
    amp = (0.6 + 0.8 * torch.rand(3, n_waves, generator=g, dtype=torch.float64)) * (0.2 / math.sqrt(n_waves / 2.0))
    fx, fy = (2.0 * math.pi / lam * torch.cos(ang)).to(X.device), (2.0 * math.pi / lam * torch.sin(ang)).to(X.device)
    phase, amp = phase.to(X.device), amp.to(X.device)
    out = torch.full((3,) + tuple(X.shape), 0.5, dtype=torch.float64, device=X.device)
    for k in range(n_waves):
        arg = fx[k] * X + fy[k] * Y
        s, c = torch.sin(arg), torch.cos(arg)
        for ch in range(3):  # sin(arg + phase) from one sin / cos pair per wave
            out[ch] += amp[ch, k] * (s * math.cos(float(phase[ch, k])) + c * math.sin(float(phase[ch, k])))
    return out.clamp_(0.0, 1.0)


def render_scene(n_views: int, H: int, W: int, seed: int = 0, device="cpu", quantise: bool = True, all_depths: bool = False,
                 cameras=None):
    """Photo-consistent synthetic sample for ``synthetic_cameras(n_views, H, W)``.

    Returns (images: n_views x [1,3,H,W] float32 in {k/255}, intrinsics [1,N,3,3], extrinsics [1,N,4,4],
    depth_gt [H,W] float32 = the surface's depth in the reference view (view 0); with ``all_depths`` a list of the n_views
    ground-truth depth maps, each in its own camera -- a geometrically consistent set of maps for the fusion tests)."""
    intr, extr = synthetic_cameras(n_views, H, W) if cameras is None else cameras  # cameras: ([1,N,3,3], [1,N,4,4]) to override
    dev = torch.device(device)
    v, u = torch.meshgrid(torch.arange(H, dtype=torch.float64, device=dev), torch.arange(W, dtype=torch.float64, device=dev),
                          indexing="ij")
    images, depth_gt, depths = [], None, []
    for i in range(n_views):
        K = torch.from_numpy(intr[0, i].astype(np.float64))
        E = torch.from_numpy(extr[0, i].astype(np.float64))
        R, t = E[:3, :3], E[:3, 3]
        Cc = -(R.T @ t)  # camera centre in world coordinates
        Kinv = torch.linalg.inv(K)
        M = R.T @ Kinv   # ray direction in world coordinates = M @ (u, v, 1)
        dx = float(M[0, 0]) * u + float(M[0, 1]) * v + float(M[0, 2])
        dy = float(M[1, 0]) * u + float(M[1, 1]) * v + float(M[1, 2])
        dz = float(M[2, 0]) * u + float(M[2, 1]) * v + float(M[2, 2])
        s = (SCENE_Z0 - float(Cc[2])) / dz
        for _ in range(10):  # Newton on g(s) = C_z + s d_z - f(x(s), y(s)); |grad f . d_xy / d_z| < 1 on this rig
            X, Y = float(Cc[0]) + s * dx, float(Cc[1]) + s * dy
            z, fx, fy = scene_height(X, Y)
            s = s - (float(Cc[2]) + s * dz - z) / (dz - fx * dx - fy * dy)
        X, Y = float(Cc[0]) + s * dx, float(Cc[1]) + s * dy
        img = scene_texture(X, Y, seed)
        if quantise:
            img = torch.round(img * 255.0) / 255.0
        images.append(img.to(torch.float32)[None].contiguous())
        if i == 0:
            depth_gt = (float(Cc[2]) + s * dz).to(torch.float32)  # E_0 = I: camera depth = world z
        if all_depths:  # depth in camera i = third row of E_i applied to the surface point
            Z = float(Cc[2]) + s * dz
            depths.append((float(R[2, 0]) * X + float(R[2, 1]) * Y + float(R[2, 2]) * Z + float(t[2])).to(torch.float32))
    return images, intr, extr, (depths if all_depths else depth_gt)


def scene_digest(images) -> str:
    """sha256 over the uint8 form of the rendered views (what the golden was computed on)."""
    import hashlib
    h = hashlib.sha256()
    for im in images:
        h.update(torch.round(im.detach().cpu() * 255.0).to(torch.uint8).numpy().tobytes())
    return h.hexdigest()


def scene_signature(images, step: int = 16):
    """A signature of the rendered views that tolerates the last-bit differences between hosts: every ``step``-th pixel of the uint8
    images and the per-view sum of all bytes.  (The sha256 of scene_digest flips when ONE of the 2e8 byte values of an 11-view
    3072x2048 scene lands on the other side of a rounding boundary: float64 sin / cos are not bit-identical across CPU generations.)"""
    u8 = [torch.round(im.detach().cpu() * 255.0).to(torch.uint8)[0] for im in images]
    thumb = np.stack([u[:, ::step, ::step].numpy() for u in u8], 0)
    sums = np.asarray([int(u.to(torch.int64).sum()) for u in u8], np.int64)
    return thumb, sums


def scene_matches(images, thumb, sums, step: int = 16):
    """(ok, note): the rendered views equal the fixture's up to isolated one-LSB differences (< 1e-4 of the sampled bytes, never more
    than one level; per-view byte sums within 1e-6)."""
    t2, s2 = scene_signature(images, step)
    if t2.shape != thumb.shape:
        return False, f"thumbnail shape {t2.shape} != {thumb.shape}"
    d = np.abs(t2.astype(np.int16) - thumb.astype(np.int16))
    frac = float((d > 0).mean())
    rel = float(np.abs(s2 - sums).max() / max(float(sums.max()), 1.0))
    ok = int(d.max()) <= 1 and frac < 1e-4 and rel < 1e-6
    return ok, f"sampled bytes that differ: {frac:.2e} (max {int(d.max())} level), per-view byte sums within {rel:.1e}"
