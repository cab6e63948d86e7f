"""Photometric + geometric consistency filtering and point-cloud fusion of the predicted maps (reference
eval.py:86-297), run on the device with torch ops in float64 (the reference does it in single-threaded numpy + cv2).

SURVEY.md 8(f) row 2 ("next"): same warp + bilinear-gather pattern as the hot path, but it consumes the per-scan maps
(all-gathered over RCCL when reference views are sharded across GPUs) instead of re-reading them from disk.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def _remap_bilinear(src: torch.Tensor, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """cv2.remap(src, x, y, INTER_LINEAR) with the default constant(0) border == bilinear / zeros grid_sample on pixel
    coordinates.  src [H,W]; x,y [H,W] pixel positions."""
    H, W = src.shape
    gx = x / ((W - 1) / 2) - 1
    gy = y / ((H - 1) / 2) - 1
    grid = torch.stack((gx, gy), dim=-1).unsqueeze(0).to(src.dtype)
    return F.grid_sample(src[None, None], grid, mode="bilinear", padding_mode="zeros", align_corners=True)[0, 0]


def reproject_with_depth(depth_ref, K_ref, E_ref, depth_src, K_src, E_src):
    """Reference eval.py:86-145 (all [H,W] / [3,3] / [4,4] float64 tensors on one device)."""
    H, W = depth_ref.shape
    dev, dt = depth_ref.device, depth_ref.dtype
    y_ref, x_ref = torch.meshgrid(torch.arange(H, device=dev, dtype=dt), torch.arange(W, device=dev, dtype=dt),
                                  indexing="ij")
    ones = torch.ones(H * W, device=dev, dtype=dt)
    pix = torch.stack((x_ref.reshape(-1), y_ref.reshape(-1), ones))
    xyz_ref = torch.linalg.inv(K_ref) @ (pix * depth_ref.reshape(-1))
    xyz_src = (E_src @ torch.linalg.inv(E_ref) @ torch.cat((xyz_ref, ones[None])))[:3]
    k_src = K_src @ xyz_src
    xy_src = k_src[:2] / k_src[2:3]
    x_src = xy_src[0].reshape(H, W).float().to(dt)  # the reference rounds the map coordinates to float32
    y_src = xy_src[1].reshape(H, W).float().to(dt)
    sampled = _remap_bilinear(depth_src, x_src, y_src)
    xyz_src2 = torch.linalg.inv(K_src) @ (torch.cat((xy_src, ones[None])) * sampled.reshape(-1))
    xyz_rep = (E_ref @ torch.linalg.inv(E_src) @ torch.cat((xyz_src2, ones[None])))[:3]
    depth_rep = xyz_rep[2].reshape(H, W)
    k_rep = K_ref @ xyz_rep
    xy_rep = k_rep[:2] / k_rep[2:3]
    return depth_rep, xy_rep[0].reshape(H, W), xy_rep[1].reshape(H, W), x_ref, y_ref


def check_geometric_consistency(depth_ref, K_ref, E_ref, depth_src, K_src, E_src, geo_pixel_thres: float,
                                geo_depth_thres: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """Reference eval.py:148-190 -> (mask [H,W] bool, reprojected depth with inconsistent pixels zeroed)."""
    depth_rep, x_rep, y_rep, x_ref, y_ref = reproject_with_depth(depth_ref, K_ref, E_ref, depth_src, K_src, E_src)
    dist = torch.sqrt((x_rep - x_ref) ** 2 + (y_rep - y_ref) ** 2)
    rel = (depth_rep - depth_ref).abs() / depth_ref
    mask = (dist < geo_pixel_thres) & (rel < geo_depth_thres)
    return mask, torch.where(mask, depth_rep, torch.zeros_like(depth_rep))


def fuse_scan(views: Dict[int, Dict], pairs: List[Tuple[int, List[int]]], geo_pixel_thres: float, geo_depth_thres: float,
              geo_mask_thres: int, photo_thres: float, device: torch.device):
    """views[id] = {depth [H,W], confidence [H,W], intrinsics [3,3], extrinsics [4,4], image [H,W,3]} (numpy or torch).
    Returns (vertices [M,3] float32, colors [M,3] uint8, masks {ref: (photo, geo, final) bool arrays}); reference
    eval.py:193-281."""
    dt = torch.float64

    def dev(a):
        return torch.as_tensor(a).to(device=device, dtype=dt)

    cache = {vid: {k: dev(v[k]) for k in ("depth", "confidence", "intrinsics", "extrinsics")} for vid, v in views.items()}
    verts, cols, masks = [], [], {}
    for ref, srcs in pairs:
        r = cache[ref]
        photo = r["confidence"] > photo_thres
        geo_sum = torch.zeros_like(r["depth"], dtype=torch.int32)
        acc = r["depth"].clone()
        for s in srcs:
            c = cache[s]
            m, rep = check_geometric_consistency(r["depth"], r["intrinsics"], r["extrinsics"], c["depth"],
                                                 c["intrinsics"], c["extrinsics"], geo_pixel_thres, geo_depth_thres)
            geo_sum += m.to(torch.int32)
            acc = acc + rep
        averaged = acc / (geo_sum + 1)
        geo = geo_sum >= geo_mask_thres
        final = photo & geo
        masks[ref] = (photo.cpu().numpy(), geo.cpu().numpy(), final.cpu().numpy())
        ys, xs = torch.nonzero(final, as_tuple=True)
        d = averaged[ys, xs]
        pix = torch.stack((xs.to(dt), ys.to(dt), torch.ones_like(d)))
        xyz_ref = torch.linalg.inv(r["intrinsics"]) @ (pix * d)
        xyz_world = (torch.linalg.inv(r["extrinsics"]) @ torch.cat((xyz_ref, torch.ones_like(d)[None])))[:3]
        verts.append(xyz_world.t().float().cpu().numpy())
        img = np.asarray(views[ref]["image"])
        cols.append((img[final.cpu().numpy()] * 255).astype(np.uint8))
    return np.concatenate(verts, 0), np.concatenate(cols, 0), masks


def write_ply(filename: str, vertices: np.ndarray, colors: np.ndarray) -> None:
    """Binary little-endian PLY with x,y,z float32 + red,green,blue uint8 per vertex (what plyfile writes at reference
    eval.py:283-297)."""
    rec = np.empty(len(vertices), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"),
                                         ("blue", "u1")])
    rec["x"], rec["y"], rec["z"] = vertices[:, 0], vertices[:, 1], vertices[:, 2]
    rec["red"], rec["green"], rec["blue"] = colors[:, 0], colors[:, 1], colors[:, 2]
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
              "property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % len(rec))
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    with open(filename, "wb") as f:
        f.write(header.encode("ascii"))
        rec.tofile(f)
