"""CPU restatement of the reference's geometric / photometric consistency filtering and point fusion -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product (patchmatchnet_amd/fusion.py
-> pmn_fuse_view, csrc/fusion.hip) never does.

Restates /root/reference/eval.py:
    reproject_with_depth          :86-145     (dtype flow kept: float32 camera matrices, int64 pixel grids, float64 products,
                                               float32 casts of the map coordinates / re-projected depth and positions)
    check_geometric_consistency   :148-190
    filter_depth, one ref view    :207-281    (photo mask, mask sum, averaged depth, final mask, world points, colours)

The one step that lives in a dependency that is absent here is ``cv2.remap(depth_src, x_src, y_src, cv2.INTER_LINEAR)``
(eval.py:129; opencv-python is unpinned in the reference's requirements.txt and not installed in this image).  It is restated from
OpenCV 4.x's published algorithm (modules/imgproc/src/imgwarp.cpp, cv::remap with CV_32FC1 maps, RemapInvoker +
remapBilinear<Cast<float,float>, RemapNoVec, float>): the float map coordinates are converted to fixed point with INTER_BITS = 5
(sx = cvRound(x * 32), round half to even), the integer part selects the 2x2 block, the 5-bit fractions index a table of float
weights (1-fx)(1-fy), fx(1-fy), (1-fx)fy, fx*fy with fx = k/32, taps outside the image read the constant border value 0.
PARITY WITH cv2 ITSELF IS UNPINNED (no cv2 here to generate fixtures); everything else is pinned by running numpy, which is what
the reference runs.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS


def remap_linear_cv2(src: np.ndarray, map_x: np.ndarray, map_y: np.ndarray) -> np.ndarray:
    """cv2.remap(src, map_x, map_y, cv2.INTER_LINEAR) for a float32 single-channel ``src`` and float32 maps, BORDER_CONSTANT 0."""
    src = np.asarray(src, np.float32)
    if src.ndim == 3:
        src = src[:, :, 0]
    H, W = src.shape
    mx = np.asarray(map_x, np.float32) * np.float32(INTER_TAB_SIZE)
    my = np.asarray(map_y, np.float32) * np.float32(INTER_TAB_SIZE)
    ok = np.isfinite(mx) & np.isfinite(my) & (np.abs(mx) < 2.0 ** 30) & (np.abs(my) < 2.0 ** 30)
    sx = np.rint(np.where(ok, mx, 0)).astype(np.int64)  # cvRound: to nearest, ties to even
    sy = np.rint(np.where(ok, my, 0)).astype(np.int64)
    ix, iy = sx >> INTER_BITS, sy >> INTER_BITS
    fx = (sx & (INTER_TAB_SIZE - 1)).astype(np.float32) / np.float32(INTER_TAB_SIZE)
    fy = (sy & (INTER_TAB_SIZE - 1)).astype(np.float32) / np.float32(INTER_TAB_SIZE)
    one = np.float32(1.0)
    w00, w01 = (one - fy) * (one - fx), (one - fy) * fx
    w10, w11 = fy * (one - fx), fy * fx

    def tap(yy, xx):
        inside = ok & (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        v = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        return np.where(inside, v, np.float32(0.0)).astype(np.float32)

    out = tap(iy, ix) * w00
    out = out + tap(iy, ix + 1) * w01
    out = out + tap(iy + 1, ix) * w10
    out = out + tap(iy + 1, ix + 1) * w11
    return out.astype(np.float32)


def reproject_with_depth(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src):
    """eval.py:86-145 -> (depth_reprojected, x_reprojected, y_reprojected), float32 [H,W]."""
    width, height = depth_ref.shape[1], depth_ref.shape[0]
    x_ref, y_ref = np.meshgrid(np.arange(0, width), np.arange(0, height))
    x_ref, y_ref = x_ref.reshape([-1]), y_ref.reshape([-1])
    with np.errstate(divide="ignore", invalid="ignore"):
        xyz_ref = np.matmul(np.linalg.inv(intrinsics_ref), np.vstack((x_ref, y_ref, np.ones_like(x_ref))) * depth_ref.reshape([-1]))
        xyz_src = np.matmul(np.matmul(extrinsics_src, np.linalg.inv(extrinsics_ref)), np.vstack((xyz_ref, np.ones_like(x_ref))))[:3]
        k_xyz_src = np.matmul(intrinsics_src, xyz_src)
        xy_src = k_xyz_src[:2] / k_xyz_src[2:3]
        x_src = xy_src[0].reshape([height, width]).astype(np.float32)
        y_src = xy_src[1].reshape([height, width]).astype(np.float32)
        sampled_depth_src = remap_linear_cv2(depth_src, x_src, y_src)
        xyz_src = np.matmul(np.linalg.inv(intrinsics_src),
                            np.vstack((xy_src, np.ones_like(x_ref))) * sampled_depth_src.reshape([-1]))
        xyz_reprojected = np.matmul(np.matmul(extrinsics_ref, np.linalg.inv(extrinsics_src)),
                                    np.vstack((xyz_src, np.ones_like(x_ref))))[:3]
        depth_reprojected = xyz_reprojected[2].reshape([height, width]).astype(np.float32)
        k_xyz_reprojected = np.matmul(intrinsics_ref, xyz_reprojected)
        xy_reprojected = k_xyz_reprojected[:2] / k_xyz_reprojected[2:3]
    x_reprojected = xy_reprojected[0].reshape([height, width]).astype(np.float32)
    y_reprojected = xy_reprojected[1].reshape([height, width]).astype(np.float32)
    return depth_reprojected, x_reprojected, y_reprojected


def check_geometric_consistency(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src,
                                geo_pixel_thres: float, geo_depth_thres: float):
    """eval.py:148-190 -> (mask bool [H,W], depth_reprojected float32 [H,W] with inconsistent pixels zeroed)."""
    width, height = depth_ref.shape[1], depth_ref.shape[0]
    x_ref, y_ref = np.meshgrid(np.arange(0, width), np.arange(0, height))
    depth_reprojected, x2d, y2d = reproject_with_depth(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src,
                                                       extrinsics_src)
    with np.errstate(divide="ignore", invalid="ignore"):
        dist = np.sqrt((x2d - x_ref) ** 2 + (y2d - y_ref) ** 2)
        depth_diff = np.abs(depth_reprojected - depth_ref)
        relative_depth_diff = depth_diff / depth_ref
        mask = np.logical_and(dist < geo_pixel_thres, relative_depth_diff < np.float32(geo_depth_thres))
    depth_reprojected = depth_reprojected.copy()
    depth_reprojected[~mask] = 0
    return mask, depth_reprojected


def fuse_view(ref: Dict, srcs: Sequence[Dict], geo_pixel_thres: float, geo_depth_thres: float, geo_mask_thres: int,
              photo_thres: float):
    """One reference view of filter_depth (eval.py:207-281).  ``ref`` / ``srcs[i]`` = {depth [H,W] f32, confidence [H,W] f32
    (ref only), intrinsics [3,3] f32, extrinsics [4,4] f32, image [H,W,3] float in [0,1] (ref only, optional)}.
    Returns dict(photo, geo, final masks; geo_sum int32; depth_avg float64; vertices [M,3] float32; colors [M,3] uint8)."""
    ref_depth = np.asarray(ref["depth"], np.float32)
    photo = np.asarray(ref["confidence"], np.float32) > photo_thres
    geo_sum = np.zeros(ref_depth.shape, np.int32)
    reproj = []
    for s in srcs:
        m, d = check_geometric_consistency(ref_depth, ref["intrinsics"], ref["extrinsics"], np.asarray(s["depth"], np.float32),
                                           s["intrinsics"], s["extrinsics"], geo_pixel_thres, geo_depth_thres)
        geo_sum += m.astype(np.int32)
        reproj.append(d)
    depth_avg = (sum(reproj) + ref_depth) / (geo_sum + 1)
    geo = geo_sum >= geo_mask_thres
    final = np.logical_and(photo, geo)
    height, width = ref_depth.shape
    x, y = np.meshgrid(np.arange(0, width), np.arange(0, height))
    x, y, depth = x[final], y[final], depth_avg[final]
    xyz_ref = np.matmul(np.linalg.inv(ref["intrinsics"]), np.vstack((x, y, np.ones_like(x))) * depth)
    xyz_world = np.matmul(np.linalg.inv(ref["extrinsics"]), np.vstack((xyz_ref, np.ones_like(x))))[:3]
    out = dict(photo=photo, geo=geo, final=final, geo_sum=geo_sum, depth_avg=np.asarray(depth_avg, np.float64),
               vertices=xyz_world.transpose((1, 0)).astype(np.float32))
    if "image" in ref and ref["image"] is not None:
        out["colors"] = (np.asarray(ref["image"])[final] * 255).astype(np.uint8)
    return out


def fuse_scan(views: Dict[int, Dict], pairs: List[Tuple[int, List[int]]], geo_pixel_thres: float, geo_depth_thres: float,
              geo_mask_thres: int, photo_thres: float):
    """All reference views of a scan (eval.py:193-281): (vertices [M,3] f32, colors [M,3] u8, {ref: (photo, geo, final)})."""
    verts, cols, masks = [], [], {}
    for ref, srcs in pairs:
        r = fuse_view(views[ref], [views[s] for s in srcs], geo_pixel_thres, geo_depth_thres, geo_mask_thres, photo_thres)
        masks[ref] = (r["photo"], r["geo"], r["final"])
        verts.append(r["vertices"])
        cols.append(r["colors"])
    return np.concatenate(verts, 0), np.concatenate(cols, 0), masks
