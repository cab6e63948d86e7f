"""Sample source for eval.py: the generic MVS folder layout of the reference (datasets/mvs.py:8-111), inference only.

    <data_path>/<scan>/images/[light/]00000000.jpg, <scan>/cams/00000000_cam.txt, <scan>/pair.txt

One sample = one reference view + its first ``num_views`` source views.  ``shard(rank, world)`` hands every rank of a
one-process-per-GPU job its contiguous block of every scan's reference views; no state is shared between samples.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import numpy as np
from torch.utils.data import Dataset

from .data_io import image_shape, read_cam_file, read_image, read_image_u8, read_pair_file


class MVSDataset(Dataset):
    def __init__(self, data_path: str, num_views: int = 10, max_dim: int = -1, scan_list: str = "",
                 num_light_idx: int = -1, cam_folder: str = "cams", pair_path: str = "pair.txt",
                 image_folder: str = "images", image_extension: str = ".jpg") -> None:
        super().__init__()
        self.data_path, self.num_views, self.max_dim = data_path, num_views, max_dim
        self.cam_folder, self.image_folder, self.image_extension = cam_folder, image_folder, image_extension
        self.load_images = True  # False: samples carry cameras and image SHAPES only (eval.py's encode-once path)
        self.uint8_images = False  # True: images that need no down-scaling come as uint8 [3,H,W]; the consumer divides by 255
        # camera files and image headers are read once per (scan, view), not once per sample the view appears in (a view is the
        # source of ~num_views other samples; with images served from the feature cache the 12 small file reads of a sample were
        # the largest item of eval.py's per-sample host time)
        self._cam_cache: Dict[str, Tuple[np.ndarray, np.ndarray, np.ndarray]] = {}
        self._shape_cache: Dict[str, Tuple[int, int, int, int]] = {}
        if os.path.isfile(scan_list):
            with open(scan_list) as f:
                scans = [ln.rstrip() for ln in f.readlines()]
        else:
            scans = [""]
        self.scans = list(scans)
        lights = [str(i) for i in range(num_light_idx)] if num_light_idx > 0 else [""]
        self.metas: List[Tuple[str, str, int, List[int]]] = []
        for scan in scans:
            pairs = read_pair_file(os.path.join(data_path, scan, pair_path))
            for light in lights:
                self.metas += [(scan, light, ref, src) for ref, src in pairs]

    def shard(self, rank: int, world_size: int) -> "MVSDataset":
        """This rank's reference views: within EVERY (scan, light) group the rank's contiguous block (dist.block_range), so that
        ownership is the same function of the position inside the scan that dist.shard_views / the per-scan all-gather use --
        whatever the number of scans and views (every rank keeps the full model replica)."""
        from .dist import block_range
        if not 0 <= rank < world_size:
            raise ValueError("rank out of range")
        groups: Dict[Tuple[str, str], List[Tuple[str, str, int, List[int]]]] = {}
        for meta in self.metas:
            groups.setdefault((meta[0], meta[1]), []).append(meta)
        kept: List[Tuple[str, str, int, List[int]]] = []
        for metas in groups.values():
            a, b = block_range(len(metas), rank, world_size)
            kept += metas[a:b]
        self.metas = kept
        return self

    def __len__(self) -> int:
        return len(self.metas)

    def scan_index(self, scan: str) -> int:
        """Position of ``scan`` in the scan list (independent of sharding)."""
        return self.scans.index(scan)

    def image_path(self, scan: str, light: str, vid: int) -> str:
        return os.path.join(self.data_path, scan, self.image_folder, light, "{:0>8}{}".format(vid, self.image_extension))

    def groups(self) -> "Dict[Tuple[str, str], List[int]]":
        """Sample indices of this (sharded) dataset per (scan, light), in order."""
        out: Dict[Tuple[str, str], List[int]] = {}
        for i, (scan, light, _, _) in enumerate(self.metas):
            out.setdefault((scan, light), []).append(i)
        return out

    def views_of(self, indices: List[int]) -> List[int]:
        """Every view id the given samples read (reference + the source views actually used), sorted."""
        ids = set()
        for i in indices:
            _, _, ref, src = self.metas[i]
            ids.update([ref] + src[:min(len(src), self.num_views)])
        return sorted(ids)


    def __getitem__(self, idx: int) -> Dict:
        scan, light, ref_view, src_views = self.metas[idx]
        view_ids = [ref_view] + src_views[:min(len(src_views), self.num_views)]
        images, intrinsics, extrinsics = [], [], []
        depth_min = depth_max = -1.0
        for i, vid in enumerate(view_ids):
            path = self.image_path(scan, light, vid)
            if self.load_images:
                raw = read_image_u8(path, self.max_dim) if self.uint8_images else None
                if raw is not None:
                    img, h0, w0 = raw, raw.shape[0], raw.shape[1]
                else:
                    img, h0, w0 = read_image(path, self.max_dim)
                images.append(np.ascontiguousarray(img.transpose(2, 0, 1)))
                hi, wi = img.shape[0], img.shape[1]
            else:
                if path not in self._shape_cache:
                    self._shape_cache[path] = image_shape(path, self.max_dim)
                hi, wi, h0, w0 = self._shape_cache[path]
                images.append(np.asarray([hi, wi], np.int64))
            cam_path = os.path.join(self.data_path, scan, self.cam_folder, "{:0>8}_cam.txt".format(vid))
            if cam_path not in self._cam_cache:
                self._cam_cache[cam_path] = read_cam_file(cam_path)
            K, E, depth_params = (a.copy() for a in self._cam_cache[cam_path])  # K is rescaled in place below
            K[0] *= wi / w0
            K[1] *= hi / h0
            intrinsics.append(K)
            extrinsics.append(E)
            if i == 0:
                depth_min, depth_max = depth_params[0], depth_params[1]
        return {"images": images, "intrinsics": np.stack(intrinsics), "extrinsics": np.stack(extrinsics),
                "depth_min": depth_min, "depth_max": depth_max, "ref_view": view_ids[0],
                "view_ids": np.asarray(view_ids, np.int64), "scan": scan, "light": light,
                "filename": os.path.join(scan, "{}", "{:0>8}".format(view_ids[0]) + "{}")}


class MVSViewDataset(Dataset):
    """The distinct images of one (scan, light) -- each decoded ONCE (eval.py's encode-once path; the reference decodes and
    encodes an image once per sample it appears in)."""

    def __init__(self, parent: MVSDataset, scan: str, light: str, view_ids: List[int]) -> None:
        super().__init__()
        self.parent, self.scan, self.light, self.view_ids = parent, scan, light, list(view_ids)

    def __len__(self) -> int:
        return len(self.view_ids)

    def __getitem__(self, idx: int) -> Dict:
        vid = self.view_ids[idx]
        path = self.parent.image_path(self.scan, self.light, vid)
        img = read_image_u8(path, self.parent.max_dim) if self.parent.uint8_images else None
        if img is None:
            img, _, _ = read_image(path, self.parent.max_dim)
        return {"image": np.ascontiguousarray(img.transpose(2, 0, 1)), "view": vid}
