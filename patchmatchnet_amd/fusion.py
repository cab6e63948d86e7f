"""Photometric + geometric consistency filtering and point-cloud fusion of the predicted maps (reference eval.py:86-297) on the
device: one HIP kernel launch per reference view (pmn_fuse_view, csrc/fusion.hip) over the per-scan [V,2,H,W] map buffer -- the
buffer the per-scan all-gather leaves on every rank, so nothing is re-read from disk and every rank can fuse its own reference
views.  The reference does this per (reference, source) pair in single-threaded numpy + cv2.remap.

SURVEY.md 8(f) row 2.  No CPU / eager fallback: the arithmetic lives in the kernel (its CPU restatement, oracle/fusion_oracle.py, is
test infrastructure).  Only the 3x3 / 4x4 camera algebra is done here on the host, with numpy's own float32 inverse / matmul so
that the kernel receives exactly the matrices the reference computes (eval.py:114-139).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from ._lib import PmnError


def camera_block(K_ref: np.ndarray, E_ref: np.ndarray, srcs: Sequence[Tuple[np.ndarray, np.ndarray]]) -> np.ndarray:
    """The ``mats`` argument of pmn_fuse_view (layout: include/pmn_hip.h): 48 floats for the reference view + 64 per source."""
    K_ref = np.asarray(K_ref, np.float32)
    E_ref = np.asarray(E_ref, np.float32)
    out = np.zeros(48 + 64 * len(srcs), np.float32)
    Kri, Eri = np.linalg.inv(K_ref), np.linalg.inv(E_ref)
    out[0:9] = Kri.reshape(-1)
    out[9:18] = K_ref.reshape(-1)
    out[18:34] = Eri.reshape(-1)
    for i, (K_src, E_src) in enumerate(srcs):
        K_src = np.asarray(K_src, np.float32)
        E_src = np.asarray(E_src, np.float32)
        b = 48 + 64 * i
        out[b:b + 16] = np.matmul(E_src, Eri).reshape(-1)
        out[b + 16:b + 25] = K_src.reshape(-1)
        out[b + 25:b + 34] = np.linalg.inv(K_src).reshape(-1)
        out[b + 34:b + 50] = np.matmul(E_ref, np.linalg.inv(E_src)).reshape(-1)
    return out


def fuse_views(maps: torch.Tensor, slot_of: Dict[int, int], cams: Dict[int, Dict], images: Dict[int, np.ndarray],
               pairs: List[Tuple[int, List[int]]], geo_pixel_thres: float, geo_depth_thres: float, geo_mask_thres: int,
               photo_thres: float, sizes: Optional[Dict[int, Tuple[int, int]]] = None, pool=None, on_view=None,
               as_records: bool = False):
    """Fuses the reference views listed in ``pairs`` (this rank's share of a scan).

    maps [V,2,H,W] device float32 (slot_of[view id] -> slot) -- or, for a scan whose views differ in size, [V,F] flat slots with
    ``sizes[view id] = (h, w)`` (every view's depth then confidence packed at the start of its slot; reference eval.py:203-237
    reads every view's maps at their own size); cams[id] = {"intrinsics" [3,3], "extrinsics" [4,4]} (numpy
    float32, intrinsics already scaled to the map size); images[ref id] = the reference view's image, [H,W,3] float in [0,1] as read_image returns it or the
    decoded uint8 bytes (which are the colours), or a Future of either.
    Returns (vertices [M,3] float32, colors [M,3] uint8, masks {ref: (photo, geo, final) bool [H,W]}) in ``pairs`` order, points
    of a view in row-major pixel order -- the reference's order (eval.py:270-281).  ``on_view(ref, (photo, geo, final))`` is called
    as soon as a view's masks are on the host, on the thread that finished the view (eval.py writes the mask PNGs there, while
    later views are still being fused).  ``as_records``: returns ([PLY vertex records of every view], None, masks) instead -- the
    records are packed on the finishing threads and ``write_ply_records`` streams them to the file, so a scan's points (1.4 GB for
    49 fully consistent 1600x1200 views) are never concatenated or copied again on the launch thread."""
    if not maps.is_cuda:
        raise PmnError("fusion runs on a ROCm GPU only (pmn_fuse_view); there is no CPU fallback")
    slot_sizes = None
    if sizes is not None:
        slot_sizes = [(1, 1)] * maps.shape[0]
        for vid, sl in slot_of.items():
            slot_sizes[sl] = tuple(sizes[vid])

    def finish(ref, m, xyz):
        """Host half of one reference view: masks and the kept points leave the device, colours are picked from the reference image
        (eval.py:270-281).  Runs on ``pool`` when one is given: the copies and numpy's boolean indexing release the GIL, and at
        50-60 ms per 1600x1200 view this half, not the kernel, is what a scan's fusion takes."""
        with torch.cuda.device(m.device):  # a pool thread starts on device 0, whatever the rank's device is
            final = m[2].bool()
            v = xyz[final].cpu().numpy()
            mk = m.cpu().numpy()
        mk = mk.view(bool) if mk.dtype == np.uint8 else mk.astype(bool)  # the kernel writes 0 / 1 bytes
        img = images[ref]
        if hasattr(img, "result"):  # a concurrent.futures.Future: eval.py decodes the reference images on a thread pool
            img = img.result()
        img = np.asarray(img)
        if img.dtype == np.uint8:
            # the decoded bytes ARE the colours: the reference's float32 k / 255.0 (datasets/data_io.py:45) then (color * 255)
            # .astype(uint8) (eval.py:275) returns k for every byte value (tests/test_io_and_dist.py) -- no float image needed
            c = img[mk[2]]
        else:
            c = (img[mk[2]] * 255).astype(np.uint8)
        if on_view is not None:
            on_view(ref, (mk[0], mk[1], mk[2]))
        if as_records:
            return ply_records(v, c), None, (mk[0], mk[1], mk[2])
        return v, c, (mk[0], mk[1], mk[2])

    pending = []
    for ref, srcs in pairs:
        block = camera_block(cams[ref]["intrinsics"], cams[ref]["extrinsics"],
                             [(cams[s]["intrinsics"], cams[s]["extrinsics"]) for s in srcs])
        mats = torch.from_numpy(block).to(maps.device)
        m, xyz, _, _ = ops.fuse_view(maps, slot_of[ref], [slot_of[s] for s in srcs], mats, geo_pixel_thres, geo_depth_thres,
                                     geo_mask_thres, photo_thres, sizes=slot_sizes)
        pending.append((ref, pool.submit(finish, ref, m, xyz) if pool is not None else finish(ref, m, xyz)))
    verts, cols, masks = [], [], {}
    for ref, res in pending:
        v, c, mk = res.result() if hasattr(res, "result") else res
        verts.append(v)
        cols.append(c)
        masks[ref] = mk
    if as_records:
        return verts, None, masks
    if not verts:
        return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8), masks
    return np.concatenate(verts, 0), np.concatenate(cols, 0), masks


def fuse_scan(views: Dict[int, Dict], pairs: List[Tuple[int, List[int]]], geo_pixel_thres: float, geo_depth_thres: float,
              geo_mask_thres: int, photo_thres: float, device: torch.device):
    """Whole-scan convenience wrapper: views[id] = {depth [H,W], confidence [H,W], intrinsics, extrinsics, image [H,W,3]}
    (numpy or torch) -> (vertices, colors, masks) as ``fuse_views``; reference eval.py:193-281."""
    ids = sorted(views)
    sizes = {v: tuple(np.shape(views[v]["depth"])) for v in ids}
    flat = max(2 * h * w for h, w in sizes.values())
    maps = torch.zeros((len(ids), flat), dtype=torch.float32)
    for i, v in enumerate(ids):  # depth then confidence, packed at the start of the view's slot
        h, w = sizes[v]
        maps[i, :h * w] = torch.as_tensor(views[v]["depth"]).float().reshape(-1)
        maps[i, h * w:2 * h * w] = torch.as_tensor(views[v]["confidence"]).float().reshape(-1)
    maps = maps.to(device)
    cams = {v: {"intrinsics": np.asarray(views[v]["intrinsics"], np.float32),
                "extrinsics": np.asarray(views[v]["extrinsics"], np.float32)} for v in ids}
    images = {r: views[r]["image"] for r, _ in pairs}
    return fuse_views(maps, {v: i for i, v in enumerate(ids)}, cams, images, pairs, geo_pixel_thres, geo_depth_thres,
                      geo_mask_thres, photo_thres, sizes=sizes)


PLY_VERTEX = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])


def ply_records(vertices: np.ndarray, colors: np.ndarray) -> np.ndarray:
    """The 15-byte vertex records of the PLY body (two strided byte copies: positions, colours)."""
    n = len(vertices)
    rec = np.empty(n, dtype=PLY_VERTEX)
    if n:
        raw = rec.view(np.uint8).reshape(n, 15)
        raw[:, :12] = np.ascontiguousarray(vertices, "<f4").view(np.uint8).reshape(n, 12)
        raw[:, 12:] = np.asarray(colors, np.uint8)
    return rec


def ply_header(n: int) -> bytes:
    return ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
            "property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % n).encode("ascii")


def write_ply(filename: str, vertices: np.ndarray, colors: np.ndarray) -> None:
    """Binary little-endian PLY with x,y,z float32 + red,green,blue uint8 per vertex (what plyfile writes at reference
    eval.py:283-297)."""
    rec = ply_records(vertices, colors)
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    with open(filename, "wb") as f:
        f.write(ply_header(len(rec)))
        rec.tofile(f)


def write_ply_records(filename: str, chunks, header: bool = True) -> int:
    """The same file as ``write_ply`` from per-view record arrays (``fuse_views(as_records=True)``), streamed chunk by chunk;
    ``header=False`` writes the bare records (eval.py's per-rank parts).  Returns the number of vertices."""
    total = sum(len(c) for c in chunks)
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    with open(filename, "wb") as f:
        if header:
            f.write(ply_header(total))
        for c in chunks:
            c.tofile(f)
    return total


# ---- round 5: the scan's PLY body packed on the device -----------------------------------------------------------------------

def fuse_views_packed(maps: torch.Tensor, slot_of: Dict[int, int], cams: Dict[int, Dict], images: Dict[int, torch.Tensor],
                      pairs: List[Tuple[int, List[int]]], geo_pixel_thres: float, geo_depth_thres: float, geo_mask_thres: int,
                      photo_thres: float, packer: "ops.PointPacker", sizes: Optional[Dict[int, Tuple[int, int]]] = None):
    """The device half of ``fuse_views`` and nothing else: per reference view of ``pairs`` one pmn_fuse_view launch and one
    pmn_pack_points (three small launches) on the CURRENT stream -- the kept points become PLY vertex records appended to
    ``packer.records`` in pair-file / row-major order (reference eval.py:270-297), nothing synchronises, nothing leaves the device.
    ``images[ref]``: the reference view's image ON THE DEVICE, [H,W,3] uint8 (the decoded bytes) or float32 in [0,1].
    A generator: yields (ref, masks) per view -- masks = the [3,H,W] uint8 device tensor (photo, geo, final) -- so that the caller
    can queue its download behind the launches.  After the last view ``packer.counts()`` holds every view's number of points."""
    if not maps.is_cuda:
        raise PmnError("fusion runs on a ROCm GPU only (pmn_fuse_view); there is no CPU fallback")
    slot_sizes = None
    if sizes is not None:
        slot_sizes = [(1, 1)] * maps.shape[0]
        for vid, sl in slot_of.items():
            slot_sizes[sl] = tuple(sizes[vid])
    for ref, srcs in pairs:
        block = camera_block(cams[ref]["intrinsics"], cams[ref]["extrinsics"],
                             [(cams[s]["intrinsics"], cams[s]["extrinsics"]) for s in srcs])
        mats = torch.from_numpy(block).to(maps.device)
        m, xyz, _, _ = ops.fuse_view(maps, slot_of[ref], [slot_of[s] for s in srcs], mats, geo_pixel_thres, geo_depth_thres,
                                     geo_mask_thres, photo_thres, sizes=slot_sizes)
        packer.append(m[2], xyz, images[ref])
        yield ref, m


class PinnedRing:
    """A few pinned host buffers of one size handed out round-robin; ``acquire`` blocks until the buffer's previous user called
    ``release`` (back-pressure for a download -> write pipeline whose consumers are pool threads)."""

    def __init__(self, nbytes: int, count: int) -> None:
        import queue
        self.nbytes = int(nbytes)
        self.free: "queue.Queue[torch.Tensor]" = queue.Queue()
        for _ in range(count):
            self.free.put(torch.empty((self.nbytes,), dtype=torch.uint8).pin_memory())

    def acquire(self) -> torch.Tensor:
        return self.free.get()

    def release(self, buf: torch.Tensor) -> None:
        self.free.put(buf)


def download_to_file(src: torch.Tensor, nbytes: int, fd: int, file_offset: int, ring: PinnedRing, pool, stream) -> list:
    """``src[:nbytes]`` (device uint8) -> the open file ``fd`` at ``file_offset``: chunks of the ring's buffer size are copied into
    pinned memory on ``stream`` and written by ``pool`` threads with pwrite at their own offsets (the copy of chunk k+1 overlaps the
    write of chunk k; several writes run at once; os.pwrite releases the GIL).  Returns the write futures."""
    futures, done = [], 0
    while done < nbytes:
        n = min(ring.nbytes, nbytes - done)
        buf = ring.acquire()
        with torch.cuda.stream(stream):
            buf[:n].copy_(src[done:done + n], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)

        def write(buf=buf, n=n, at=file_offset + done, ev=ev):
            try:
                ev.synchronize()
                mv, put = memoryview(buf.numpy())[:n], 0
                while put < n:
                    put += os.pwrite(fd, mv[put:], at + put)
            finally:
                ring.release(buf)

        futures.append(pool.submit(write))
        done += n
    return futures
