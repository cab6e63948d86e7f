"""pmn_fuse_view (csrc/fusion.hip) against the CPU restatement of the reference's consistency filtering + fusion
(oracle/fusion_oracle.py <- reference eval.py:86-190, :207-281) on a synthetic multi-view scene, through the C ABI."""
import numpy as np
import pytest
import torch

import synth
from oracle import fusion_oracle as FO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scene(V, H, W, seed):
    rng = np.random.default_rng(seed)
    intr, extr = synth.synthetic_cameras(V, H * 2, W * 2)
    intr = intr[0].copy()
    intr[:, :2] *= 0.5  # cameras of the half-resolution maps
    extr = extr[0]
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    views = {}
    for v in range(V):
        # a slanted plane seen from every camera would need ray casting; consistency only needs plausible, mutually overlapping
        # maps: a smooth surface around z = 650 plus noise, holes and a few outliers
        d = 650.0 + 40.0 * np.sin(xx / 17.0 + v) * np.cos(yy / 13.0) + rng.standard_normal((H, W)) * (0.5 + 1.5 * (v % 2))
        d = d.astype(np.float32)
        d[rng.random((H, W)) < 0.01] = 0.0
        d[rng.random((H, W)) < 0.01] *= 1.3
        c = rng.random((H, W)).astype(np.float32)
        views[v * 3 + 1] = dict(depth=d, confidence=c, intrinsics=intr[v].astype(np.float32), extrinsics=extr[v].astype(np.float32),
                                image=rng.random((H, W, 3)).astype(np.float32))
    return views


@pytest.mark.parametrize("H,W,V", [(60, 80, 4), (37, 53, 3), (300, 400, 6)])
def test_fuse_view_matches_oracle(H, W, V):
    assert torch.cuda.is_available(), "GPU tests selected but no ROCm device is visible"
    from patchmatchnet_amd import fusion, ops
    views = _scene(V, H, W, seed=H + V)
    ids = sorted(views)
    maps = torch.stack([torch.stack((torch.from_numpy(views[v]["depth"]), torch.from_numpy(views[v]["confidence"]))) for v in ids]).to(DEV)
    slot_of = {v: i for i, v in enumerate(ids)}
    thr_px, thr_d, thr_n, thr_p = 1.0, 0.01, 2 if V > 3 else 1, 0.3
    for ref in ids[:2]:
        srcs = [s for s in ids if s != ref]
        block = fusion.camera_block(views[ref]["intrinsics"], views[ref]["extrinsics"],
                                    [(views[s]["intrinsics"], views[s]["extrinsics"]) for s in srcs])
        masks, xyz, davg, gsum = ops.fuse_view(maps, slot_of[ref], [slot_of[s] for s in srcs], torch.from_numpy(block).to(DEV),
                                               thr_px, thr_d, thr_n, thr_p, want_depth_avg=True, want_geo_sum=True)
        torch.cuda.synchronize()
        want = FO.fuse_view(views[ref], [views[s] for s in srcs], thr_px, thr_d, thr_n, thr_p)
        masks, gsum, davg, xyz = masks.cpu().numpy().astype(bool), gsum.cpu().numpy(), davg.cpu().numpy(), xyz.cpu().numpy()
        np.testing.assert_array_equal(masks[0], want["photo"])
        # the per-source masks may flip only where a criterion sits on its threshold (float64 products summed in another order)
        bad = gsum != want["geo_sum"]
        assert float(bad.mean()) < 2e-3, float(bad.mean())
        assert int(np.abs(gsum - want["geo_sum"]).max()) <= 1
        same = ~bad
        assert want["geo_sum"].max() >= thr_n and same.mean() > 0.99
        np.testing.assert_array_equal(masks[1][same], want["geo"][same])
        np.testing.assert_array_equal(masks[2][same], want["final"][same])
        ok = same & np.isfinite(want["depth_avg"]) & (want["depth_avg"] != 0)
        rel = np.abs(davg[ok] - want["depth_avg"][ok]) / np.abs(want["depth_avg"][ok])
        assert rel.max() < 1e-6, float(rel.max())
        # world points of the pixels both sides keep
        both = masks[2] & want["final"] & same
        idx = np.cumsum(want["final"].reshape(-1)) - 1  # position of a final pixel in the oracle's (row-major) vertex list
        got_pts = xyz[both]
        want_pts = want["vertices"][idx[both.reshape(-1)]]
        assert got_pts.shape[0] > 100
        assert np.abs(got_pts - want_pts).max() / np.abs(want_pts).max() < 1e-6


def test_fuse_scan_product_wrapper_and_ply(tmp_path):
    assert torch.cuda.is_available(), "GPU tests selected but no ROCm device is visible"
    from patchmatchnet_amd import fusion
    views = _scene(4, 48, 64, seed=1)
    ids = sorted(views)
    pairs = [(r, [s for s in ids if s != r][:2]) for r in ids]
    v, c, masks = fusion.fuse_scan(views, pairs, 1.0, 0.01, 1, 0.3, torch.device(DEV))
    vo, co, mo = FO.fuse_scan(views, pairs, 1.0, 0.01, 1, 0.3)
    assert abs(len(v) - len(vo)) <= 0.002 * len(vo) + 2 and v.dtype == np.float32 and c.dtype == np.uint8 and len(c) == len(v)
    agree = np.mean([np.mean(masks[r][2] == mo[r][2]) for r in ids])
    assert agree > 0.998
    fusion.write_ply(str(tmp_path / "f.ply"), v, c)
    assert (tmp_path / "f.ply").stat().st_size == len(fusion.ply_header(len(v))) + 15 * len(v)


def _consistent_scene(V, H, W, sizes=None, seed=0):
    """Geometrically CONSISTENT maps: every view's ground-truth depth of the analytic surface of synth.render_scene in its own
    camera (+ 0.02 mm noise), so that the >= geo_mask_thres branch of the reference's filter (eval.py:248-252) holds on nearly every
    pixel that the source views see -- by construction, not by luck of a noise level.  ``sizes[v]`` = (h, w) re-renders view v at
    its own size with its intrinsics scaled (eval.py:214-229: per-view image size after --image_max_dim)."""
    rng = np.random.default_rng(seed)
    views = {}
    for v in range(V):
        h, w = (H, W) if sizes is None else sizes[v]
        imgs, intr, extr, depths = synth.render_scene(V, h, w, seed=seed, all_depths=True)
        d = depths[v].numpy() + rng.standard_normal((h, w)).astype(np.float32) * 0.02
        c = (0.4 + 0.6 * rng.random((h, w))).astype(np.float32)
        views[v * 2 + 3] = dict(depth=d.astype(np.float32), confidence=c, intrinsics=intr[0, v].astype(np.float32),
                                extrinsics=extr[0, v].astype(np.float32), image=imgs[v][0].permute(1, 2, 0).numpy())
    return views


@pytest.mark.parametrize("mixed", [False, True])
def test_fuse_scan_on_a_consistent_scene_incl_mixed_view_sizes(mixed):
    """pmn_fuse_view through fusion.fuse_scan vs the fusion oracle on a scene whose maps ARE consistent; ``mixed``: every view at
    its own size (reference eval.py:203-237 reads each view's maps at their own size -- round 2 raised on such scans)."""
    assert torch.cuda.is_available(), "GPU tests selected but no ROCm device is visible"
    from patchmatchnet_amd import fusion
    V, H, W = 5, 120, 160
    sizes = [(120, 160), (96, 128), (120, 160), (144, 192), (90, 120)] if mixed else None
    views = _consistent_scene(V, H, W, sizes)
    ids = sorted(views)
    pairs = [(r, [s for s in ids if s != r]) for r in ids]
    thr = (1.0, 0.01, 3, 0.5)
    v, c, masks = fusion.fuse_scan(views, pairs, *thr, torch.device(DEV))
    vo, co, mo = FO.fuse_scan(views, pairs, *thr)
    for r in ids:
        assert masks[r][2].shape == views[r]["depth"].shape
        np.testing.assert_array_equal(masks[r][0], mo[r][0])  # photometric mask: a plain comparison
        assert float(np.mean(masks[r][1] != mo[r][1])) < 2e-3  # geometric mask: equal off threshold ties
        assert mo[r][1].mean() > 0.5, (r, mo[r][1].mean())    # ... and it is genuinely ON for most of the view
    assert abs(len(v) - len(vo)) <= 0.002 * len(vo) + 2
    same = all(np.array_equal(masks[r][2], mo[r][2]) for r in ids)
    if same:
        assert np.abs(v - vo).max() / np.abs(vo).max() < 1e-6
        np.testing.assert_array_equal(c, co)


@pytest.mark.parametrize("mixed,float_images", [(False, False), (False, True), (True, True)])
def test_packed_records_are_the_host_paths_bytes(mixed, float_images, tmp_path):
    """pmn_pack_points (round 5): the scan's PLY body packed on the device view after view == the per-view host path of rounds 3-4
    (boolean index on the device, numpy's boolean index of the image, fusion.ply_records), BYTE FOR BYTE -- reference
    eval.py:270-297: valid pixels row-major, x y z float32 + (color * 255).astype(uint8).  uint8 images = the decoded bytes
    (eval.py's threaded decode path), float images = a resized image; ``mixed``: every view at its own size.  Also: the download of
    the record buffer through pinned chunks into a file at an offset, and the overflow report."""
    assert torch.cuda.is_available(), "GPU tests selected but no ROCm device is visible"
    import os
    from concurrent.futures import ThreadPoolExecutor
    from patchmatchnet_amd import fusion, ops
    V, H, W = 5, 120, 160
    sizes = [(120, 160), (96, 128), (120, 160), (144, 192), (90, 120)] if mixed else None
    views = _consistent_scene(V, H, W, sizes, seed=3)
    ids = sorted(views)
    for v in ids:  # images as eval.py holds them
        im = views[v]["image"]
        views[v]["image"] = np.ascontiguousarray(im, np.float32) if float_images else np.ascontiguousarray((im * 255).astype(np.uint8))
    pairs = [(r, [s for s in ids if s != r]) for r in ids]
    thr = (1.0, 0.01, 3, 0.5)
    vsizes = {v: tuple(views[v]["depth"].shape) for v in ids}
    flat = max(2 * h * w for h, w in vsizes.values())
    maps = torch.zeros((len(ids), flat), dtype=torch.float32)
    for i, v in enumerate(ids):
        h, w = vsizes[v]
        maps[i, :h * w] = torch.from_numpy(views[v]["depth"]).reshape(-1)
        maps[i, h * w:2 * h * w] = torch.from_numpy(views[v]["confidence"]).reshape(-1)
    maps = maps.to(DEV)
    slot_of = {v: i for i, v in enumerate(ids)}
    cams = {v: {"intrinsics": views[v]["intrinsics"], "extrinsics": views[v]["extrinsics"]} for v in ids}
    # host path (rounds 3-4)
    recs, _, masks = fusion.fuse_views(maps, slot_of, cams, {v: views[v]["image"] for v in ids}, pairs, *thr, sizes=vsizes, as_records=True)
    want = b"".join(r.tobytes() for r in recs)
    assert len(want) > 15 * 1000
    # device path
    packer = ops.PointPacker(sum(h * w for h, w in vsizes.values()), torch.device(DEV), max_views=8)
    dev_images = {v: torch.from_numpy(views[v]["image"]).to(DEV) for v in ids}
    got_masks = {ref: m.cpu().numpy().astype(bool) for ref, m in fusion.fuse_views_packed(maps, slot_of, cams, dev_images, pairs, *thr,
                                                                                          packer, sizes=vsizes)}
    counts = packer.counts()
    assert counts == [len(r) for r in recs]
    got = packer.records[:15 * sum(counts)].cpu().numpy().tobytes()
    assert got == want
    for r in ids:
        for k in range(3):
            np.testing.assert_array_equal(got_masks[r][k], masks[r][k])
    # the body through pinned chunks into a file behind a header (chunk size far below the body: several chunks, two writers)
    path = str(tmp_path / "body.bin")
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    ring = fusion.PinnedRing(4096 * 15, 3)
    with ThreadPoolExecutor(2) as pool:
        os.pwrite(fd, b"HEADER", 0)
        for f in fusion.download_to_file(packer.records, len(want), fd, 6, ring, pool, torch.cuda.current_stream()):
            f.result()
    os.close(fd)
    assert open(path, "rb").read() == b"HEADER" + want
    # a second scan re-uses the packer from record 0; a view that does not fit is reported, nothing is written past the buffer
    packer.reset()
    small = ops.PointPacker(100, torch.device(DEV), max_views=8)
    for _ in fusion.fuse_views_packed(maps, slot_of, cams, dev_images, pairs[:1], *thr, small, sizes=vsizes):
        pass
    with pytest.raises(Exception, match="too small"):
        small.counts()
