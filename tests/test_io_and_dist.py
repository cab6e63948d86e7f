"""CPU tests of the callers either side of the hot path: on-disk formats, the sample source, rank sharding (one and several
scans), the per-scan all-gather (world_size 2 over gloo).  These check the code against layouts written down HERE; the same code
against the REFERENCE's own datasets/ modules and reference-written files is tests/test_reference_io.py.  The consistency-fusion arithmetic is in tests/test_fusion_oracle.py
(oracle, CPU) and tests/test_fusion_gpu.py (pmn_fuse_view against it)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import synth
from patchmatchnet_amd import data_io, fusion
from patchmatchnet_amd import _lib
from patchmatchnet_amd import dist as pdist
from patchmatchnet_amd.mvs import MVSDataset


def test_pfm_roundtrip_and_layout(tmp_path):
    a = np.random.default_rng(0).standard_normal((7, 5)).astype(np.float32)
    p = str(tmp_path / "a.pfm")
    data_io.save_map(p, a)
    raw = open(p, "rb").read()
    assert raw.startswith(b"Pf\n5 7\n-1.000000\n")  # header bytes of the reference writer (data_io.py:289-300)
    payload = np.frombuffer(raw[len(b"Pf\n5 7\n-1.000000\n"):], "<f4").reshape(7, 5)
    np.testing.assert_array_equal(payload, a[::-1])  # rows stored bottom-up
    b = data_io.read_map(p)
    assert b.shape == (7, 5, 1)
    np.testing.assert_array_equal(b[..., 0], a)
    c = np.random.default_rng(1).random((4, 6, 3)).astype(np.float32)
    data_io.save_pfm(str(tmp_path / "c.pfm"), c)
    np.testing.assert_array_equal(data_io.read_pfm(str(tmp_path / "c.pfm"))[0], c)
    with pytest.raises(Exception):
        data_io.save_pfm(str(tmp_path / "d.pfm"), a.astype(np.float64))
    # eval.py's writer threads hand over maps that were flipped on the device: same file, byte for byte
    data_io.save_map(str(tmp_path / "e.pfm"), np.ascontiguousarray(a[::-1]), rows_flipped=True)
    assert open(str(tmp_path / "e.pfm"), "rb").read() == raw


def test_colmap_bin_roundtrip_and_layout(tmp_path):
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    p = str(tmp_path / "a.bin")
    data_io.save_map(p, a)
    raw = open(p, "rb").read()
    assert raw.startswith(b"4&3&1&")
    # reference save_bin (data_io.py:209-217): transpose to (W,H), flatten in Fortran order => row-major x-fastest
    np.testing.assert_array_equal(np.frombuffer(raw[6:], "<f4"), np.transpose(a, (1, 0)).reshape(-1, order="F"))
    b = data_io.read_map(p)
    assert b.shape == (3, 4, 1)
    np.testing.assert_array_equal(b[..., 0], a)
    # a non-contiguous map takes the general path and writes the same bytes
    wide = np.zeros((3, 8), np.float32)
    wide[:, ::2] = a
    data_io.save_map(str(tmp_path / "v.bin"), wide[:, ::2])
    assert open(str(tmp_path / "v.bin"), "rb").read() == raw
    with pytest.raises(Exception):
        data_io.read_map(str(tmp_path / "x.png"))


def test_resize_matches_plain_bilinear():
    img = np.random.default_rng(2).random((40, 64, 3)).astype(np.float32)
    out, h0, w0 = data_io.scale_to_max_dim(img, 32)
    assert (h0, w0) == (40, 64) and out.shape == (20, 32, 3)
    ref = torch.nn.functional.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None], size=(20, 32), mode="bilinear",
                                          align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(out - ref).max() < 1e-6
    same, _, _ = data_io.scale_to_max_dim(img, 128)
    assert same is img
    same, _, _ = data_io.scale_to_max_dim(img, -1)
    assert same is img


def test_scan_reader_and_sharding(tmp_path):
    synth.write_scan(str(tmp_path), "scan1", n_views=5, H=64, W=96, n_src=3)
    with open(tmp_path / "list.txt", "w") as f:
        f.write("scan1\n")
    ds = MVSDataset(str(tmp_path), num_views=2, max_dim=-1, scan_list=str(tmp_path / "list.txt"))
    assert len(ds) == 5
    s = ds[1]
    assert len(s["images"]) == 3 and s["images"][0].shape == (3, 64, 96)
    assert s["intrinsics"].shape == (3, 3, 3) and s["extrinsics"].shape == (3, 4, 4)
    assert float(s["depth_min"]) == 425.0 and float(s["depth_max"]) == 935.0
    assert s["filename"].format("depth_est", ".pfm") == os.path.join("scan1", "depth_est", "00000001.pfm")
    intr, extr = synth.synthetic_cameras(5, 64, 96)
    np.testing.assert_allclose(s["intrinsics"][0], intr[0, 1], rtol=1e-6)
    np.testing.assert_allclose(s["extrinsics"][0], extr[0, 1], atol=1e-5)
    # down-scaling rescales the intrinsics rows (reference mvs.py:86-87)
    ds2 = MVSDataset(str(tmp_path), num_views=2, max_dim=48, scan_list=str(tmp_path / "list.txt"))
    s2 = ds2[0]
    assert s2["images"][0].shape == (3, 32, 48)
    np.testing.assert_allclose(s2["intrinsics"][0][0], intr[0, 0][0] * 0.5, rtol=1e-6)
    # rank sharding partitions the reference views
    seen = []
    for r in range(3):
        d = MVSDataset(str(tmp_path), num_views=2, scan_list=str(tmp_path / "list.txt")).shard(r, 3)
        seen += [m[2] for m in d.metas]
    assert sorted(seen) == [0, 1, 2, 3, 4]
    assert pdist.shard_views([0, 1, 2, 3, 4], 1, 3) == [2, 3]  # contiguous blocks: 2 + 2 + 1


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharding_agrees_with_the_gather_on_several_scans(tmp_path, world):
    """Ownership must be the same function of the position INSIDE a scan on both sides -- the sample source (MVSDataset.shard)
    and the per-scan all-gather / fusion (dist.shard_views) -- whatever the scan sizes.  (A global round-robin slice of the meta
    list shifts by the previous scans' sizes: with 2 scans x 5 views and 2 ranks, rank 0 would own views [1, 3] of the second
    scan while the gather expects [0, 2, 4], and fusion dies after all depth inference has run.)"""
    synth.write_scan(str(tmp_path), "scanA", n_views=5, H=32, W=48, n_src=2)
    synth.write_scan(str(tmp_path), "scanB", n_views=7, H=32, W=48, n_src=2)
    synth.write_scan(str(tmp_path), "scanC", n_views=3, H=32, W=48, n_src=2)
    with open(tmp_path / "list.txt", "w") as f:
        f.write("scanA\nscanB\nscanC\n")
    seen = {}
    for r in range(world):
        d = MVSDataset(str(tmp_path), num_views=2, scan_list=str(tmp_path / "list.txt")).shard(r, world)
        for scan, n in (("scanA", 5), ("scanB", 7), ("scanC", 3)):
            mine = [m[2] for m in d.metas if m[0] == scan]
            assert mine == pdist.shard_views(list(range(n)), r, world), (scan, r)
            seen.setdefault(scan, []).extend(mine)
            # ... and the groups stay contiguous per scan (the encode-once path walks them group by group)
        assert [k[0] for k in d.groups()] == [s for s in ("scanA", "scanB", "scanC") if any(m[0] == s for m in d.metas)]
    assert {k: sorted(v) for k, v in seen.items()} == {"scanA": list(range(5)), "scanB": list(range(7)), "scanC": list(range(3))}


def test_camera_only_samples_and_view_dataset(tmp_path):
    """eval.py's encode-once path: samples without pixels carry the same cameras (intrinsics scaled from the image header
    alone) and the view dataset yields every distinct image of a group once."""
    from patchmatchnet_amd.mvs import MVSViewDataset
    synth.write_scan(str(tmp_path), "scan1", n_views=5, H=64, W=96, n_src=3)
    with open(tmp_path / "list.txt", "w") as f:
        f.write("scan1\n")
    ds = MVSDataset(str(tmp_path), num_views=2, max_dim=48, scan_list=str(tmp_path / "list.txt"))
    full = ds[1]
    ds.load_images = False
    lean = ds[1]
    np.testing.assert_array_equal(full["intrinsics"], lean["intrinsics"])
    np.testing.assert_array_equal(full["extrinsics"], lean["extrinsics"])
    assert [tuple(x) for x in lean["images"]] == [im.shape[1:] for im in full["images"]]
    groups = ds.groups()
    assert list(groups) == [("scan1", "")] and groups[("scan1", "")] == list(range(5))
    views = ds.views_of(groups[("scan1", "")])
    assert views == sorted(set(views)) and set(views) <= set(range(5))
    vds = MVSViewDataset(ds, "scan1", "", views)
    assert len(vds) == len(views)
    item = vds[0]
    assert item["view"] == views[0] and item["image"].shape == full["images"][0].shape


def test_degenerate_depth_range_writes_nan_maps_or_is_refused(tmp_path):
    """The kernels' precondition (include/pmn_hip.h): 0 < depth_min < depth_max, finite.  eval.py refuses a sample with a degenerate
    range (the reference then divides by zero, models/patchmatch.py:656-657) while the values are host numbers; the dataset itself
    stays the reference's (it hands out whatever line 11 of the camera file holds, tests/test_reference_io.py)."""
    import importlib.util
    import pytest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("pmn_eval_cli", os.path.join(root, "eval.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    synth.write_scan(str(tmp_path), "scan1", n_views=3, H=64, W=96, n_src=2)
    with open(tmp_path / "list.txt", "w") as f:
        f.write("scan1\n")
    ds = MVSDataset(str(tmp_path), num_views=2, scan_list=str(tmp_path / "list.txt"))
    good = ds[1]
    ev._check_depth_range(good)
    ev._check_depth_range({"depth_min": torch.tensor([good["depth_min"]], dtype=torch.float64),  # the collated form
                           "depth_max": torch.tensor([good["depth_max"]], dtype=torch.float64), "filename": [good["filename"]]})
    for lo, hi in ((425.0, 425.0), (425.0, 2.5), (0.0, 935.0), (-1.0, 935.0), (425.0, float("inf")), (float("nan"), 935.0)):
        # default: the run goes on like the reference's -- the sample gets a stand-in range inside the kernels' domain and
        # is marked, so that _write_maps writes NaN maps for it (the reference writes inf / NaN maps there)
        s = dict(good, depth_min=np.float32(lo), depth_max=np.float32(hi))
        ev._check_depth_range(s)
        assert s["_degenerate"] == [True] and 0 < float(s["depth_min"]) < float(s["depth_max"])
        c = {"depth_min": torch.tensor([425.0, lo], dtype=torch.float64), "depth_max": torch.tensor([935.0, hi], dtype=torch.float64),
             "filename": ["a", "b"]}
        ev._check_depth_range(c)  # the collated form, one good and one bad element
        assert c["_degenerate"] == [False, True] and c["depth_min"].tolist()[0] == 425.0 and c["depth_max"].tolist() == [935.0, 2.0]
        # --strict_depth_range 1: round 5's behaviour
        ev.STRICT_DEPTH_RANGE[0] = True
        try:
            with pytest.raises(Exception, match="depth range"):
                ev._check_depth_range(dict(good, depth_min=np.float32(lo), depth_max=np.float32(hi)))
        finally:
            ev.STRICT_DEPTH_RANGE[0] = False


def _gather_worker(rank, world, port, H, W, ids, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, device = pdist.init_from_env("cpu")
    assert (r, w) == (rank, world)
    local = {vid: torch.full((2, H, W), float(vid)) + torch.arange(W).float() for vid in pdist.shard_views(ids, r, w)}
    out = pdist.gather_scan_maps(local, ids, H, W, device)
    ok = sorted(out) == sorted(ids) and all(
        torch.equal(out[v], torch.full((2, H, W), float(v)) + torch.arange(W).float()) for v in ids)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_scan_all_gather_world2_gloo():
    """The N>1 path: 2 ranks, 5 views (ragged: 3 + 2, one padding slot) gathered with one all_gather_into_tensor."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, 6, 8, [3, 7, 11, 20, 42], q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_gather_single_process_is_identity():
    local = {5: torch.rand(2, 4, 4), 9: torch.rand(2, 4, 4)}
    out = pdist.gather_scan_maps(local, [5, 9], 4, 4, torch.device("cpu"))
    assert torch.equal(out[5], local[5]) and torch.equal(out[9], local[9])


def test_uint8_images_are_the_float_images_times_255(tmp_path):
    """MVSDataset.uint8_images (eval.py's upload format): the bytes whose float32 division by 255 IS read_image's output; files
    that read_image would down-scale keep coming as float32."""
    data = str(tmp_path / "data")
    synth.write_scan(data, "s", n_views=3, H=48, W=64, n_src=2)
    with open(os.path.join(data, "list.txt"), "w") as f:
        f.write("s\n")
    ds = MVSDataset(data, num_views=2, scan_list=os.path.join(data, "list.txt"))
    ref = ds[0]
    ds.uint8_images = True
    got = ds[0]
    for a, b in zip(ref["images"], got["images"]):
        assert b.dtype == np.uint8 and b.shape == a.shape
        np.testing.assert_array_equal(b.astype(np.float32) / 255.0, a)
    np.testing.assert_array_equal(ref["intrinsics"], got["intrinsics"])
    small = MVSDataset(data, num_views=2, max_dim=32, scan_list=os.path.join(data, "list.txt"))
    small.uint8_images = True
    assert small[0]["images"][0].dtype == np.float32 and small[0]["images"][0].shape == (3, 24, 32)
    from patchmatchnet_amd.mvs import MVSViewDataset
    v = MVSViewDataset(ds, "s", "", [0, 1])[1]
    assert v["image"].dtype == np.uint8 and v["view"] == 1


def test_graph_replay_signature_tracks_shapes_and_aliasing():
    """GraphedForward keys its captured graphs by everything a capture bakes in: input shapes, which inputs are the SAME tensor
    (they share one static buffer) and the feature pyramids' shapes.  Pure host logic; the replay itself is a GPU test."""
    from patchmatchnet_amd.graph import GraphedForward as G
    a, b, c = torch.zeros(1, 3, 8, 8), torch.zeros(1, 3, 8, 8), torch.zeros(1, 3, 16, 8)
    K = torch.zeros(1, 3, 3, 3)
    assert G._alias_pattern([a, a, b, a, b]) == (0, 0, 2, 0, 2)
    assert G._signature([a, b, b], K, None) != G._signature([a, b, a], K, None)      # different aliasing
    assert G._signature([a, b, b], K, None) == G._signature([b, a, a], K, None)      # same shapes, same pattern
    assert G._signature([a, b], K, None) != G._signature([c, c.clone()], K, None)    # different sizes
    f1 = [{1: torch.zeros(1, 16, 4, 4), 2: torch.zeros(1, 32, 2, 2)}]
    f2 = [{1: torch.zeros(1, 16, 4, 4), 2: torch.zeros(1, 32, 2, 3)}]
    assert G._signature([a], K, f1) != G._signature([a], K, f2) and G._signature([a], K, f1) != G._signature([a], K, None)
    with pytest.raises(_lib.PmnError):
        G(None)([a], K, K, K, K)  # CPU tensors: there is no CPU path


def test_ply_writer_layout(tmp_path):
    v = np.random.default_rng(0).random((11, 3)).astype(np.float32)
    c = (np.random.default_rng(1).random((11, 3)) * 255).astype(np.uint8)
    ply = str(tmp_path / "o" / "fused.ply")
    fusion.write_ply(ply, v, c)
    raw = open(ply, "rb").read()
    assert raw.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex 11\n")
    body = raw.split(b"end_header\n", 1)[1]
    assert len(body) == 15 * 11
    rec = np.frombuffer(body, dtype=fusion.ply_records(v, c).dtype)
    np.testing.assert_array_equal(rec["x"], v[:, 0])
    np.testing.assert_array_equal(rec["blue"], c[:, 2])
    assert fusion.ply_header(11) + fusion.ply_records(v, c).tobytes() == raw
    # the streamed form eval.py uses (per-view record chunks, incl. an empty one) writes the same bytes; the bare form is the body
    chunks = [fusion.ply_records(v[:4], c[:4]), fusion.ply_records(v[:0], c[:0]), fusion.ply_records(v[4:], c[4:])]
    assert fusion.write_ply_records(str(tmp_path / "s.ply"), chunks) == 11
    assert open(str(tmp_path / "s.ply"), "rb").read() == raw
    fusion.write_ply_records(str(tmp_path / "s.part0"), chunks, header=False)
    assert open(str(tmp_path / "s.part0"), "rb").read() == body


def test_fusion_has_no_cpu_path():
    with pytest.raises(_lib.PmnError):
        fusion.fuse_views(torch.zeros(2, 2, 4, 4), {0: 0, 1: 1}, {}, {}, [], 1.0, 0.01, 1, 0.5)


def test_rescaled_projection_reproduces_the_per_view_warp():
    """ops.rescale_projection_rows (source maps of different sizes zero-padded into one buffer, CPU-checkable part): warping with
    the PADDED size and the rescaled projection rows lands on the positions the reference's warp computes with the view's OWN
    size (models/module.py:161-181: normalise by the reference map, un-normalise by the source map), to fp32 rounding."""
    import ctypes
    from oracle import oracle as O
    from patchmatchnet_amd import ops
    h, w, D = 24, 40, 6
    sizes, padded = [(24, 40), (16, 32), (20, 36)], (24, 40)
    intr, extr = synth.synthetic_cameras(4, h, w)
    proj = synth.stage_projections(intr, extr, 1.0)
    rels = []
    for v, (hv, wv) in enumerate(sizes):
        K = intr[0, v + 1].astype(np.float32).copy()
        K[0] *= wv / w
        K[1] *= hv / h
        pr = extr[0, v + 1].astype(np.float32).copy()
        pr[:3, :4] = K @ extr[0, v + 1, :3, :4].astype(np.float32)
        rels.append(np.matmul(pr, np.linalg.inv(proj[0, 0])).astype(np.float32))
    rel = torch.from_numpy(np.stack(rels))[None]  # [1,3,4,4]
    scaled = ops.rescale_projection_rows(rel, sizes, padded).numpy()
    assert ops.rescale_projection_rows(rel, [padded] * 3, padded) is rel  # nothing to do: same tensor
    depth = np.ascontiguousarray(np.random.default_rng(0).uniform(450, 900, (D, h, w)).astype(np.float32))
    f32p = ctypes.POINTER(ctypes.c_float)

    def positions(P, hs, ws):
        rot, trans = np.ascontiguousarray(P[:3, :3]), np.ascontiguousarray(P[:3, 3])
        ix, iy = np.empty((D, h, w), np.float32), np.empty((D, h, w), np.float32)
        O.lib().pmo_warp_positions(rot.ctypes.data_as(f32p), trans.ctypes.data_as(f32p), depth.ctypes.data_as(f32p), D, h, w, hs, ws,
                                   ix.ctypes.data_as(f32p), iy.ctypes.data_as(f32p))
        return ix, iy

    for v, (hv, wv) in enumerate(sizes):
        want = positions(rel[0, v].numpy(), hv, wv)          # the reference's arithmetic at the view's own size
        got = positions(scaled[0, v], *padded)               # what the kernel computes on the padded buffer
        for a, b in zip(got, want):
            assert np.abs(a - b).max() < 2e-3, (v, float(np.abs(a - b).max()))  # positions up to a few hundred px: fp32 rounding


def test_point_colours_are_the_decoded_bytes():
    """fusion takes point colours straight from the decoded BYTES when the image is not down-scaled; the reference goes byte ->
    float32 / 255.0 (datasets/data_io.py:45) -> (color * 255).astype(uint8) (eval.py:275).  That round trip is the identity for
    every byte value, so both give the same colours."""
    k = np.arange(256, dtype=np.uint8)
    img = np.array(k, dtype=np.float32) / 255.0
    np.testing.assert_array_equal((img * 255).astype(np.uint8), k)


# ---- round 4 host logic ---------------------------------------------------------------------------------------------------------

def test_resize_exact_half_takes_the_area_path_and_other_scales_use_opencvs_scale():
    """cv::resize(INTER_LINEAR) switches to INTER_AREA for an exact 2x down-scale (resize.cpp) and otherwise derives its source
    step as 1 / (dst / src) in double (reference datasets/data_io.py:26-29 calls it through cv2.resize)."""
    from patchmatchnet_amd import data_io
    rng = np.random.default_rng(0)
    img = rng.random((12, 16, 3)).astype(np.float32)
    got = data_io.resize_bilinear(img, 6, 8)
    want = (((img[0::2, 0::2] + img[0::2, 1::2]) + img[1::2, 0::2]) + img[1::2, 1::2]) * np.float32(0.25)
    assert got.dtype == np.float32 and np.array_equal(got, want)
    # not an exact half: half-pixel-centre bilinear taps; 7 -> 3 samples columns 0.6667, 3.0, 5.3333
    s0, s1, w0, w1 = data_io._linear_taps(3, 7)
    assert list(s0) == [0, 3, 5] and list(s1) == [1, 4, 6]
    assert abs(float(w1[0]) - 2.0 / 3.0) < 1e-6 and float(w1[1]) == 0.0  # 3.0 exactly: the integer position keeps weight 0 on the right


def test_single_rank_rendezvous_takes_a_free_port_and_several_ranks_need_one(monkeypatch):
    """dist.init_from_env: WORLD_SIZE=1 without MASTER_PORT binds an ephemeral port (two single-rank jobs on one host used to collide
    on a fixed default); WORLD_SIZE>1 without MASTER_PORT is an error instead of a silent default."""
    import torch.distributed as tdist
    from patchmatchnet_amd import dist as pdist
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.delenv("MASTER_PORT", raising=False)
    with pytest.raises(RuntimeError, match="MASTER_PORT"):
        pdist.init_from_env("cpu")
    monkeypatch.setenv("WORLD_SIZE", "1")
    try:
        rank, world, device = pdist.init_from_env("cpu")
        assert (rank, world, device.type) == (0, 1, "cpu") and tdist.is_initialized()
        assert int(os.environ["MASTER_PORT"]) > 1024
    finally:
        if tdist.is_initialized():
            tdist.destroy_process_group()
        os.environ.pop("MASTER_PORT", None)
    assert pdist.bind_to_device_node(torch.device("cpu")) == "numa binding off"


def test_f16_split_refuses_weights_outside_float16s_range_and_the_modules_fall_back():
    """params.split_f16 raises F16DomainError at |x| >= 65504; FeatureNet / Refinement then pack without the fp16-split operands,
    remember why, and warn (the forward itself needs a GPU: tests/test_hip_parity.py)."""
    import warnings
    import patchmatchnet_amd as P
    from patchmatchnet_amd import params as PP
    hi, lo = PP.split_f16(np.array([1.0, -3.5e4, 6.0e-8]))
    assert hi.dtype == np.float16 and np.isfinite(hi).all() and np.isfinite(lo).all()
    for bad in (65504.0, -7.0e4, np.inf, np.nan):
        with pytest.raises(PP.F16DomainError):
            PP.split_f16(np.array([0.5, bad]))
    net = P.net.FeatureNet().eval()
    with torch.no_grad():
        net.conv6.bn.weight.fill_(5.0e7)
        net.conv6.bn.running_var.fill_(1.0)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        pk = net._packed()
    assert net.f16_domain_error is not None and "conv6" in net.f16_domain_error
    assert not any(k.endswith("_f16s") for k in pk) and any("fp32 kernels" in str(w.message) for w in caught)
    ok = P.net.FeatureNet().eval()
    assert ok.f16_domain_error is None and "conv6_f16s" in ok._packed() and ok.f16_domain_error is None
    ref = P.net.Refinement().eval()
    with torch.no_grad():
        ref.conv3.bn.weight.fill_(9.0e7)
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        assert "conv3_f16s" not in ref._packed() and "conv3" in ref.f16_domain_error


def test_adjust_image_dims_mutates_like_the_reference():
    """adjust_image_dims on CPU tensors against what the REFERENCE left behind on the same 100 x 130 sample (tests/golden/
    cascade_resized_100x130.npz: reference models/net.py:304-318): the caller's intrinsics, rescaled in place, bit for bit; the
    caller's image list, now holding the stretched 96 x 128 images; the original size returned for the outputs."""
    import goldenutil as GU
    from patchmatchnet_amd.net import adjust_image_dims
    g = GU.load_npz("cascade_resized_100x130.npz")
    imgs = synth.synthetic_images(3, 100, 130)
    intr, _ = synth.synthetic_cameras(3, 100, 130)
    K = torch.from_numpy(intr.copy())
    out, K2, h0, w0 = adjust_image_dims(imgs, K)
    assert K2 is K and (h0, w0) == (100, 130)
    np.testing.assert_array_equal(K.numpy(), g["intrinsics_after"])
    assert all(tuple(im.shape) == (1, 3, 96, 128) for im in out) and out is imgs
    want = torch.nn.functional.interpolate(synth.synthetic_images(3, 100, 130)[1], size=[96, 128], mode="bilinear", align_corners=False)
    assert torch.equal(out[1], want)
