"""The host code either side of the hot path (on-disk formats, the sample source) pinned against
the REFERENCE's own ``datasets/data_io.py`` and ``datasets/mvs.py`` -- not against byte layouts written down in this repository.

Live part (authoring container: /root/reference is imported read-only, cv2 stubbed -- the only cv2 call of these modules is
``cv2.resize`` in scale_to_max_dim, reference datasets/data_io.py:26-29): sample dicts of MVSDataset, camera / pair parsing, PFM /
.bin bytes both ways, mask images.  Committed part (runs anywhere, incl. the GPU box): tests/golden/io_reference.npz holds files
WRITTEN BY THE REFERENCE for seeded arrays (``python tests/test_reference_io.py`` regenerates it); this repository's writers
must produce the same bytes and its readers must read them back.  ``cv2.resize`` itself is unpinned (no OpenCV here, unpinned
in the reference's requirements): patchmatchnet_amd.data_io.resize_bilinear restates OpenCV's float resizeLinear and is pinned
to a hand-computed vector below.
"""
import importlib
import os
import sys
import types

import numpy as np
import pytest

import refutil
import synth
from patchmatchnet_amd import data_io
from patchmatchnet_amd.mvs import MVSDataset

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden", "io_reference.npz")
needs_reference = pytest.mark.skipif(not refutil.have_reference(), reason="reference checkout not present")


def reference_datasets():
    """The reference's datasets.data_io and datasets.mvs, imported (never copied); cv2 is stubbed with this repository's
    restatement of cv2.resize for the one call site that needs it (--image_max_dim)."""
    saved = {k: sys.modules.get(k) for k in ("cv2", "datasets", "datasets.data_io", "datasets.mvs")}
    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR = 1
    cv2.resize = lambda img, size, interpolation=None: data_io.resize_bilinear(img, size[1], size[0])
    sys.modules["cv2"] = cv2
    for k in ("datasets", "datasets.data_io", "datasets.mvs"):
        sys.modules.pop(k, None)
    sys.path.insert(0, refutil.REFERENCE_ROOT)
    try:
        ref_io = importlib.import_module("datasets.data_io")
        ref_mvs = importlib.import_module("datasets.mvs")
    finally:
        sys.path.remove(refutil.REFERENCE_ROOT)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return ref_io, ref_mvs


def seeded_maps():
    rng = np.random.default_rng(42)
    return {"hw": rng.standard_normal((13, 21)).astype(np.float32) * 700.0,
            "hw1": rng.random((9, 14, 1)).astype(np.float32),
            "hw3": rng.random((6, 5, 3)).astype(np.float32),
            "tall": np.ascontiguousarray((rng.standard_normal((31, 7)) * 3).astype(np.float32)[:, ::-1])}  # a non-contiguous map


def write_scan_variants(root):
    """Generated scans that exercise the parser edge cases the reference handles: a viewpoint without source views (dropped), more
    source views than --num_views and fewer, a camera file without the depth line, one with four depth parameters, light folders."""
    synth.write_scan(root, "scanA", n_views=5, H=48, W=64, n_src=3)
    pair = os.path.join(root, "scanA", "pair.txt")
    lines = open(pair).read().split("\n")
    lines[0] = "6"
    lines += ["7", "0 "]  # viewpoint 7 has no source views: dropped by both readers
    lines = [ln for ln in lines if ln != ""]
    open(pair, "w").write("\n".join(lines) + "\n")
    cam = os.path.join(root, "scanA", "cams", "00000002_cam.txt")
    txt = open(cam).read().rstrip("\n").split("\n")
    txt[-1] = "425.0 2.5 192 935.0"  # MVSNet-style depth line: min, interval, planes, max
    open(cam, "w").write("\n".join(txt) + "\n")
    # lights: images/<light>/<id>.jpg
    synth.write_scan(root, "scanL", n_views=3, H=40, W=56, n_src=2)
    for light in ("0", "1"):
        os.makedirs(os.path.join(root, "scanL", "images", light), exist_ok=True)
        for v in range(3):
            src = os.path.join(root, "scanL", "images", "{:0>8}.jpg".format(v))
            dst = os.path.join(root, "scanL", "images", light, "{:0>8}.jpg".format((v + int(light)) % 3))
            open(dst, "wb").write(open(src, "rb").read())
    with open(os.path.join(root, "list.txt"), "w") as f:
        f.write("scanA\nscanL\n")


def assert_samples_equal(ours, theirs):
    assert len(ours["images"]) == len(theirs["images"])
    for a, b in zip(ours["images"], theirs["images"]):
        assert a.dtype == b.dtype == np.float32 and a.shape == b.shape
        np.testing.assert_array_equal(a, b)
    for k in ("intrinsics", "extrinsics"):
        assert ours[k].dtype == theirs[k].dtype and ours[k].shape == theirs[k].shape
        np.testing.assert_array_equal(ours[k], theirs[k])
    for k in ("depth_min", "depth_max"):
        assert np.float32(ours[k]) == np.float32(theirs[k]) and type(ours[k]) is type(theirs[k])
    assert ours["filename"] == theirs["filename"]


@needs_reference
def test_cam_and_pair_files_parse_like_the_reference(tmp_path):
    ref_io, _ = reference_datasets()
    write_scan_variants(str(tmp_path))
    for scan, views in (("scanA", range(5)), ("scanL", range(3))):
        for v in views:
            p = os.path.join(str(tmp_path), scan, "cams", "{:0>8}_cam.txt".format(v))
            for a, b in zip(data_io.read_cam_file(p), ref_io.read_cam_file(p)):
                assert a.dtype == b.dtype and a.shape == b.shape
                np.testing.assert_array_equal(a, b)
        p = os.path.join(str(tmp_path), scan, "pair.txt")
        assert data_io.read_pair_file(p) == ref_io.read_pair_file(p)
    assert 7 not in [r for r, _ in data_io.read_pair_file(os.path.join(str(tmp_path), "scanA", "pair.txt"))]
    # no depth line at all: both return an empty vector
    short = str(tmp_path / "short_cam.txt")
    txt = open(os.path.join(str(tmp_path), "scanA", "cams", "00000000_cam.txt")).read().rstrip("\n").split("\n")[:10]
    open(short, "w").write("\n".join(txt) + "\n")
    assert data_io.read_cam_file(short)[2].size == ref_io.read_cam_file(short)[2].size == 0


@needs_reference
@pytest.mark.parametrize("num_views,max_dim,lights,scan_list", [(2, -1, -1, True), (10, -1, -1, True), (2, -1, 2, True),
                                                               (3, 40, -1, True), (2, -1, -1, False)])
def test_mvsdataset_samples_equal_the_reference(tmp_path, num_views, max_dim, lights, scan_list):
    """datasets/mvs.py:34-111: every sample dict field the inference path reads, for every index (images bit-exact without
    resize; with --image_max_dim through the restated cv2.resize on both sides, which pins the scaling rule, the intrinsics
    rescale and the original-size bookkeeping)."""
    _, ref_mvs = reference_datasets()
    root = str(tmp_path)
    write_scan_variants(root)
    if scan_list:
        path, sl = root, os.path.join(root, "list.txt")
        if lights <= 0:
            sl_use = sl
        else:  # only scanL has light folders
            sl_use = os.path.join(root, "listL.txt")
            open(sl_use, "w").write("scanL\n")
    else:
        path, sl_use = os.path.join(root, "scanA"), ""  # no scan list: the data path IS the scan (scans = [''])
    kw = dict(num_views=num_views, max_dim=max_dim, scan_list=sl_use, num_light_idx=lights)
    ours, theirs = MVSDataset(path, **kw), ref_mvs.MVSDataset(path, **kw)
    assert len(ours) == len(theirs) and ours.metas == theirs.metas
    for i in range(len(ours)):
        assert_samples_equal(ours[i], theirs[i])
    if num_views == 10:
        assert len(ours[0]["images"]) == 1 + 3  # fewer source views available than asked for


@needs_reference
def test_map_writers_and_readers_are_byte_compatible_with_the_reference(tmp_path):
    """datasets/data_io.py:165-223 (.bin), :226-302 (PFM), :50-64 (save_image): same bytes out, same arrays in, both ways."""
    ref_io, _ = reference_datasets()
    for name, a in seeded_maps().items():
        for ext in (".pfm", ".bin"):
            mine, theirs = str(tmp_path / (name + "_mine" + ext)), str(tmp_path / (name + "_ref" + ext))
            data_io.save_map(mine, a)
            ref_io.save_map(theirs, a)
            assert open(mine, "rb").read() == open(theirs, "rb").read(), (name, ext)
            for reader, path in ((data_io.read_map, theirs), (ref_io.read_map, mine)):
                got = reader(path)
                assert got.shape == (a.shape[0], a.shape[1], 1 if a.ndim == 2 else a.shape[2])
                np.testing.assert_array_equal(got.reshape(a.shape), a)
        if a.ndim == 2:  # the device-flipped PFM path of eval.py's MapWriter
            flipped = str(tmp_path / (name + "_flip.pfm"))
            data_io.save_map(flipped, np.ascontiguousarray(a[::-1]), rows_flipped=True)
            assert open(flipped, "rb").read() == open(str(tmp_path / (name + "_ref.pfm")), "rb").read()
    rng = np.random.default_rng(5)
    for name, img in (("mask", rng.random((11, 9)) > 0.5), ("float", rng.random((8, 6, 3)).astype(np.float32)),
                      ("int", (rng.random((5, 7)) * 300).astype(np.int64))):
        mine, theirs = str(tmp_path / (name + "_mine.png")), str(tmp_path / (name + "_ref.png"))
        data_io.save_image(mine, img)
        ref_io.save_image(theirs, img)
        assert open(mine, "rb").read() == open(theirs, "rb").read(), name
    with pytest.raises(Exception):
        ref_io.save_map(str(tmp_path / "bad.pfm"), np.zeros((2, 2), np.float64))
    with pytest.raises(Exception):
        data_io.save_map(str(tmp_path / "bad.pfm"), np.zeros((2, 2), np.float64))


@needs_reference
def test_read_image_flow_equals_the_reference(tmp_path):
    """datasets/data_io.py:13-47: decode, /255 in float32, the scale rule (int() truncation of both sides, only when 0 < scale
    < 1) and the returned original size; eval.py's uint8 fast path reproduces the same float image after the device division."""
    ref_io, _ = reference_datasets()
    synth.write_scan(str(tmp_path), "s", n_views=2, H=50, W=70, n_src=1)
    p = os.path.join(str(tmp_path), "s", "images", "00000001.jpg")
    for max_dim in (-1, 0, 70, 100, 69, 35, 33):
        a, ha, wa = data_io.read_image(p, max_dim)
        b, hb, wb = ref_io.read_image(p, max_dim)
        assert (ha, wa) == (hb, wb) == (50, 70) and a.shape == b.shape and a.dtype == b.dtype
        np.testing.assert_array_equal(a, b)
        assert data_io.image_shape(p, max_dim) == (a.shape[0], a.shape[1], 50, 70)
    u8 = data_io.read_image_u8(p, -1)
    np.testing.assert_array_equal(u8.astype(np.float32) / 255.0, ref_io.read_image(p, -1)[0])
    assert data_io.read_image_u8(p, 35) is None


def test_resize_is_opencvs_float_linear_on_a_hand_computed_vector():
    """cv2.resize(INTER_LINEAR) restated: 1x5 -> 1x3 (scale 5/3).  fx = (d + 0.5) * 5/3 - 0.5 = 0.3333, 2.0, 3.6667 ->
    taps (0,1,w=.3333), (2,3,w=0), (3,4,w=.6667); 4 rows -> 2 rows: fy = 0.5, 2.5 -> rows (0,1,.5), (2,3,.5)."""
    src = np.array([[0, 10, 20, 30, 40]], np.float32) + np.array([[0], [100], [200], [300]], np.float32)
    got = data_io.resize_bilinear(src, 2, 3)
    fx = np.float32((0 + 0.5) * (5.0 / 3.0) - 0.5)
    fx2 = np.float32((2 + 0.5) * (5.0 / 3.0) - 0.5) - np.float32(3.0)
    row = np.array([0 * (np.float32(1) - fx) + np.float32(10) * fx, 20.0, np.float32(30) * (np.float32(1) - fx2) + np.float32(40) * fx2],
                   np.float32)
    want = np.stack([(row + 0) * np.float32(0.5) + (row + 100) * np.float32(0.5),
                     (row + 200) * np.float32(0.5) + (row + 300) * np.float32(0.5)]).astype(np.float32)
    np.testing.assert_allclose(got, want, rtol=2e-7, atol=0)  # (float32 association of the +100 row offsets)
    assert abs(float(got[0, 0]) - (50.0 + 10.0 / 3.0)) < 1e-4 and abs(float(got[1, 2]) - (250.0 + 30.0 + 20.0 / 3.0)) < 1e-4
    # up-scaling clamps at the borders (s < 0 -> tap 0 with weight 1; s >= src - 1 -> the last pixel)
    up = data_io.resize_bilinear(np.array([[1.0, 3.0]], np.float32), 1, 4)
    np.testing.assert_allclose(up, [[1.0, 1.5, 2.5, 3.0]], atol=1e-6)


def make_golden():
    """Files written by the REFERENCE's writers for the seeded arrays (authoring container)."""
    import tempfile
    ref_io, _ = reference_datasets()
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for name, a in seeded_maps().items():
            for ext in (".pfm", ".bin"):
                p = os.path.join(d, name + ext)
                ref_io.save_map(p, a)
                out[name + ext] = np.frombuffer(open(p, "rb").read(), np.uint8)
        mask = np.random.default_rng(5).random((11, 9)) > 0.5
        p = os.path.join(d, "mask.png")
        ref_io.save_image(p, mask)
        out["mask.png"] = np.frombuffer(open(p, "rb").read(), np.uint8)
    np.savez_compressed(GOLDEN, **out)
    print("wrote", GOLDEN)


def test_writers_reproduce_the_reference_written_golden_files(tmp_path):
    """Runs anywhere (no reference needed): tests/golden/io_reference.npz = bytes the reference's own save_map / save_image wrote."""
    g = np.load(GOLDEN)
    for name, a in seeded_maps().items():
        for ext in (".pfm", ".bin"):
            p = str(tmp_path / (name + ext))
            data_io.save_map(p, a)
            assert open(p, "rb").read() == g[name + ext].tobytes(), (name, ext)
            ref_file = str(tmp_path / ("ref_" + name + ext))
            open(ref_file, "wb").write(g[name + ext].tobytes())
            np.testing.assert_array_equal(data_io.read_map(ref_file).reshape(a.shape), a)
    mask = np.random.default_rng(5).random((11, 9)) > 0.5
    p = str(tmp_path / "mask.png")
    data_io.save_image(p, mask)
    assert open(p, "rb").read() == g["mask.png"].tobytes()


if __name__ == "__main__":
    make_golden()
