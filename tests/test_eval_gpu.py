"""GPU test of the CLI boundary: eval.py on a generated DTU-layout scan writes the reference's output files and the maps
equal a direct PatchmatchNet.forward call."""
import os
import sys

import numpy as np
import pytest
import torch

import goldenutil as GU
import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eval_cli_end_to_end(tmp_path):
    assert torch.cuda.is_available(), "GPU tests selected but no ROCm device is visible"
    sys.path.insert(0, ROOT)
    import eval as pm_eval
    import patchmatchnet_amd as P
    from patchmatchnet_amd import data_io
    from patchmatchnet_amd.mvs import MVSDataset

    data = str(tmp_path / "data")
    out = str(tmp_path / "out")
    synth.write_scan(data, "scan9", n_views=4, H=96, W=128, n_src=2)
    with open(os.path.join(data, "list.txt"), "w") as f:
        f.write("scan9\n")
    ckpt = os.path.join(GU.GOLDEN_DIR, "params_000007.npz")
    torch.manual_seed(3)
    pm_eval.main(["--input_folder", data, "--output_folder", out, "--checkpoint_path", ckpt, "--scan_list",
                  os.path.join(data, "list.txt"), "--num_views", "2", "--geo_mask_thres", "1", "--photo_thres", "0.1",
                  "--num_workers", "0", "--file_format", ".pfm"])
    for v in range(4):
        for kind in ("depth_est", "confidence"):
            p = os.path.join(out, "scan9", kind, "{:0>8}.pfm".format(v))
            assert os.path.isfile(p), p
            m = data_io.read_map(p)
            assert m.shape == (96, 128, 1) and np.isfinite(m).all()
        assert os.path.isfile(os.path.join(out, "scan9", "mask", "{:0>8}_final.png".format(v)))
    assert os.path.isfile(os.path.join(out, "scan9", "fused.ply"))
    # the written depth of view 0 equals a direct forward with the same seed (same RNG draw for the stage-3 noise)
    _, params, kw = GU.load_case("default")
    model = P.PatchmatchNet(**kw)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    model = model.cuda().eval()
    s = MVSDataset(data, num_views=2, scan_list=os.path.join(data, "list.txt"))[0]
    torch.manual_seed(3)
    with torch.no_grad():
        depth, conf, _ = model([torch.from_numpy(i)[None].cuda() for i in s["images"]],
                               torch.from_numpy(s["intrinsics"])[None].cuda(), torch.from_numpy(s["extrinsics"])[None].cuda(),
                               torch.tensor([s["depth_min"]]).cuda(), torch.tensor([s["depth_max"]]).cuda())
    got = data_io.read_map(os.path.join(out, "scan9", "depth_est", "00000000.pfm"))[..., 0]
    np.testing.assert_array_equal(got, depth[0, 0].cpu().numpy())
    # the per-scan feature cache must not change a single bit: rerun without it and compare every written map
    out2 = str(tmp_path / "out_nocache")
    torch.manual_seed(3)
    pm_eval.main(["--input_folder", data, "--output_folder", out2, "--checkpoint_path", ckpt, "--scan_list",
                  os.path.join(data, "list.txt"), "--num_views", "2", "--output_type", "depth", "--num_workers", "0",
                  "--feature_cache", "0"])
    for v in range(4):
        for kind in ("depth_est", "confidence"):
            a = data_io.read_map(os.path.join(out, "scan9", kind, "{:0>8}.pfm".format(v)))
            b = data_io.read_map(os.path.join(out2, "scan9", kind, "{:0>8}.pfm".format(v)))
            np.testing.assert_array_equal(a, b)
    assert float(depth.min()) >= 425.0 * 0.9 and float(depth.max()) <= 935.0 * 1.1


def test_uint8_upload_restores_numpy_division():
    """eval.py uploads decoded bytes and divides by 255 on the device; every byte value must give numpy's float32 x / 255.0
    (what the reference's read_image feeds the network, datasets/data_io.py:34-47)."""
    assert torch.cuda.is_available(), "GPU tests selected but no ROCm device is visible"
    sys.path.insert(0, ROOT)
    import eval as pm_eval
    pre = pm_eval.DevicePrefetcher([], torch.device("cuda:0"))
    got = pre._upload(torch.arange(256, dtype=torch.uint8)).cpu().numpy()
    np.testing.assert_array_equal(got, np.arange(256, dtype=np.uint8).astype(np.float32) / 255.0)


def test_hip_graph_replay_writes_the_same_bytes(tmp_path):
    """eval.py --hip_graph 1 (default: one HIP-graph replay per sample, patchmatchnet_amd/graph.py) against --hip_graph 0 (every
    kernel launched from Python): every map byte-identical, in the plain path and in the encode-once path, with the stage-3
    random draw seeded per sample (the captured Philox kernel must draw what the eager forward draws)."""
    assert torch.cuda.is_available(), "GPU tests selected but no ROCm device is visible"
    sys.path.insert(0, ROOT)
    import eval as pm_eval
    from patchmatchnet_amd import data_io
    data = str(tmp_path / "data")
    synth.write_scan(data, "scanG", n_views=5, H=96, W=128, n_src=2)
    with open(os.path.join(data, "list.txt"), "w") as f:
        f.write("scanG\n")
    ckpt = os.path.join(GU.GOLDEN_DIR, "params_000007.npz")
    outs = {}
    for graph in ("0", "1"):
        for cache in ("0", "64"):
            out = str(tmp_path / ("out_g" + graph + "_c" + cache))
            pm_eval.main(["--input_folder", data, "--output_folder", out, "--checkpoint_path", ckpt, "--scan_list",
                          os.path.join(data, "list.txt"), "--num_views", "2", "--output_type", "depth", "--num_workers", "0",
                          "--sample_seed", "5", "--hip_graph", graph, "--feature_cache", cache])
            outs[(graph, cache)] = out
    base = outs[("0", "0")]
    for key, out in outs.items():
        for v in range(5):
            for kind in ("depth_est", "confidence"):
                a = open(os.path.join(base, "scanG", kind, "{:0>8}.pfm".format(v)), "rb").read()
                b = open(os.path.join(out, "scanG", kind, "{:0>8}.pfm".format(v)), "rb").read()
                assert a == b, (key, kind, v)
    d = data_io.read_map(os.path.join(base, "scanG", "depth_est", "00000003.pfm"))
    assert np.isfinite(d).all() and d.min() > 300


def test_samples_in_flight_draw_their_own_noise(tmp_path):
    """--in_flight 3 against --in_flight 1 at a size where the GPU, not the host, is the bottleneck (slots really overlap): every
    map byte-identical with --sample_seed.  The stage-3 draw of a replayed forward is made eagerly per slot (graph.py); a Philox
    kernel captured INSIDE the graphs would share one seed / offset pair between all slots and a slot could draw with its
    neighbour's offset."""
    assert torch.cuda.is_available(), "GPU tests selected but no ROCm device is visible"
    sys.path.insert(0, ROOT)
    import eval as pm_eval
    data = str(tmp_path / "data")
    synth.write_scan(data, "scanF", n_views=7, H=576, W=768, n_src=4)
    with open(os.path.join(data, "list.txt"), "w") as f:
        f.write("scanF\n")
    ckpt = os.path.join(GU.GOLDEN_DIR, "params_000007.npz")
    outs = {}
    for flight in ("1", "3"):
        out = str(tmp_path / ("out_f" + flight))
        pm_eval.main(["--input_folder", data, "--output_folder", out, "--checkpoint_path", ckpt, "--scan_list",
                      os.path.join(data, "list.txt"), "--num_views", "4", "--output_type", "depth", "--num_workers", "0",
                      "--sample_seed", "7", "--hip_graph", "1", "--in_flight", flight])
        outs[flight] = out
    for v in range(7):
        for kind in ("depth_est", "confidence"):
            a = open(os.path.join(outs["1"], "scanF", kind, "{:0>8}.pfm".format(v)), "rb").read()
            b = open(os.path.join(outs["3"], "scanF", kind, "{:0>8}.pfm".format(v)), "rb").read()
            assert a == b, (kind, v)


def _run_eval(cmd_args, env_extra, cwd):
    import subprocess
    env = dict(os.environ)
    env.update(env_extra)
    return subprocess.Popen([sys.executable, os.path.join(ROOT, "eval.py")] + cmd_args, env=env, cwd=cwd,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def test_two_ranks_on_one_gpu_write_the_same_bytes(tmp_path):
    """The multi-rank path for real: 2 processes (gloo, both on cuda:0) run eval.py over TWO scans whose view counts do not
    divide by the world size; every map, mask and fused.ply must be byte-identical to the single-process run (block sharding,
    per-scan all-gather, per-rank fusion, PLY stitching; reference eval.py:33, 373-383)."""
    assert torch.cuda.is_available(), "GPU tests selected but no ROCm device is visible"
    data = str(tmp_path / "data")
    synth.write_scan(data, "scanA", n_views=5, H=96, W=128, n_src=2)
    synth.write_scan(data, "scanB", n_views=3, H=96, W=128, n_src=2)
    with open(os.path.join(data, "list.txt"), "w") as f:
        f.write("scanA\nscanB\n")
    ckpt = os.path.join(GU.GOLDEN_DIR, "params_000007.npz")
    common = ["--input_folder", data, "--checkpoint_path", ckpt, "--scan_list", os.path.join(data, "list.txt"), "--num_views", "2",
              "--geo_mask_thres", "1", "--photo_thres", "0.1", "--num_workers", "0", "--sample_seed", "11"]
    out1, out2 = str(tmp_path / "out1"), str(tmp_path / "out2")
    p = _run_eval(common + ["--output_folder", out1], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, str(tmp_path))
    log = p.communicate(timeout=900)[0]
    assert p.returncode == 0, log[-3000:]
    port = str(29600 + os.getpid() % 1000)
    procs = [_run_eval(common + ["--output_folder", out2],
                       {"WORLD_SIZE": "2", "RANK": str(r), "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port,
                        "PMN_DIST_BACKEND": "gloo"}, str(tmp_path)) for r in range(2)]
    logs = [q.communicate(timeout=900)[0] for q in procs]
    for q, lg in zip(procs, logs):
        assert q.returncode == 0, lg[-3000:]
    n_files = 0
    for root, _, files in os.walk(out1):
        for name in files:
            a = os.path.join(root, name)
            b = os.path.join(out2, os.path.relpath(a, out1))
            assert os.path.isfile(b), b
            assert open(a, "rb").read() == open(b, "rb").read(), b
            n_files += 1
    assert n_files == 2 + 8 * 2 + 8 * 3  # 2 fused.ply, depth + confidence and 3 masks for each of the 8 views
    assert not [n for _, _, fs in os.walk(out2) for n in fs if ".part" in n]
    assert os.path.getsize(os.path.join(out1, "scanA", "fused.ply")) > 1000
