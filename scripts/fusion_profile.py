#!/usr/bin/env python
"""Where a scan's fusion stage goes (eval.py --output_type both on one generated 49-view 1600x1200 scan): wall time of
fusion.fuse_views (kernels + host halves on the pool), of the PLY write, the number of fused points, and the device time of
pmn_fuse_view per view.   python scripts/fusion_profile.py [n_views=49]"""
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import synth  # noqa: E402
import eval as pm_eval  # noqa: E402
from patchmatchnet_amd import fusion, ops  # noqa: E402

n_views = int(sys.argv[1]) if len(sys.argv) > 1 else 49
base = "/dev/shm/pmn_fusion_profile" if os.path.isdir("/dev/shm") else "/tmp/pmn_fusion_profile"
shutil.rmtree(base, ignore_errors=True)
data = os.path.join(base, "data")
synth.write_scene_scan(data, "scan1", n_views, 1200, 1600, n_src=10, seed=0, device="cuda")
open(os.path.join(data, "list.txt"), "w").write("scan1\n")
ckpt = os.path.join(ROOT, "tests", "golden", "params_000007.npz")
T = {}


def timed(mod, name):
    inner = getattr(mod, name)

    def wrapper(*a, **k):
        t = time.time()
        r = inner(*a, **k)
        T.setdefault(name, []).append(time.time() - t)
        return r
    setattr(mod, name, wrapper)


kernel_ms = []
inner_fuse_view = ops.fuse_view


def fuse_view_timed(*a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = inner_fuse_view(*a, **k)
    e1.record()
    kernel_ms.append((e0, e1))
    return r


ops.fuse_view = fuse_view_timed
for name in ("fuse_views", "write_ply", "write_ply_records"):
    if hasattr(fusion, name):
        timed(fusion, name)
timed(pm_eval, "save_image")
for rep in range(2):
    T.clear()
    kernel_ms.clear()
    out = os.path.join(base, "out")
    shutil.rmtree(out, ignore_errors=True)
    pm_eval.main(["--input_folder", data, "--output_folder", out, "--checkpoint_path", ckpt, "--scan_list",
                  os.path.join(data, "list.txt"), "--num_views", "5", "--file_format", ".pfm", "--output_type", "both",
                  "--geo_mask_thres", "3"])
    torch.cuda.synchronize()
    ply = os.path.join(out, "scan1", "fused.ply")
    print("PROFILE rep %d: %s; pmn_fuse_view %.3f ms per view (events); fused.ply %.1f MB = %d points; mask PNGs %.1f ms each "
          "(sum over threads %.2f s)" % (
              rep, {k: round(sum(v), 3) for k, v in T.items() if k != "save_image"},
              float(np.mean([a.elapsed_time(b) for a, b in kernel_ms])), os.path.getsize(ply) / 1e6,
              (os.path.getsize(ply) - 200) // 15, 1e3 * float(np.mean(T["save_image"])), sum(T["save_image"])), flush=True)
shutil.rmtree(base, ignore_errors=True)
