#!/usr/bin/env python
"""eval.py on a generated 1600x1200 scan (16 views, 5 source views each): plain path vs encode-once path, wall time of the
depth stage (decode + forward + PFM write).  Development aid; everything lives under a temp dir."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
import eval as pm_eval

n_views = int(sys.argv[1]) if len(sys.argv) > 1 else 16
with tempfile.TemporaryDirectory() as tmp:
    data = os.path.join(tmp, "data")
    synth.write_scan(data, "scan1", n_views=n_views, H=1200, W=1600, n_src=5)
    with open(os.path.join(data, "list.txt"), "w") as f:
        f.write("scan1\n")
    ckpt = os.path.join(ROOT, "tests", "golden", "params_000007.npz")
    for cache in ("0", "64", "0", "64"):
        out = os.path.join(tmp, "out" + cache)
        t = time.time()
        pm_eval.main(["--input_folder", data, "--output_folder", out, "--checkpoint_path", ckpt, "--scan_list",
                      os.path.join(data, "list.txt"), "--num_views", "5", "--output_type", "depth", "--num_workers", os.environ.get("EVAL_WORKERS", "0"),
                      "--feature_cache", cache, "--file_format", ".pfm"])
        dt = time.time() - t
        print(f"RESULT feature_cache={cache}: {dt:.2f} s for {n_views} samples -> {n_views / dt:.1f} samples/s", flush=True)
