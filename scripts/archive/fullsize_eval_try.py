import os, sys, tempfile, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
import eval as pm_eval
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    data = os.path.join(tmp, "data")
    t = time.time()
    synth.write_scan(data, "scan1", n_views=12, H=1200, W=1600, n_src=5)
    print("scan written in %.1f s" % (time.time() - t), flush=True)
    with open(os.path.join(data, "list.txt"), "w") as f:
        f.write("scan1\n")
    ckpt = os.path.join(ROOT, "tests", "golden", "params_000007.npz")
    for extra in (["--feature_cache", "64"], ["--feature_cache", "0"]):
        out = os.path.join(tmp, "out" + extra[1])
        t = time.time()
        pm_eval.main(["--input_folder", data, "--output_folder", out, "--checkpoint_path", ckpt, "--scan_list",
                      os.path.join(data, "list.txt"), "--num_views", "5", "--num_workers", "0", "--geo_mask_thres", "2"] + extra)
        print("RESULT %s: depth + fusion of 12 views in %.2f s; fused.ply %d bytes" % (extra, time.time() - t, os.path.getsize(os.path.join(out, "scan1", "fused.ply"))), flush=True)
