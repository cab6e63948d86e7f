#!/usr/bin/env python
"""eval.py's maps against per-sample EAGER forwards under the same seeds, at full size (one generated 1600x1200 scan of V views):
does the product pipeline (feature cache, graph replay, samples in flight, uploads / downloads / decode threads beside it) hand out
the bits of the plain forward?   python scripts/eval_verify_probe.py [views] [extra eval flags...]"""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import goldenutil as GU  # noqa: E402
import synth  # noqa: E402
import eval as pm_eval  # noqa: E402
import patchmatchnet_amd as P  # noqa: E402
from patchmatchnet_amd import data_io  # noqa: E402
from patchmatchnet_amd.mvs import MVSDataset  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 12
extra = sys.argv[2:]
tmp = tempfile.mkdtemp(prefix="pmn_verify_", dir="/dev/shm")
data = os.path.join(tmp, "data")
synth.write_scene_scan(data, "scan1", V, 1200, 1600, n_src=10, seed=0, device="cuda")
open(os.path.join(data, "list.txt"), "w").write("scan1\n")
ckpt = os.path.join(GU.GOLDEN_DIR, "params_000007.npz")
out = os.path.join(tmp, "out")
so, sys.stdout = sys.stdout, open(os.devnull, "w")
try:
    pm_eval.main(["--input_folder", data, "--output_folder", out, "--checkpoint_path", ckpt, "--scan_list", os.path.join(data, "list.txt"),
                  "--num_views", "5", "--output_type", "depth", "--sample_seed", "9"] + extra)
finally:
    sys.stdout = so
_, params, kw = GU.load_case("default")
model = P.PatchmatchNet(**kw)
model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
model = model.cuda().eval()
ds = MVSDataset(data, num_views=5, scan_list=os.path.join(data, "list.txt"))
bad = []
with torch.no_grad():
    for i in range(len(ds)):
        s = ds[i]
        torch.manual_seed(9 + 1000003 * 0 + int(s["ref_view"]))  # eval.py's --sample_seed rule
        d, c, _ = model([torch.from_numpy(x)[None].cuda() for x in s["images"]], torch.from_numpy(s["intrinsics"])[None].cuda(),
                        torch.from_numpy(s["extrinsics"])[None].cuda(), torch.tensor([s["depth_min"]]).cuda(), torch.tensor([s["depth_max"]]).cuda())
        gd = data_io.read_map(os.path.join(out, "scan1", "depth_est", "{:0>8}.pfm".format(int(s["ref_view"]))))[..., 0]
        gc = data_io.read_map(os.path.join(out, "scan1", "confidence", "{:0>8}.pfm".format(int(s["ref_view"]))))[..., 0]
        wd, wc = d[0, 0].cpu().numpy(), c[0].cpu().numpy()
        if not (np.array_equal(gd, wd) and np.array_equal(gc, wc)):
            bad.append((int(s["ref_view"]), int((gd != wd).sum()), float((np.abs(gd - wd) / np.abs(wd)).max())))
print(f"eval.py {' '.join(extra) or '(default flags)'} on {V} views at 1600x1200 [GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'unset')}]:",
      f"{len(bad)} of {len(ds)} maps differ from the eager forward", bad[:4])
shutil.rmtree(tmp, ignore_errors=True)
