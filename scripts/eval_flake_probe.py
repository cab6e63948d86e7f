#!/usr/bin/env python
"""Hunting an intermittent mismatch seen in tests/test_eval_gpu.py::test_eval_cli_end_to_end: eval.py (feature cache, graph replay,
two samples in flight; --output_type both / depth) against a direct eager forward per sample under the same seed, repeated.
Reports which run, view and map deviates from the direct forward, and by how much."""
import os
import shutil
import sys
import tempfile

ROOT = os.environ.get("PMN_PROBE_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
tmp = tempfile.mkdtemp(prefix="pmn_flake_")
data = os.path.join(tmp, "data")
synth.write_scan(data, "scan9", n_views=4, H=96, W=128, n_src=2)
open(os.path.join(data, "list.txt"), "w").write("scan9\n")
ckpt = os.path.join(GU.GOLDEN_DIR, "params_000007.npz")
_, params, kw = GU.load_case("default")
model = P.PatchmatchNet(**kw)
model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
model = model.cuda().eval()
ds = MVSDataset(data, num_views=2, scan_list=os.path.join(data, "list.txt"))
torch.manual_seed(3)
want = {}
with torch.no_grad():
    for i in range(len(ds)):
        s = ds[i]
        d, c, _ = model([torch.from_numpy(x)[None].cuda() for x in s["images"]], torch.from_numpy(s["intrinsics"])[None].cuda(),
                        torch.from_numpy(s["extrinsics"])[None].cuda(), torch.tensor([s["depth_min"]]).cuda(), torch.tensor([s["depth_max"]]).cuda())
        want[int(s["ref_view"])] = (d[0, 0].cpu().numpy(), c[0].cpu().numpy())
FUSE = ["--geo_mask_thres", "1", "--photo_thres", "0.1"]
NOCACHE = ["--output_type", "depth", "--feature_cache", "0"]
firsts = {"both_async": ["--output_type", "both"] + FUSE, "both_inline": ["--output_type", "both", "--fuse_async", "0"] + FUSE,
          "depth_cache": ["--output_type", "depth"], "nothing": None}
PLAIN = {"plain": [], "plain_inflight1": ["--in_flight", "1"], "plain_eager": ["--hip_graph", "0"], "plain_inflight3": ["--in_flight", "3"]}
if len(sys.argv) > 2 and sys.argv[2] == "plain":  # which ingredient of the plain path itself: [variant] repeated, nothing before it
    firsts = {k: None for k in PLAIN}
elif len(sys.argv) > 2:
    firsts = {k: v for k, v in firsts.items() if k in sys.argv[2:]}
devnull = open(os.devnull, "w")


def run(out, extra):
    shutil.rmtree(out, ignore_errors=True)
    torch.manual_seed(3)
    so = sys.stdout
    sys.stdout = devnull
    try:
        pm_eval.main(["--input_folder", data, "--output_folder", out, "--checkpoint_path", ckpt, "--scan_list",
                      os.path.join(data, "list.txt"), "--num_views", "2", "--num_workers", "0"] + extra)
    finally:
        sys.stdout = so


bad = {}
for first, extra in firsts.items():  # the sequence [first, plain path] repeated: what does `first` leave behind for the plain-path run?
    for t in range(trials):
        if extra is not None:
            run(os.path.join(tmp, "out_first"), extra)
        out = os.path.join(tmp, "out_plain")
        run(out, NOCACHE + PLAIN.get(first, []))
        for v in range(4):
            for k, kind in enumerate(("depth_est", "confidence")):
                got = data_io.read_map(os.path.join(out, "scan9", kind, "{:0>8}.pfm".format(v)))[..., 0]
                if not np.array_equal(got, want[v][k]):
                    bad.setdefault((first, v, kind), []).append((t, int((got != want[v][k]).sum()), float(np.abs(got - want[v][k]).max())))
print("trials", trials, "deviations from the direct forward:", bad if bad else "none")
shutil.rmtree(tmp, ignore_errors=True)
