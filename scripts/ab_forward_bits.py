#!/usr/bin/env python
"""Do two builds of libpmn_hip.so give the same BITS?  One process per build (a process loads one library):
    python scripts/ab_forward_bits.py --lib A.so --out /tmp/a.npz ; python scripts/ab_forward_bits.py --out /tmp/b.npz ; ... --compare /tmp/a.npz /tmp/b.npz
Dumps the final depth / confidence, every stage's depth maps, the view weights' arg-max and FeatureWeightNet's output of one bench
sample (seeded stage-3 draw, eager forward with the debug hooks)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--out", default=None)
ap.add_argument("--compare", nargs=2, default=None)
ap.add_argument("--width", type=int, default=1600)
ap.add_argument("--height", type=int, default=1200)
a = ap.parse_args()
if a.compare:
    x, y = np.load(a.compare[0]), np.load(a.compare[1])
    bad = 0
    for k in x.files:
        same = np.array_equal(x[k], y[k])
        bad += not same
        print(f"{k:28s} {'equal bits' if same else 'DIFFERENT: max |diff| %.3e, %d of %d elements' % (float(np.abs(x[k].astype(np.float64) - y[k].astype(np.float64)).max()), int((x[k] != y[k]).sum()), x[k].size)}")
    sys.exit(1 if bad else 0)
import torch
if a.lib:
    from patchmatchnet_amd import _lib
    _lib.LIB_PATH = os.path.abspath(a.lib)
import bench
import patchmatchnet_amd as P
from patchmatchnet_amd import ops
dev = torch.device("cuda", 0)
model = P.PatchmatchNet(**bench.DEFAULT_KW)
bench.load_weights(model)
model = model.to(dev).eval()
s = bench.make_samples(1, 6, a.height, a.width, dev, 0)[0]
out = {}
real_fw = ops.feature_weight
def fw_hook(*args, **kw):
    r = real_fw(*args, **kw)
    out[f"feature_weight_{len([k for k in out if k.startswith('feature_weight')])}"] = r.detach().cpu().numpy()
    return r
ops.feature_weight = fw_hook
torch.manual_seed(11)
with torch.no_grad():
    depth, conf, stages = model([im for im in s["images"]], s["intrinsics"].clone(), s["extrinsics"], s["depth_min"], s["depth_max"])
torch.cuda.synchronize()
out["depth"], out["confidence"] = depth.cpu().numpy(), conf.cpu().numpy()
for st, lst in stages.items():
    for i, d in enumerate(lst):
        out[f"stage{st}_depth{i}"] = d.cpu().numpy()
np.savez(a.out, **out)
print("wrote", a.out, sorted(out))
