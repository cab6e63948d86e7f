#!/usr/bin/env python
"""Whole forwards captured WITH their debug records (every stage's offsets, hypotheses, weights, scores, depths stay addressable),
three graphs replayed concurrently: which recorded tensor is the FIRST (in execution order) to differ from the eager forward's?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402
import patchmatchnet_amd as P  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
H, W, nv, S = int(os.environ.get("PMN_PROBE_H", 1200)), int(os.environ.get("PMN_PROBE_W", 1600)), 6, 3
dev = torch.device("cuda", 0)
model = P.PatchmatchNet(**bench.DEFAULT_KW)
bench.load_weights(model)
model = model.to(dev).eval()
samples = bench.make_samples(S, nv, H, W, dev, 0)
noise = [torch.rand(1, 48, H // 8, W // 8, generator=torch.Generator().manual_seed(50 + k)).to(dev) for k in range(S)]
ORDER = ("eval_offsets", "propa_offsets", "feature_weight", "depth_sample", "view_weights", "similarity", "score", "depth")


def run(k):
    s, dbg, rec = samples[k], {}, []
    feats = model.extract_features([im for im in s["images"]])
    for v, f in enumerate(feats):
        for st in (3, 2, 1):
            rec.append((f"feat_v{v}_s{st}", f[st]))
    depth, conf, _ = model([im for im in s["images"]], s["intrinsics"].clone(), s["extrinsics"], s["depth_min"], s["depth_max"],
                           noise=noise[k], debug=dbg, features=feats)
    for st in (3, 2, 1):
        for it, x in enumerate(dbg[st]):
            for key in ORDER:
                if key in x and torch.is_tensor(x[key]):
                    rec.append((f"s{st}_it{it + 1}_{key}", x[key]))
    rec += [("final_depth", depth), ("confidence", conf)]
    return rec


with torch.no_grad():
    want = [[(n, t.clone()) for n, t in run(k)] for k in range(S)]
    torch.cuda.synchronize()
    graphs, recs, streams = [], [], [torch.cuda.Stream(dev) for _ in range(S)]
    for k in range(S):
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            r = run(k)
        graphs.append(g)
        recs.append(r)
    torch.cuda.synchronize()
    first = {}
    for r in range(rounds):
        for k in range(S):
            with torch.cuda.stream(streams[k]):
                graphs[k].replay()
        torch.cuda.synchronize()
        for k in range(S):
            for (n, t), (_, w) in zip(recs[k], want[k]):
                if not torch.equal(t, w):
                    first.setdefault((k, n), []).append((r, int((t != w).sum())))
                    break
print(f"{S} whole forwards with debug records replayed concurrently x{rounds} at {W}x{H}: first tensor that differs:",
      {k: (len(v), v[:2]) for k, v in first.items()} if first else "none")
