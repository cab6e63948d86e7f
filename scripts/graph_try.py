import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import patchmatchnet_amd as P
import bench as B
dev = torch.device("cuda", 0)
model = P.PatchmatchNet(**B.DEFAULT_KW); B.load_weights(model); model = model.to(dev).eval()
samples = B.make_samples(2, 6, 1200, 1600, dev, 0)
s = samples[0]
static = dict(images=[im.clone() for im in s["images"]], intrinsics=s["intrinsics"].clone(), extrinsics=s["extrinsics"].clone(),
              depth_min=s["depth_min"].clone(), depth_max=s["depth_max"].clone())
def fwd():
    return model(list(static["images"]), static["intrinsics"], static["extrinsics"], static["depth_min"], static["depth_max"])
with torch.no_grad():
    for _ in range(3): fwd()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): fwd()
    torch.cuda.current_stream().wait_stream(side)
    try:
        with torch.cuda.graph(g):
            out = fwd()
    except Exception as e:
        print("CAPTURE FAILED:", type(e).__name__, str(e)[:300]); sys.exit(0)
    torch.cuda.synchronize()
    eager = fwd(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20): fwd()
    torch.cuda.synchronize(); te = (time.perf_counter() - t) / 20
    t = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t) / 20
    print("eager %.3f ms  graph %.3f ms" % (te * 1e3, tg * 1e3))
    # correctness of replay with new inputs
    s2 = samples[1]
    for a, b in zip(static["images"], s2["images"]): a.copy_(b)
    g.replay(); torch.cuda.synchronize()
    d_graph = out[0].clone()
    torch.manual_seed(0)
    print("depth finite:", bool(torch.isfinite(d_graph).all()), float(d_graph.min()), float(d_graph.max()))
