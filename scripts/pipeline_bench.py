#!/usr/bin/env python
"""eval.py-style rate with the file decode taken out: pre-decoded samples in pinned host memory -> DevicePrefetcher (H2D one
sample ahead on a side stream) -> PatchmatchNet.forward -> MapWriter (D2H on a side stream + PFM / bin writes by a thread pool),
against the forward-only rate of bench.py on the same samples.  This is the part of the eval pipeline this repository owns; the
JPEG decode in front of it is CPU work that scales with DataLoader workers.

    python scripts/pipeline_bench.py [--samples 64] [--format .pfm]
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import eval as pm_eval  # noqa: E402
import patchmatchnet_amd as P  # noqa: E402
from patchmatchnet_amd.graph import GraphedForward  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--distinct", type=int, default=8)
    ap.add_argument("--format", default=".pfm")
    ap.add_argument("--outdirs", nargs="+", default=["/dev/shm", "/tmp"], help="parents of the output folder (tmpfs / disk)")
    ap.add_argument("--writer_threads", type=int, nargs="+", default=[4])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    H, W, N = 1200, 1600, 5
    model = P.PatchmatchNet(**bench.DEFAULT_KW)
    bench.load_weights(model)
    model = model.to(dev).eval()
    dev_samples = bench.make_samples(args.distinct, N + 1, H, W, dev, 0)
    host = [{"images": [im.cpu().pin_memory() for im in s["images"]], "intrinsics": s["intrinsics"].cpu().pin_memory(),
             "extrinsics": s["extrinsics"].cpu().pin_memory(), "depth_min": s["depth_min"].cpu().pin_memory(),
             "depth_max": s["depth_max"].cpu().pin_memory()} for s in dev_samples]

    host_u8 = [dict(h, images=[(im * 255).round().to(torch.uint8).pin_memory() for im in h["images"]]) for h in host]

    def forward(s):
        return model(list(s["images"]), s["intrinsics"], s["extrinsics"], s["depth_min"], s["depth_max"])

    graphed = GraphedForward(model)

    def forward_graph(s):
        d, c = graphed(list(s["images"]), s["intrinsics"], s["extrinsics"], s["depth_min"], s["depth_max"])
        return d, c, None

    # two samples in flight, as eval.py --in_flight 2: a HIP stream + a replay slot each
    main_stream = torch.cuda.current_stream(dev)
    streams2 = [torch.cuda.Stream(dev) for _ in range(2)]
    slots2 = [GraphedForward(model) for _ in range(2)]

    with torch.no_grad():
        for i in range(30):
            forward(dev_samples[i % args.distinct])
        torch.cuda.synchronize()
        t = time.time()
        for i in range(args.samples):
            forward(dev_samples[i % args.distinct])
        torch.cuda.synchronize()
        fwd = args.samples / (time.time() - t)

        out = {}
        runs = [("h2d+forward", 1, args.outdirs[0])] + [("h2d+forward+d2h+write", t, o) for o in args.outdirs
                                                        for t in args.writer_threads]
        runs = [(m, t, o, src, g) for g in ("eager", "graph", "graphx2") for src in ("float32", "uint8") for m, t, o in runs]
        for mode, threads, parent, src, how in runs:
            mode = how + " " + src + " " + mode
            fwd_fn = forward if how == "eager" else forward_graph
            with tempfile.TemporaryDirectory(dir=parent) as tmp:
                writer = pm_eval.MapWriter(dev, args.format, workers=threads)
                pool = host if src == "float32" else host_u8
                loader = (pool[i % args.distinct] for i in range(args.samples + 16))
                n, t = 0, None
                for s in pm_eval.DevicePrefetcher(loader, dev):
                    if n == 16:  # the first samples allocate the pinned buffers (hipHostMalloc, once per run) and start the thread pool
                        torch.cuda.synchronize()
                        writer.drain()
                        t = time.time()
                    st = main_stream
                    if how == "graphx2":
                        st = streams2[n % 2]
                        st.wait_stream(main_stream)
                        for tt in list(s["images"]) + [s["intrinsics"], s["extrinsics"], s["depth_min"], s["depth_max"]]:
                            tt.record_stream(st)
                    with torch.cuda.stream(st):
                        if how == "graphx2":
                            depth, conf = slots2[n % 2](list(s["images"]), s["intrinsics"], s["extrinsics"], s["depth_min"],
                                                        s["depth_max"])
                        else:
                            depth, conf, _ = fwd_fn(s)
                        if mode.endswith("write"):
                            writer.submit(torch.stack((depth[0, 0], conf[0]), 0),
                                          os.path.join(tmp, "depth_est", "%08d%s" % (n, args.format)),
                                          os.path.join(tmp, "confidence", "%08d%s" % (n, args.format)))
                    n += 1
                torch.cuda.synchronize()
                writer.close()
                out[mode] = max(out.get(mode, 0.0), args.samples / (time.time() - t))
                print("%s, %d writer threads, files under %s: %.1f samples/s" % (mode, threads, parent, args.samples / (time.time() - t)),
                      flush=True)
    res = {"forward_only_per_s": round(fwd, 1), **{k + "_per_s": round(v, 1) for k, v in out.items()},
           "ratio_full_pipeline_eager_float32_upload": round(out["eager float32 h2d+forward+d2h+write"] / fwd, 3),
           "ratio_full_pipeline_graph_uint8_upload": round(out["graph uint8 h2d+forward+d2h+write"] / fwd, 3),
           "ratio_full_pipeline_graphx2_uint8_upload": round(out["graphx2 uint8 h2d+forward+d2h+write"] / fwd, 3),
           "samples": args.samples,
           "h2d_MB_per_sample": {"float32": round((N + 1) * 3 * H * W * 4 / 1e6, 1), "uint8": round((N + 1) * 3 * H * W / 1e6, 1)},
           "d2h_MB_per_sample": round(2 * H * W * 4 / 1e6, 1),
           "format": args.format}
    print("PIPELINE " + json.dumps(res))


if __name__ == "__main__":
    main()
