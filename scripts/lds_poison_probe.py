#!/usr/bin/env python
"""Does any kernel of the forward read LDS it has not written?  An eager forward (debug hooks on) runs while a second stream keeps
launching workgroups that fill their LDS with a bit pattern (NaN, a huge float, fp16 NaNs); every recorded tensor is compared with the
undisturbed forward's.  A kernel that consumes stale LDS shows up as the first tensor that differs."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402
import patchmatchnet_amd as P  # noqa: E402

dev = torch.device("cuda", 0)
L = ctypes.CDLL(os.path.join(ROOT, "build", "pw", "liblds_poison.so"))
L.lds_poison.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
model = P.PatchmatchNet(**bench.DEFAULT_KW)
bench.load_weights(model)
model = model.to(dev).eval()
sink = torch.zeros(4, dtype=torch.int32, device=dev)
side = torch.cuda.Stream(dev)
for (H, W, nv) in ((96, 128, 3), (480, 640, 6)):
    s = bench.make_samples(1, nv, H, W, dev, 0)[0]
    noise = torch.rand(1, 48, H // 8, W // 8, generator=torch.Generator().manual_seed(7)).to(dev)

    def forward(pattern=None):
        dbg = {}
        with torch.no_grad():
            if pattern is not None:
                for _ in range(60):
                    L.lds_poison(4096, 64 * 1024, pattern, sink.data_ptr(), side.cuda_stream)
            feats = model.extract_features([im for im in s["images"]])
            depth, conf, dpm = model([im for im in s["images"]], s["intrinsics"].clone(), s["extrinsics"], s["depth_min"], s["depth_max"],
                                     noise=noise, debug=dbg, features=feats)
        torch.cuda.synchronize()
        rec = {}
        for v, f in enumerate(feats):
            for st in (3, 2, 1):
                rec[f"a_feat_v{v}_s{st}"] = f[st]
        for st in (3, 2, 1):
            for it, x in enumerate(dbg[st]):
                for k in ("eval_offsets", "propa_offsets", "depth_sample", "feature_weight", "view_weights", "similarity", "score", "depth"):
                    if k in x and x[k] is not None and torch.is_tensor(x[k]):
                        rec[f"b_s{st}_it{it + 1}_{k}"] = x[k]
        rec["c_depth"], rec["c_confidence"] = depth, conf
        return {k: v.clone() for k, v in rec.items()}

    clean = forward()
    for name, pattern in (("NaN", 0x7FC00000), ("3e38", 0x7F7FFFFF), ("fp16 NaNs", 0x7E007E00), ("ones", 0x3F800000)):
        hits = {}
        for rep in range(6):
            got = forward(pattern)
            for k in sorted(got):
                if not torch.equal(got[k], clean[k]):
                    hits.setdefault(k, 0)
                    hits[k] += 1
        print(f"{W}x{H} LDS pattern {name}: tensors that differ from the undisturbed forward (of 6 runs):", hits if hits else "none")
