#!/usr/bin/env python
"""A/B of two builds of conv_f16s.hip (libf16s_old.so = the committed kernel text, libf16s_new.so = the working tree; build lines in
scripts/experiments/README.md): pmn_conv2d_f16s on FeatureNet's six layer shapes at the bench sizes (six 1600x1200 views), bits
compared, then 30 launches each, alternating."""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
from patchmatchnet_amd import params as PP  # noqa: E402

libs = {}
TAGS = sys.argv[1].split(",") if len(sys.argv) > 1 else ["old", "new"]
ONLY = int(sys.argv[2]) if len(sys.argv) > 2 else -1  # index of the one shape to run
for tag in TAGS:
    lib = ctypes.CDLL(os.path.join(HERE, "libf16s_%s.so" % tag))
    lib.pmn_conv2d_f16s.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 8 + [ctypes.c_void_p]
    lib.pmn_conv2d_f16s.restype = ctypes.c_int
    libs[tag] = lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
shapes = [(5, 2, 8, 16, 1200, 1600), (3, 1, 16, 16, 600, 800), (5, 2, 16, 32, 600, 800), (3, 1, 32, 32, 300, 400),
          (5, 2, 32, 64, 300, 400), (3, 1, 64, 64, 150, 200)]


def run(lib, x, w, sh, out, k, s, cin, cout):
    N, H, W, _ = x.shape
    rc = lib.pmn_conv2d_f16s(x.data_ptr(), w.data_ptr(), sh.data_ptr(), out.data_ptr(), N, H, W, cin, cout, k, s, 1, st)
    assert rc == 0, rc


for si, (k, s, cin, cout, H, W) in enumerate(shapes):
    if ONLY >= 0 and si != ONLY:
        continue
    wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
    wp, shp = PP.pack_conv_f16s(wt, bias=0.1 * torch.randn(cout, generator=g))
    w, sh = torch.from_numpy(wp).to(dev), torch.from_numpy(shp).to(dev)
    for (N, h, w_) in ((6, H, W), (2, 37, 51)):
        x = torch.randn(N, h, w_, cin, generator=g).to(dev)
        Ho, Wo = (h - 1) // s + 1, (w_ - 1) // s + 1
        outs = {t: torch.full((N, Ho, Wo, cout), float(i), device=dev) for i, t in enumerate(libs)}
        for t in libs:
            run(libs[t], x, w, sh, outs[t], k, s, cin, cout)
        torch.cuda.synchronize()
        same = all(torch.equal(outs[TAGS[0]], outs[t]) for t in TAGS)
        print("k%d s%d %d->%d  %dx%dx%d: equal bits %s (max |diff| %.2e)" % (k, s, cin, cout, N, h, w_, same,
                                                                            max(float((outs[TAGS[0]] - outs[t]).abs().max()) for t in TAGS)), flush=True)
    x = torch.randn(6, H, W, cin, generator=g).to(dev)
    out = torch.empty(6, (H - 1) // s + 1, (W - 1) // s + 1, cout, device=dev)
    res = {t: [] for t in TAGS}
    for rep in range(5 if ONLY >= 0 else 3):
        for t in TAGS:
            for _ in range(5):
                run(libs[t], x, w, sh, out, k, s, cin, cout)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                run(libs[t], x, w, sh, out, k, s, cin, cout)
            e1.record()
            torch.cuda.synchronize()
            res[t].append(1e3 * e0.elapsed_time(e1) / 30)
    print("    us per launch (six views): " + " | ".join(t + " " + " ".join("%.1f" % v for v in res[t]) for t in TAGS), flush=True)
