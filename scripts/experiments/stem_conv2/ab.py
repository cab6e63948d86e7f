#!/usr/bin/env python
"""Candidate: FeatureNet conv0 + conv1 + conv2 in one kernel (stem_conv2.hip) against pmn_stem_f16s + pmn_conv2d_f16s (8 -> 16, 5x5 s2).

    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -shared -I include -I patchmatchnet_amd/csrc \
          -o scripts/experiments/stem_conv2/libstem_conv2.so scripts/experiments/stem_conv2/stem_conv2.hip patchmatchnet_amd/csrc/conv_f16s.hip
    python scripts/experiments/stem_conv2/ab.py

The candidate performs the product kernels' operations in the product kernels' order: outputs must be equal BIT FOR BIT."""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
from patchmatchnet_amd import params as PP  # noqa: E402

lib = ctypes.CDLL(os.path.join(HERE, "libstem_conv2.so"))
P, I = ctypes.c_void_p, ctypes.c_int
lib.stem_conv2.argtypes = [P] * 8 + [I] * 3 + [P]
lib.pmn_stem_f16s.argtypes = [P] * 6 + [I] * 3 + [P]
lib.pmn_conv2d_f16s.argtypes = [P] * 4 + [I] * 8 + [P]
for f in (lib.stem_conv2, lib.pmn_stem_f16s, lib.pmn_conv2d_f16s):
    f.restype = I
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(7)


def bn(c):
    return (0.5 + torch.rand(c, generator=gen), 0.1 * torch.randn(c, generator=gen), 0.1 * torch.randn(c, generator=gen),
            0.5 + torch.rand(c, generator=gen))


d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
w0, s0 = (d(a) for a in PP.pack_conv(0.3 * torch.randn(8, 3, 3, 3, generator=gen), bn=bn(8)))
w1a, s1 = (d(a) for a in PP.pack_stem_conv1_f16s(0.2 * torch.randn(8, 8, 3, 3, generator=gen), bn=bn(8)))
w2b, s2 = (d(a) for a in PP.pack_conv_f16s(0.1 * torch.randn(16, 8, 5, 5, generator=gen), bn=bn(16)))
st = torch.cuda.current_stream().cuda_stream


def product(img, mid, out):
    N, _, H, W = img.shape
    assert lib.pmn_stem_f16s(img.data_ptr(), w0.data_ptr(), s0.data_ptr(), w1a.data_ptr(), s1.data_ptr(), mid.data_ptr(), N, H, W, st) == 0
    assert lib.pmn_conv2d_f16s(mid.data_ptr(), w2b.data_ptr(), s2.data_ptr(), out.data_ptr(), N, H, W, 8, 16, 5, 2, 1, st) == 0


def fused(img, out):
    N, _, H, W = img.shape
    assert lib.stem_conv2(img.data_ptr(), w0.data_ptr(), s0.data_ptr(), w1a.data_ptr(), s1.data_ptr(), w2b.data_ptr(), s2.data_ptr(),
                          out.data_ptr(), N, H, W, st) == 0


for (N, H, W) in ((2, 37, 51), (1, 64, 96), (1, 50, 68), (2, 33, 40), (6, 1200, 1600)):
    img = torch.rand(N, 3, H, W, generator=gen).to(dev)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    mid = torch.empty(N, H, W, 8, device=dev)
    oa, ob = torch.full((N, Ho, Wo, 16), -1.0, device=dev), torch.full((N, Ho, Wo, 16), -2.0, device=dev)
    product(img, mid, oa)
    fused(img, ob)
    torch.cuda.synchronize()
    print("N%d %dx%d: equal bits %s, max |diff| %.3e (output scale %.2f)" % (N, W, H, bool(torch.equal(oa, ob)),
                                                                            float((oa - ob).abs().max()), float(oa.abs().max())), flush=True)
res = {"product": [], "fused": []}
for rep in range(4):
    for name in ("product", "fused"):
        run = (lambda: product(img, mid, oa)) if name == "product" else (lambda: fused(img, ob))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        res[name].append(1e3 * e0.elapsed_time(e1) / 20)
print("us per six 1600x1200 views: " + " | ".join(k + " " + " ".join("%.1f" % v for v in vs) for k, vs in res.items()), flush=True)
