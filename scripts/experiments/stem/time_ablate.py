#!/usr/bin/env python
"""Times the stem kernel with one phase dropped at a time (libstem_abl.so, built from stem_ablate.hip):
bit 0 no global loads, 1 no conv0 FMA loop, 2 no MFMA loop, 3 no stores, 4 no second conv0 pass.  1600x1200, one view per launch
(as the product launches it) and six per launch."""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, sys.argv[1] if len(sys.argv) > 1 else "libstem_abl.so"))
lib.stem_abl.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
lib.stem_abl.restype = ctypes.c_int
masks = [int(m) for m in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 4, 8, 16, 6, 9, 15, 31, 0]
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
w0 = (torch.randn(3, 3, 3, 8, generator=g) * 0.2).to(dev)
s0 = (torch.randn(8, generator=g) * 0.1).to(dev)
w1a = (torch.randn(3, 2, 64, 8, generator=g) * 0.1).half().to(dev)
s1 = (torch.randn(8, generator=g) * 0.1).to(dev)
H, W = 1200, 1600
for N in (1, 6):
    img = torch.rand(N, 3, H, W, generator=g).to(dev)
    out = torch.empty(N, H, W, 8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for var in [m_ for m_ in (64, 128) if m_ in masks]:  # variant 2 against the product kernel's text, incl. an image whose tiles are cut by the border
        for (h2, w2) in ((H, W), (1187, 1596), (50, 68)):
            im2 = torch.rand(N, 3, h2, w2, generator=g).to(dev)
            a, b = torch.full((N, h2, w2, 8), -1.0, device=dev), torch.full((N, h2, w2, 8), -2.0, device=dev)
            lib.stem_abl(im2.data_ptr(), w0.data_ptr(), s0.data_ptr(), w1a.data_ptr(), s1.data_ptr(), a.data_ptr(), N, h2, w2, 0, st)
            lib.stem_abl(im2.data_ptr(), w0.data_ptr(), s0.data_ptr(), w1a.data_ptr(), s1.data_ptr(), b.data_ptr(), N, h2, w2, var, st)
            torch.cuda.synchronize()
            print("N=%d %dx%d variant %d vs product text: max |diff| %.3e, equal bits: %s" % (
                N, w2, h2, var, float((a - b).abs().max()), bool(torch.equal(a, b))), flush=True)
    for abl in masks:
        for _ in range(5):
            lib.stem_abl(img.data_ptr(), w0.data_ptr(), s0.data_ptr(), w1a.data_ptr(), s1.data_ptr(), out.data_ptr(), N, H, W, abl, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 40
        e0.record()
        for _ in range(reps):
            lib.stem_abl(img.data_ptr(), w0.data_ptr(), s0.data_ptr(), w1a.data_ptr(), s1.data_ptr(), out.data_ptr(), N, H, W, abl, st)
        e1.record()
        torch.cuda.synchronize()
        print("N=%d abl=%2d  %.1f us per launch  (%.1f us per view)" % (N, abl, 1e3 * e0.elapsed_time(e1) / reps,
                                                                      1e3 * e0.elapsed_time(e1) / reps / N), flush=True)
