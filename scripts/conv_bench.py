#!/usr/bin/env python
"""Times FeatureNet's conv layers at the cfg-2 shapes (6 views of 1600x1200): VALU kernel (pmn_conv2d) vs matrix-core
kernel (pmn_conv2d_mfma).  Prints us and TFLOP/s per layer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import patchmatchnet_amd as P
from patchmatchnet_amd import params as PP, ops

dev = "cuda:0"
layers = [(8, 16, 5, 2, 1200, 1600), (16, 16, 3, 1, 600, 800), (16, 32, 5, 2, 600, 800), (32, 32, 3, 1, 300, 400),
          (32, 64, 5, 2, 300, 400), (64, 64, 3, 1, 150, 200)]
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for cin, cout, K, s, H, W in layers:
    x = torch.randn(6, H, W, cin, device=dev)
    wt = 0.1 * torch.randn(cout, cin, K, K)
    w, sh = PP.pack_conv(wt)
    w, sh = torch.from_numpy(w).to(dev), torch.from_numpy(sh).to(dev)
    t_valu = timeit(lambda: ops.conv2d(x, w, sh, cout, K, s, K // 2, relu=True))
    Ho, Wo = H // s, W // s
    gf = 6 * Ho * Wo * K * K * cin * cout * 2 / 1e9
    line = f"{cin:3d}->{cout:3d} {K}x{K} s{s} @ {H}x{W}: valu {t_valu:7.1f} us {gf / t_valu * 1e3:6.1f} TF/s"
    if (cin, cout, K, s) in ops.MFMA_CONV_SHAPES:
        wm, sm = PP.pack_conv_mfma(wt)
        wm, sm = torch.from_numpy(wm).to(dev), torch.from_numpy(sm).to(dev)
        t_m = timeit(lambda: ops.conv2d_mfma(x, wm, sm, K, s, K // 2, relu=True))
        line += f" | mfma {t_m:7.1f} us {gf / t_m * 1e3:6.1f} TF/s"
    if K == 3 and s == 1 and cin == cout and cin in (16, 32, 64):
        ww, sw = PP.pack_conv_wino(wt)
        ww, sw = torch.from_numpy(ww).to(dev), torch.from_numpy(sw).to(dev)
        t_w = timeit(lambda: ops.conv3x3_wino(x, ww, sw, relu=True))
        line += f" | wino {t_w:7.1f} us {gf / t_w * 1e3:6.1f} TF/s(direct-equivalent)"
    if K == 5 and s == 2 and (cin, cout) in ((8, 16), (16, 32), (32, 64)):
        w5, s5 = PP.pack_conv5x5s2_wino(wt)
        w5, s5 = torch.from_numpy(w5).to(dev), torch.from_numpy(s5).to(dev)
        t_w = timeit(lambda: ops.conv5x5s2_wino(x, w5, s5, relu=True))
        line += f" | wino {t_w:7.1f} us {gf / t_w * 1e3:6.1f} TF/s(direct-equivalent)"
    print(line, flush=True)
