#!/usr/bin/env python
"""Numerics study (CPU) for running FeatureNet's MFMA convolutions on the fp16 matrix cores with SPLIT operands -- candidate to lift
the conv layers off the fp32 MFMA rate (157 TF/s = 1/16 of the fp16 rate on gfx950; no xf32).

  x = x_hi + x_lo / 2048 with x_hi = fp16(x), x_lo = fp16((x - x_hi) * 2048)      (22 significant bits, lo kept in fp16's normal range)
  conv(x, w) ~= sum[ x_hi * w_hi ]  +  ( sum[ x_hi * w_lo ] + sum[ x_lo * w_hi ] ) / 2048            (the lo * lo term, 2^-22, dropped)

Every product of two fp16 numbers is exact in fp32 (11 x 11 significant bits); the MFMA accumulates in fp32.  Emulated here with
float64 products of the fp16 values rounded into fp32 accumulators per k-block of 16 (the MFMA's K), for the checkpoint's own
layers on realistic activations; compared with fp32 direct and fp32 Winograd (what the kernels of rounds 1-2 compute) against
fp64.  Three fp16 MFMAs per k-step = 16 / 3 = 5.3x the fp32 MFMA rate."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.nn.functional as F
import goldenutil as GU
from patchmatchnet_amd.net import FeatureNet


def split16(t):
    hi = t.to(torch.float16)
    lo = ((t - hi.float()) * 2048.0).to(torch.float16)
    return hi, lo


def conv_split(x, w, stride, pad):
    """fp32 accumulation of the three fp16 product sums (products exact; accumulation order = taps outer, channels inner)."""
    xh, xl = split16(x)
    wh, wl = split16(w)
    # conv2d in float32 on the fp16 values: the products are exact in fp32, the fp32 accumulation stands in for the MFMA's
    main = F.conv2d(xh.float(), wh.float(), None, stride, pad)
    low = F.conv2d(xh.float(), wl.float(), None, stride, pad) + F.conv2d(xl.float(), wh.float(), None, stride, pad)
    return main + low * (1.0 / 2048.0)


g, params, kw = GU.load_case("default")
net = FeatureNet()
net.load_state_dict({k[len("feature."):]: torch.from_numpy(v) for k, v in params.items() if k.startswith("feature.")})
net.eval()
x = torch.cat([torch.from_numpy(g[f"image_{v}"]) for v in range(int(g["n_views"]))], 0)
with torch.no_grad():
    t = x
    for i in range(11):
        m = getattr(net, f"conv{i}")
        if i >= 2:
            s = m.bn.weight.double() / torch.sqrt(m.bn.running_var.double() + m.bn.eps)
            w = m.conv.weight.double() * s[:, None, None, None]
            st, pd = m.conv.stride, m.conv.padding
            ref = F.conv2d(t.double(), w, None, st, pd)
            d32 = F.conv2d(t, w.float(), None, st, pd).double()
            sp = conv_split(t, w.float(), st, pd).double()
            scale = ref.abs().max()
            print(f"conv{i}: in {tuple(t.shape)} |x| max {float(t.abs().max()):.3g} |w| max {float(w.abs().max()):.3g} min-nonzero |x| "
                  f"{float(t[t != 0].abs().min()):.2e}:  fp32 direct max {float((d32 - ref).abs().max() / scale):.2e}   "
                  f"fp16x2 split max {float((sp - ref).abs().max() / scale):.2e} rms {float(((sp - ref) ** 2).mean().sqrt() / scale):.2e}")
        t = m(t)
