#!/usr/bin/env python
"""Numerics study for a Winograd F(2x2,3x3) version of FeatureNet's 3x3 stride-1 layers (candidate for the next round: 2.25x
fewer matrix-core MACs on conv3,4,6,7,9,10 = 80 of the 156 GFLOP of a six-view FeatureNet).  Runs on CPU: fp32 Winograd with
fp64-transformed filters vs fp32 direct convolution, both against fp64, on the checkpoint's own layers and realistic
activations (the golden case's images pushed through the preceding layers)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.nn.functional as F
import goldenutil as GU
from patchmatchnet_amd.net import FeatureNet

G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)

def winograd_conv(x, w, dtype):
    """x [N,C,H,W] (H, W even), w [K,C,3,3]; padding 1; arithmetic in `dtype` (filters transformed in fp64 first)."""
    N, C, H, W = x.shape
    U = (G @ w.double() @ G.t()).to(dtype)                      # [K,C,4,4]
    xp = F.pad(x.to(dtype), (1, 1, 1, 1))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                   # [N,C,H/2,W/2,4,4]
    V = Bt.to(dtype) @ tiles @ Bt.t().to(dtype)                  # input transform
    M = torch.einsum("kcab,nchwab->nkhwab", U, V)               # 16 GEMMs
    Y = At.to(dtype) @ M @ At.t().to(dtype)                      # [N,K,H/2,W/2,2,2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, -1, H, W)

g, params, kw = GU.load_case("default")
net = FeatureNet()
net.load_state_dict({k[len("feature."):]: torch.from_numpy(v) for k, v in params.items() if k.startswith("feature.")})
net.eval()
x = torch.cat([torch.from_numpy(g[f"image_{v}"]) for v in range(int(g["n_views"]))], 0)
acts = {}
with torch.no_grad():
    t = x
    for i in range(11):
        acts[i] = t
        t = getattr(net, f"conv{i}")(t)
    for i in (3, 4, 6, 7, 9, 10):
        m = getattr(net, f"conv{i}")
        s = m.bn.weight.double() / torch.sqrt(m.bn.running_var.double() + m.bn.eps)
        w = m.conv.weight.double() * s[:, None, None, None]
        xin = acts[i]
        ref = F.conv2d(xin.double(), w, None, 1, 1)
        d32 = F.conv2d(xin, w.float(), None, 1, 1).double()
        w32 = winograd_conv(xin, w, torch.float32).double()
        scale = ref.abs().max()
        print(f"conv{i}: {tuple(xin.shape)} -> direct fp32 max err {float((d32 - ref).abs().max() / scale):.2e}   "
              f"winograd fp32 max err {float((w32 - ref).abs().max() / scale):.2e}   rms {float(((w32 - ref) ** 2).mean().sqrt() / scale):.2e}")

# ---- 5x5 stride-2 layers (conv2, conv5, conv8): phase decomposition into four stride-1 sub-convolutions (3x3, 3x2, 2x3, 2x2 taps)
# each in Winograd form F(2, 3) / F(2, 2) per dimension: 49 instead of 100 multiplies per 2x2 output tile and input channel.
G3 = torch.tensor([[1, 0], [1, 1], [0, 1]], dtype=torch.float64)
Bt3 = torch.tensor([[1, -1, 0], [0, 1, 0], [0, 1, -1]], dtype=torch.float64)
At3 = torch.tensor([[1, 1, 0], [0, 1, -1]], dtype=torch.float64)
GS, BS, AS = {0: G, 1: G3}, {0: Bt, 1: Bt3}, {0: At, 1: At3}

def winograd_conv5s2(x, w, dtype):
    """x [N,C,H,W] with H, W multiples of 4; w [K,C,5,5]; padding 2, stride 2."""
    N, C, H, W = x.shape
    xp = F.pad(x.to(dtype), (2, 2 + 4, 2, 2 + 4))
    out = 0
    for r in (0, 1):
        for s in (0, 1):
            sub = w.double()[:, :, r::2, s::2]
            U = (GS[r] @ sub @ GS[s].t()).to(dtype)                            # [K,C,nr,ns]
            X = xp[:, :, r::2, s::2]                                            # phase image
            nr, ns = BS[r].shape[0], BS[s].shape[0]
            tiles = X.unfold(2, nr, 2).unfold(3, ns, 2)[:, :, :H // 4, :W // 4]  # [N,C,th,tw,nr,ns]
            V = BS[r].to(dtype) @ tiles @ BS[s].t().to(dtype)
            M = torch.einsum("kcab,nchwab->nkhwab", U, V)
            out = out + AS[r].to(dtype) @ M @ AS[s].t().to(dtype)              # [N,K,th,tw,2,2]
    return out.permute(0, 1, 2, 4, 3, 5).reshape(N, -1, H // 2, W // 2)

with torch.no_grad():
    for i in (2, 5, 8):
        m = getattr(net, f"conv{i}")
        s = m.bn.weight.double() / torch.sqrt(m.bn.running_var.double() + m.bn.eps)
        w = m.conv.weight.double() * s[:, None, None, None]
        xin = acts[i]
        ref = F.conv2d(xin.double(), w, None, 2, 2)
        d32 = F.conv2d(xin, w.float(), None, 2, 2).double()
        w32 = winograd_conv5s2(xin, w, torch.float32).double()
        w64 = winograd_conv5s2(xin, w, torch.float64)
        scale = ref.abs().max()
        print(f"conv{i} (5x5 s2): {tuple(xin.shape)} -> direct fp32 max err {float((d32 - ref).abs().max() / scale):.2e}   "
              f"winograd fp32 max err {float((w32 - ref).abs().max() / scale):.2e}   (fp64 identity check {float((w64 - ref).abs().max() / scale):.1e})")

# ---- F(4x4,3x3) for the 3x3 layers: 36 multiplies per 16 outputs (2.25 per output vs 4 for F(2x2,3x3)); fp32 error check
G6 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
Bt6 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
At6 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)

def winograd43(x, w, dtype):
    N, C, H, W = x.shape
    U = (G6 @ w.double() @ G6.t()).to(dtype)
    xp = F.pad(x.to(dtype), (1, 1 + 4, 1, 1 + 4))
    tiles = xp.unfold(2, 6, 4).unfold(3, 6, 4)[:, :, :(H + 3) // 4, :(W + 3) // 4]
    V = Bt6.to(dtype) @ tiles @ Bt6.t().to(dtype)
    M = torch.einsum("kcab,nchwab->nkhwab", U, V)
    Y = At6.to(dtype) @ M @ At6.t().to(dtype)
    th, tw = Y.shape[2], Y.shape[3]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, -1, th * 4, tw * 4)[:, :, :H, :W]

with torch.no_grad():
    for i in (6, 7, 9, 10):
        m = getattr(net, f"conv{i}")
        s = m.bn.weight.double() / torch.sqrt(m.bn.running_var.double() + m.bn.eps)
        w = m.conv.weight.double() * s[:, None, None, None]
        xin = acts[i]
        ref = F.conv2d(xin.double(), w, None, 1, 1)
        w32 = winograd43(xin, w, torch.float32).double()
        w64 = winograd43(xin, w, torch.float64)
        scale = ref.abs().max()
        print(f"conv{i} F(4x4,3x3): fp32 max err {float((w32 - ref).abs().max() / scale):.2e}   (fp64 identity check {float((w64 - ref).abs().max() / scale):.1e})")
