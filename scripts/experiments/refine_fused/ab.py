#!/usr/bin/env python
"""Candidate: Refinement's full-resolution half in one kernel (refine_fused.hip) against pmn_refine_front + pmn_refine_tail.

    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -shared -I include -I patchmatchnet_amd/csrc \
          -o scripts/experiments/refine_fused/librefine_fused.so scripts/experiments/refine_fused/refine_fused.hip patchmatchnet_amd/csrc/refine.hip
    python scripts/experiments/refine_fused/ab.py            # on a GPU box: parity of both against float64 torch, then timing
    python scripts/experiments/refine_fused/ab.py --cpu      # here: the operand packing against a numpy emulation of the MFMA sum

Without a GPU only the packing / k-block convention of conv3's A operands is checked (numpy emulation of exactly what the kernel
sums: hi*hi + (hi*lo + lo*hi)/2048 over blocks q = 4 ks + kb = (tap, channel half))."""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
from patchmatchnet_amd import params as PP  # noqa: E402


def pack_refine_conv3_f16s(weight, bn, eps=PP.BN_EPS):
    """Refinement.conv3 (16 -> 8, 3x3 + BatchNorm) as the A operands (rows = output channels) of v_mfma_f32_16x16x32_f16 ->
    (float16 [5 k-steps][2 (hi|lo)][64 lanes][8], float32 shift [8]).  Lane l = 16 kb + row (rows 8..15 zero); k-block q = 4 ks + kb =
    2 tap + cb (tap = dy * 3 + dx, cb = half of the 16 input channels; blocks 18, 19 zero); the 8 values = input channels 8 cb .. 8 cb + 7."""
    w = PP._np64(weight)
    assert w.shape == (8, 16, 3, 3)
    g, b, m, v = (PP._np64(t) for t in bn)
    sc = g / np.sqrt(v + eps)
    w = w * sc[:, None, None, None]
    shift = b - m * sc
    full = np.zeros((5, 4, 16, 8), np.float64)
    for q in range(18):
        tap, cb = q >> 1, q & 1
        dy, dx = divmod(tap, 3)
        full[q // 4, q % 4, :8, :] = w[:, 8 * cb:8 * cb + 8, dy, dx]
    hi, lo = PP.split_f16(full.reshape(5, 64, 8))
    return np.ascontiguousarray(np.stack((hi, lo), axis=1)), np.ascontiguousarray(shift.astype(np.float32))


def make_weights(gen):
    def bn(c):
        return (0.5 + torch.rand(c, generator=gen), 0.1 * torch.randn(c, generator=gen), 0.1 * torch.randn(c, generator=gen),
                0.5 + torch.rand(c, generator=gen))
    return {"conv0": 0.3 * torch.randn(8, 3, 3, 3, generator=gen), "bn0": bn(8),
            "deconv": 0.2 * torch.randn(8, 8, 3, 3, generator=gen), "bnd": bn(8),
            "conv3": 0.15 * torch.randn(8, 16, 3, 3, generator=gen), "bn3": bn(8),
            "res": 0.2 * torch.randn(1, 8, 3, 3, generator=gen)}


def bn_apply(t, bn):
    g, b, m, v = (x.double() for x in bn)
    return torch.nn.functional.batch_norm(t, m, v, g, b, False, 0.0, PP.BN_EPS)


def reference64(wts, img, t2_nchw, dnorm, dmin, dmax):
    """net.py:110-122 from conv0 / deconv on, in float64."""
    F = torch.nn.functional
    f = torch.relu(bn_apply(F.conv2d(img.double(), wts["conv0"].double(), None, 1, 1), wts["bn0"]))
    up = torch.relu(bn_apply(F.conv_transpose2d(t2_nchw.double(), wts["deconv"].double(), None, 2, 1, 1), wts["bnd"]))
    x16 = torch.cat((up, f), 1)
    c3 = torch.relu(bn_apply(F.conv2d(x16, wts["conv3"].double(), None, 1, 1), wts["bn3"]))
    res = F.conv2d(c3, wts["res"].double(), None, 1, 1)
    d = F.interpolate(dnorm.double(), scale_factor=2, mode="nearest") + res
    lo, hi = dmin.double().view(-1, 1, 1, 1), dmax.double().view(-1, 1, 1, 1)
    return d * (hi - lo) + lo, x16, c3


def cpu_check():
    gen = torch.Generator().manual_seed(3)
    wts = make_weights(gen)
    B, H, W = 1, 12, 16
    img, t2 = torch.rand(B, 3, H, W, generator=gen), torch.rand(B, 8, H // 2, W // 2, generator=gen)
    dnorm, dmin, dmax = torch.rand(B, 1, H // 2, W // 2, generator=gen), torch.tensor([400.0]), torch.tensor([900.0])
    _, x16, c3 = reference64(wts, img, t2, dnorm, dmin, dmax)
    w3a, s3 = pack_refine_conv3_f16s(wts["conv3"], wts["bn3"])
    A = w3a.astype(np.float64)  # [5][2][64][8]
    xh, xl = PP.split_f16(x16[0].numpy())  # [16,H,W]
    xh, xl = np.pad(xh.astype(np.float64), ((0, 0), (1, 1), (1, 1))), np.pad(xl.astype(np.float64), ((0, 0), (1, 1), (1, 1)))
    worst = 0.0
    for (y, x) in [(0, 0), (3, 5), (H - 1, W - 1), (6, 0), (H - 1, 7)]:
        main, low = np.zeros(8), np.zeros(8)
        for ks in range(5):
            for kb in range(4):
                q = min(4 * ks + kb, 17)  # the kernel's clamp for the two padding blocks (their weights are zero)
                tap, cb = q >> 1, q & 1
                dy, dx = divmod(tap, 3)
                bh, bl = xh[8 * cb:8 * cb + 8, y + dy, x + dx], xl[8 * cb:8 * cb + 8, y + dy, x + dx]
                for row in range(8):
                    ah, al = A[ks, 0, 16 * kb + row], A[ks, 1, 16 * kb + row]
                    main[row] += float(ah @ bh)
                    low[row] += float(ah @ bl) + float(al @ bh)
        got = np.maximum(main + low / 2048.0 + s3.astype(np.float64), 0.0)
        want = c3[0, :, y, x].numpy()
        worst = max(worst, float(np.abs(got - want).max() / max(np.abs(c3).max().item(), 1e-9)))
    assert (A[:, :, [16 * kb + r for kb in range(4) for r in range(8, 16)]] == 0).all(), "rows 8..15 must be zero"
    print("cpu check: conv3 through the packed A operands (k-block convention of the kernel) vs float64: rel err %.2e" % worst)
    assert worst < 2e-6


def gpu_run():
    lib = ctypes.CDLL(os.path.join(HERE, "librefine_fused.so"))
    P, I = ctypes.c_void_p, ctypes.c_int
    lib.refine_fused.argtypes = [P] * 13 + [I] * 3 + [P]
    lib.pmn_refine_front.argtypes = [P] * 7 + [I] * 3 + [P]
    lib.pmn_refine_tail.argtypes = [P] * 8 + [I] * 3 + [P]
    for f in (lib.refine_fused, lib.pmn_refine_front, lib.pmn_refine_tail):
        f.restype = I
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(5)
    wts = make_weights(gen)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    w0, s0 = (d(a) for a in PP.pack_conv(wts["conv0"], bn=wts["bn0"]))
    wd, sd = (d(a) for a in PP.pack_deconv(wts["deconv"], bn=wts["bnd"]))
    w3, s3, wr = (d(a) for a in PP.pack_refine_tail(wts["conv3"], wts["bn3"], wts["res"]))
    w3a, s3a = (d(a) for a in pack_refine_conv3_f16s(wts["conv3"], wts["bn3"]))
    st = torch.cuda.current_stream().cuda_stream

    def product(img, t2, dnorm, dmin, dmax, x16, out):
        B, _, H, W = img.shape
        assert lib.pmn_refine_front(img.data_ptr(), t2.data_ptr(), w0.data_ptr(), s0.data_ptr(), wd.data_ptr(), sd.data_ptr(),
                                    x16.data_ptr(), B, H, W, st) == 0
        assert lib.pmn_refine_tail(x16.data_ptr(), w3.data_ptr(), s3.data_ptr(), wr.data_ptr(), dnorm.data_ptr(), dmin.data_ptr(),
                                   dmax.data_ptr(), out.data_ptr(), B, H, W, st) == 0

    def fused(img, t2, dnorm, dmin, dmax, out):
        B, _, H, W = img.shape
        assert lib.refine_fused(img.data_ptr(), t2.data_ptr(), w0.data_ptr(), s0.data_ptr(), wd.data_ptr(), sd.data_ptr(),
                                w3a.data_ptr(), s3a.data_ptr(), wr.data_ptr(), dnorm.data_ptr(), dmin.data_ptr(), dmax.data_ptr(),
                                out.data_ptr(), B, H, W, st) == 0

    for (B, H, W) in ((2, 38, 52), (1, 64, 80), (1, 50, 70), (1, 1200, 1600)):
        img = torch.rand(B, 3, H, W, generator=gen)
        t2n = torch.rand(B, 8, H // 2, W // 2, generator=gen)
        dnorm = torch.rand(B, 1, H // 2, W // 2, generator=gen)
        dmin, dmax = torch.full((B,), 425.0), torch.full((B,), 935.0)
        ref = reference64(wts, img, t2n, dnorm, dmin, dmax)[0] if H < 200 else None
        gi, gt = img.to(dev), t2n.permute(0, 2, 3, 1).contiguous().to(dev)
        gd, gmin, gmax = dnorm.to(dev), dmin.to(dev), dmax.to(dev)
        x16 = torch.empty(B, H, W, 16, device=dev)
        oa, ob = torch.full((B, 1, H, W), -1.0, device=dev), torch.full((B, 1, H, W), -2.0, device=dev)
        product(gi, gt, gd, gmin, gmax, x16, oa)
        fused(gi, gt, gd, gmin, gmax, ob)
        torch.cuda.synchronize()
        span = 510.0
        msg = "B%d %dx%d: fused vs product max |diff| / span %.2e" % (B, W, H, float((oa - ob).abs().max()) / span)
        if ref is not None:
            msg += "; vs float64: product %.2e, fused %.2e" % (float((oa.double().cpu() - ref).abs().max()) / span,
                                                                float((ob.double().cpu() - ref).abs().max()) / span)
        print(msg, flush=True)
    # timing at 1600x1200, B = 1 (the bench workload), alternating
    res = {"product": [], "fused": []}
    for rep in range(4):
        for name in ("product", "fused"):
            run = (lambda: product(gi, gt, gd, gmin, gmax, x16, oa)) if name == "product" else (lambda: fused(gi, gt, gd, gmin, gmax, ob))
            for _ in range(5):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                run()
            e1.record()
            torch.cuda.synchronize()
            res[name].append(1e3 * e0.elapsed_time(e1) / 30)
    print("us per 1600x1200 depth map: " + " | ".join(k + " " + " ".join("%.1f" % v for v in vs) for k, vs in res.items()), flush=True)


if __name__ == "__main__":
    cpu_check()
    if "--cpu" not in sys.argv and torch.cuda.is_available():
        gpu_run()
