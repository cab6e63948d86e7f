"""CPU emulation of pmn_conv2d_f16s (csrc/conv_f16s.hip): the lane layouts of v_mfma_f32_16x16x32_f16, the k-block / tap / chunk
indexing, the LDS patch addressing and the host-side weight packing (params.pack_conv_f16s) restated in numpy and checked against
torch's float64 convolution -- the kernel is written against this emulation (there is no GPU in the authoring container), and the
GPU parity tests (tests/test_hip_parity.py) then check the real thing.  Also pins the NUMERICS of the split-operand scheme: three
fp16 products per fp32 product (hi*hi, hi*lo, lo*hi; fp32 accumulation) reproduce an fp32 convolution to ~3e-7 of the output scale
on FeatureNet's own layers (reference models/net.py:17-37; scripts/fp16_split_study.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from patchmatchnet_amd import params

LAYERS = [(5, 2, 8, 16), (3, 1, 16, 16), (5, 2, 16, 32), (3, 1, 32, 32), (5, 2, 32, 64), (3, 1, 64, 64)]  # (K, stride, cin, cout)


def emulate(x_nhwc, wpk, shift, K, S, cin, cout, relu=True, dil=1, CC=None):
    """The kernel, wave by wave: x [N,H,W,cin] float32 -> [N,Ho,Wo,cout] float32 (cout = the padded channel count)."""
    N, H, W, _ = x_nhwc.shape
    pad = dil * (K // 2)
    Ho, Wo = (H - 1) // S + 1, (W - 1) // S + 1
    CC = CC or params.f16s_chunk(cin, K)
    ncb, chunks, NT = CC // 8, cin // CC, cout // 16
    MT = 4 if S == 1 else 2
    TH, TW = 4 * MT, 16
    PH, PW = (TH - 1) * S + (K - 1) * dil + 1, (TW - 1) * S + (K - 1) * dil + 1
    nq = K * K * ncb
    ksteps = (nq + 3) // 4
    assert wpk.shape == (chunks, ksteps, NT, 2, 64, 8)
    out = np.zeros((N, Ho, Wo, cout), np.float32)
    lane = np.arange(64)
    li, kb = lane % 16, lane // 16
    for n in range(N):
        for oy0 in range(0, Ho, TH):
            for ox0 in range(0, Wo, TW):
                iy0, ix0 = oy0 * S - pad, ox0 * S - pad
                acc_main = np.zeros((4, MT, NT, 16, 16), np.float64)  # [wave][t][nt][row i][col n]
                acc_low = np.zeros((4, MT, NT, 16, 16), np.float64)
                for ch in range(chunks):
                    # patch staging: zero outside the image, split into hi / lo planes [PH][PW][CC]
                    patch = np.zeros((PH, PW, CC), np.float32)
                    for py in range(PH):
                        for px in range(PW):
                            gy, gx = iy0 + py, ix0 + px
                            if 0 <= gy < H and 0 <= gx < W:
                                patch[py, px] = x_nhwc[n, gy, gx, ch * CC:(ch + 1) * CC]
                    phi, plo = params.split_f16(patch)
                    for wv in range(4):
                        for ks in range(ksteps):
                            q = np.minimum(4 * ks + kb, nq - 1)  # padding blocks read a valid address (their weights are 0)
                            tap, cb = q // ncb, q % ncb
                            dy, dx = tap // K, tap % K
                            for t in range(MT):
                                row = (wv * MT + t) * S + dy * dil
                                col = li * S + dx * dil
                                a_hi = phi[row, col][np.arange(64)[:, None], (cb * 8)[:, None] + np.arange(8)[None]].astype(np.float64)
                                a_lo = plo[row, col][np.arange(64)[:, None], (cb * 8)[:, None] + np.arange(8)[None]].astype(np.float64)
                                A_hi = np.zeros((16, 32)); A_lo = np.zeros((16, 32))
                                for e in range(8):
                                    A_hi[li, 8 * kb + e] = a_hi[:, e]
                                    A_lo[li, 8 * kb + e] = a_lo[:, e]
                                for nt in range(NT):
                                    b_hi = wpk[ch, ks, nt, 0].astype(np.float64)  # [lane][8]: lane = 16 kb + n
                                    b_lo = wpk[ch, ks, nt, 1].astype(np.float64)
                                    B_hi = np.zeros((32, 16)); B_lo = np.zeros((32, 16))
                                    for e in range(8):
                                        B_hi[8 * kb + e, li] = b_hi[:, e]
                                        B_lo[8 * kb + e, li] = b_lo[:, e]
                                    acc_main[wv, t, nt] += A_hi @ B_hi
                                    acc_low[wv, t, nt] += A_hi @ B_lo + A_lo @ B_hi
                # epilogue: lane l holds D[row = 4 kb + r][col = n]: pixel ox0 + row, channel 16 nt + n
                for wv in range(4):
                    for t in range(MT):
                        oy = oy0 + wv * MT + t
                        if oy >= Ho:
                            continue
                        for nt in range(NT):
                            v = acc_main[wv, t, nt] + acc_low[wv, t, nt] / params.F16S_LO_SCALE + shift[16 * nt:16 * nt + 16][None, :]
                            if relu:
                                v = np.maximum(v, 0)
                            for i in range(16):
                                if ox0 + i < Wo:
                                    out[n, oy, ox0 + i, 16 * nt:16 * nt + 16] = v[i]
    return out


@pytest.mark.parametrize("K,S,cin,cout", LAYERS)
def test_emulated_kernel_matches_float64_convolution(K, S, cin, cout):
    g = torch.Generator().manual_seed(K * 100 + cin)
    H, W = (21, 37) if S == 1 else (26, 41)  # ragged tiles in both directions
    x = torch.randn(1, cin, H, W, generator=g) * (torch.rand(1, cin, 1, 1, generator=g) * 4)
    x[:, :, 3:5, 7:9] = 1e-5 * torch.randn(1, cin, 2, 2, generator=g)  # tiny values: the lo part must not underflow into nothing
    w = torch.randn(cout, cin, K, K, generator=g) * 0.2
    bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, torch.randn(cout, generator=g) * 0.1,
          torch.rand(cout, generator=g) + 0.5)
    wpk, shift = params.pack_conv_f16s(w, bn=bn)
    got = emulate(x.permute(0, 2, 3, 1).contiguous().numpy(), wpk, shift, K, S, cin, cout)
    sc = bn[0].double() / torch.sqrt(bn[3].double() + params.BN_EPS)
    want = F.conv2d(x.double(), w.double() * sc[:, None, None, None], (bn[1].double() - bn[2].double() * sc), S, K // 2).clamp_min(0)
    want = want.permute(0, 2, 3, 1).numpy()
    assert got.shape == want.shape
    err = np.abs(got - want).max() / np.abs(want).max()
    assert err < 5e-7, err  # fp32-convolution quality (an fp32 direct convolution of these shapes: 2-4e-7)


def test_split_keeps_22_bits():
    x = np.float32(np.random.default_rng(0).standard_normal(10000) * np.logspace(-4, 3, 10000))
    hi, lo = params.split_f16(x)
    rec = hi.astype(np.float64) + lo.astype(np.float64) / params.F16S_LO_SCALE
    err = np.abs(rec - x)
    big = np.abs(x) >= 2e-4  # hi is a NORMAL fp16 number down to 6.1e-5: 22 bits from there up (fp16 overflows at 65504)
    assert (err[big] / np.abs(x[big])).max() < 2.0 ** -21
    assert err[~big].max() < 1e-10  # below that the absolute error is what matters: 2^-24 (fp16 subnormal step) / 2048


def test_emulated_stem_conv1_on_the_matrix_cores():
    """pmn_stem_f16s step (3): conv1 with the roles swapped (A rows = output channels, B columns = pixels), the mid-patch addressing
    and params.pack_stem_conv1_f16s, against float64; conv0 is taken from torch (the kernel's VALU part is pmn_stem's)."""
    g = torch.Generator().manual_seed(7)
    H, W = 21, 35
    x = torch.rand(1, 3, H, W, generator=g)
    w0, w1 = torch.randn(8, 3, 3, 3, generator=g) * 0.3, torch.randn(8, 8, 3, 3, generator=g) * 0.2
    bn1 = (torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1, torch.randn(8, generator=g) * 0.1, torch.rand(8, generator=g) + 0.5)
    mid = F.relu(F.conv2d(x, w0, None, 1, 1))                        # [1,8,H,W] fp32 (conv0 + ReLU)
    w1a, shift = params.pack_stem_conv1_f16s(w1, bn1)
    assert w1a.shape == (3, 2, 64, 8) and w1a.dtype == np.float16
    sc = bn1[0].double() / torch.sqrt(bn1[3].double() + params.BN_EPS)
    want = F.relu(F.conv2d(mid.double(), w1.double() * sc[:, None, None, None], bn1[1].double() - bn1[2].double() * sc, 1, 1))
    want = want[0].permute(1, 2, 0).numpy()
    midn = mid[0].permute(1, 2, 0).numpy()
    lane = np.arange(64)
    li, kb = lane % 16, lane // 16
    out = np.zeros((H, W, 8), np.float32)
    MW = 18
    for oy0 in range(0, H, 16):
        for ox0 in range(0, W, 16):
            patch = np.zeros((MW, MW, 8), np.float32)                # conv0 output on the halo patch, ZERO outside the image
            for r in range(MW):
                for q in range(MW):
                    gy, gx = oy0 - 1 + r, ox0 - 1 + q
                    if 0 <= gy < H and 0 <= gx < W:
                        patch[r, q] = midn[gy, gx]
            ph, pl = params.split_f16(patch)
            for wv in range(4):
                accM = np.zeros((4, 16, 16)); accL = np.zeros((4, 16, 16))   # [t][row = cout][col = pixel]
                for ks in range(3):
                    q = np.minimum(4 * ks + kb, 8)
                    dy, dx = q // 3, q % 3
                    A_hi = np.zeros((16, 32)); A_lo = np.zeros((16, 32))
                    for e in range(8):
                        A_hi[li, 8 * kb + e] = w1a[ks, 0, :, e].astype(np.float64)
                        A_lo[li, 8 * kb + e] = w1a[ks, 1, :, e].astype(np.float64)
                    for t in range(4):
                        row = wv * 4 + t + dy
                        col = li + dx
                        B_hi = np.zeros((32, 16)); B_lo = np.zeros((32, 16))
                        for e in range(8):
                            B_hi[8 * kb + e, li] = ph[row, col, e].astype(np.float64)
                            B_lo[8 * kb + e, li] = pl[row, col, e].astype(np.float64)
                        accM[t] += A_hi @ B_hi
                        accL[t] += A_hi @ B_lo + A_lo @ B_hi
                for t in range(4):
                    oy = oy0 + wv * 4 + t
                    if oy >= H:
                        continue
                    v = np.maximum(accM[t] + accL[t] / params.F16S_LO_SCALE + np.pad(shift, (0, 8))[:, None], 0)  # [cout][pixel]
                    for i in range(16):
                        if ox0 + i < W:
                            out[oy, ox0 + i] = v[:8, i]
    err = np.abs(out - want).max() / np.abs(want).max()
    assert err < 5e-7, err


@pytest.mark.parametrize("cin,dil,n_p,n_e", [(64, 2, 32, 18), (32, 4, 16, 18), (16, 6, 0, 18)])
def test_emulated_offset_heads(cin, dil, n_p, n_e):
    """pmn_offset_heads_f16s: dilated taps, output rows zero-padded to a multiple of 16 (params.pack_offset_heads_f16s), bias, no ReLU."""
    g = torch.Generator().manual_seed(cin + dil)
    H, W = 19, 35
    x = torch.randn(1, cin, H, W, generator=g)
    cout = n_p + n_e
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    b = torch.randn(cout, generator=g)
    wpk, shift = params.pack_offset_heads_f16s(w, b)
    coutp = (cout + 15) // 16 * 16
    assert wpk.shape == (cin // 16, 5, coutp // 16, 2, 64, 8) and shift.shape == (coutp,)
    got = emulate(x.permute(0, 2, 3, 1).contiguous().numpy(), wpk, shift, 3, 1, cin, coutp, relu=False, dil=dil, CC=16)[..., :cout]
    want = F.conv2d(x.double(), w.double(), b.double(), 1, dil, dil).permute(0, 2, 3, 1).numpy()
    err = np.abs(got - want).max() / np.abs(want).max()
    assert err < 5e-7, err


def test_refine_conv3_packing_against_float64():
    """params.pack_refine_conv3_f16s (pmn_refine_fused's conv3: A operands = output channels, k-block q = 4 ks + kb = 2 tap + channel
    half) summed exactly as the kernel sums it -- hi*hi + (hi*lo + lo*hi) / 2048 over the five k-steps, the two padding blocks read
    block 17 against zero weights -- vs conv3 + BatchNorm + ReLU in float64 (reference models/net.py:90, 117)."""
    gen = torch.Generator().manual_seed(3)
    w3 = 0.15 * torch.randn(8, 16, 3, 3, generator=gen)
    bn = (0.5 + torch.rand(8, generator=gen), 0.1 * torch.randn(8, generator=gen), 0.1 * torch.randn(8, generator=gen),
          0.5 + torch.rand(8, generator=gen))
    x16 = torch.relu(torch.randn(1, 16, 9, 11, generator=gen)) * 3.0
    c3 = torch.nn.functional.conv2d(x16.double(), w3.double(), None, 1, 1)
    c3 = torch.relu(torch.nn.functional.batch_norm(c3, bn[2].double(), bn[3].double(), bn[0].double(), bn[1].double(), False, 0.0,
                                                   params.BN_EPS))[0].numpy()
    w3a, s3 = params.pack_refine_conv3_f16s(w3, bn)
    assert w3a.shape == (5, 2, 64, 8) and w3a.dtype == np.float16 and s3.shape == (8,)
    A = w3a.astype(np.float64)
    assert (A[:, :, [16 * kb + r for kb in range(4) for r in range(8, 16)]] == 0).all()  # MFMA rows 8..15 are padding
    assert (A[4, :, 32:] == 0).all()  # k-blocks 18, 19
    xh, xl = params.split_f16(x16[0].numpy())
    xh, xl = (np.pad(a.astype(np.float64), ((0, 0), (1, 1), (1, 1))) for a in (xh, xl))
    worst = 0.0
    for y in range(9):
        for x in (0, 4, 10):
            main, low = np.zeros(8), np.zeros(8)
            for ks in range(5):
                for kb in range(4):
                    q = min(4 * ks + kb, 17)
                    tap, cb = q >> 1, q & 1
                    dy, dx = divmod(tap, 3)
                    bh, bl = xh[8 * cb:8 * cb + 8, y + dy, x + dx], xl[8 * cb:8 * cb + 8, y + dy, x + dx]
                    ah, al = A[ks, 0, 16 * kb:16 * kb + 8], A[ks, 1, 16 * kb:16 * kb + 8]
                    main += ah @ bh
                    low += ah @ bl + al @ bh
            got = np.maximum(main + low / 2048.0 + s3.astype(np.float64), 0.0)
            worst = max(worst, float(np.abs(got - c3[:, y, x]).max() / np.abs(c3).max()))
    assert worst < 1e-6, worst
