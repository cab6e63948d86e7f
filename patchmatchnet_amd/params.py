"""Host-side packing of the learned weights the kernels take through the kernel-argument segment.

MLP block layout (include/pmn_hip.h, PMN_MLP_FLOATS = 340 float32), ordered for the kernels' fused layer-1/2 walk:
    16 records of 20 floats, one per hidden unit j:  w0[j][0..7] (first G used) | w1[0..7][j] | t0[j] | 3 pad
    then  t1[8] | w2[8] | b2 | 3 pad
BatchNorm3d (eval mode, eps = 1e-5; reference models/module.py:43-72) is folded in float64 and rounded once:
    scale = gamma / sqrt(var + eps);  w' = w * scale;  t = beta - mean * scale
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from ._lib import MLP_FLOATS

BN_EPS = 1e-5


def _np64(t: torch.Tensor) -> np.ndarray:
    return t.detach().to("cpu", torch.float64).numpy()


def pack_mlp(conv0_w, bn0, conv1_w, bn1, last_w, last_b, eps: float = BN_EPS) -> np.ndarray:
    """bn0 / bn1 = (weight, bias, running_mean, running_var) tensors.  Returns float32[340]."""
    w0 = _np64(conv0_w).reshape(16, -1)
    G = w0.shape[1]
    assert G in (4, 8), f"unsupported group count {G}"
    w1 = _np64(conv1_w).reshape(8, 16)
    w2 = _np64(last_w).reshape(8)
    b2 = float(_np64(last_b).reshape(-1)[0])
    g0, b0, m0, v0 = (_np64(t) for t in bn0)
    g1, b1, m1, v1 = (_np64(t) for t in bn1)
    s0 = g0 / np.sqrt(v0 + eps)
    s1 = g1 / np.sqrt(v1 + eps)
    blk = np.zeros(MLP_FLOATS, np.float64)
    w0f = w0 * s0[:, None]          # [16, G]
    w1f = w1 * s1[:, None]          # [8, 16]
    t0 = b0 - m0 * s0
    for j in range(16):
        blk[20 * j:20 * j + G] = w0f[j]
        blk[20 * j + 8:20 * j + 16] = w1f[:, j]
        blk[20 * j + 16] = t0[j]
    blk[320:328] = b1 - m1 * s1
    blk[328:336] = w2
    blk[336] = b2
    return np.ascontiguousarray(blk.astype(np.float32))


def propagation_table(neighbors: int, dilation: int) -> np.ndarray:
    """(dy,dx) base offsets of adaptive propagation (reference models/patchmatch.py:331-360) as int32[2K]."""
    d = dilation
    ring = [(-d, -d), (-d, 0), (-d, d), (0, -d), (0, d), (d, -d), (d, 0), (d, d)]
    if neighbors == 4:
        t = [(-d, 0), (0, -d), (0, d), (d, 0)]
    elif neighbors == 8:
        t = ring
    elif neighbors == 16:
        t = ring + [(2 * a, 2 * b) for a, b in ring]
    else:
        raise NotImplementedError
    return np.ascontiguousarray(np.asarray(t, np.int32).reshape(-1))


def evaluation_table(neighbors: int, dilation: int) -> np.ndarray:
    """(dy,dx) base offsets of adaptive evaluation (reference models/patchmatch.py:361-392); dilation - 1."""
    d = dilation - 1
    nine = [(-d, -d), (-d, 0), (-d, d), (0, -d), (0, 0), (0, d), (d, -d), (d, 0), (d, d)]
    if neighbors == 9:
        t = nine
    elif neighbors == 17:
        t = nine + [(2 * a, 2 * b) for a, b in nine if a != 0 or b != 0]
    else:
        raise NotImplementedError
    return np.ascontiguousarray(np.asarray(t, np.int32).reshape(-1))


def versions(tensors: Sequence[torch.Tensor]) -> List[int]:
    """Cheap cache key: autograd version counters + storage addresses of the source tensors."""
    return [t._version for t in tensors] + [t.data_ptr() for t in tensors]


def pack_conv(weight: torch.Tensor, bn=None, bias=None, eps: float = BN_EPS):
    """Conv2d weight [cout,cin,K,K] (+ BatchNorm2d tensors (weight, bias, running_mean, running_var) or a conv bias) ->
    (float32 [K,K,cin,coutp], float32 [coutp]) in the layout pmn_conv2d reads; BatchNorm folded in float64."""
    w = _np64(weight)
    cout, cin, K, _ = w.shape
    if bn is not None:
        g, b, m, v = (_np64(t) for t in bn)
        s = g / np.sqrt(v + eps)
        w = w * s[:, None, None, None]
        shift = b - m * s
    elif bias is not None:
        shift = _np64(bias)
    else:
        shift = np.zeros(cout)
    tile = 8 if cout <= 8 else 16
    coutp = (cout + tile - 1) // tile * tile
    wp = np.zeros((K, K, cin, coutp), np.float64)
    wp[..., :cout] = w.transpose(2, 3, 1, 0)
    sp = np.zeros(coutp, np.float64)
    sp[:cout] = shift
    return np.ascontiguousarray(wp.astype(np.float32)), np.ascontiguousarray(sp.astype(np.float32))


def fold_fpn(output1, inner1, b_inner1, inner2, b_inner2, output2, output3):
    """FeatureNet's FPN head (reference models/net.py:57-67) is linear, so its 1x1 convolutions compose (float64 here):
        [f3 | u8] = W8 conv10,   [f2 | u4] = up(u8) + W4 conv7 + b4,   f1 = up(u4) + W2 conv4 + b2
    with W8 = [output1; output2; output3], W4 = [output2; output3] @ inner1, W2 = output3 @ inner2 (biases alike; a 1x1
    convolution commutes with bilinear up-sampling because the taps sum to one).  Arguments are the Conv2d weights
    [cout,cin,1,1] / biases; returns {8: (w [64,112], b [112]), 4: (w [32,48], b [48]), 2: (w [16,16], b [16])} as float32
    arrays in the [cin][cout] layout pmn_fpn_level reads."""
    O1, I1, I2, O2, O3 = (_np64(t)[:, :, 0, 0] for t in (output1, inner1, inner2, output2, output3))
    b1, b2 = _np64(b_inner1), _np64(b_inner2)
    O23 = np.concatenate([O2, O3], 0)  # [48,64]
    W8 = np.concatenate([O1, O23], 0)  # [112,64]
    W4, b4 = O23 @ I1, O23 @ b1        # [48,32], [48]
    W2, bb2 = O3 @ I2, O3 @ b2         # [16,16], [16]
    f = lambda a: np.ascontiguousarray(a.astype(np.float32))
    return {8: (f(W8.T), np.zeros(W8.shape[0], np.float32)), 4: (f(W4.T), f(b4)), 2: (f(W2.T), f(bb2))}


def pack_conv_mfma(weight: torch.Tensor, bn=None, bias=None, eps: float = BN_EPS):
    """Conv2d weight [cout,cin,K,K] (+ BatchNorm2d tensors or a conv bias) -> (float32 [K*K, cin/8, coutp/32, 64, 4],
    float32 [coutp]), coutp = cout rounded up to 32, in the B-operand order of pmn_conv2d_mfma: lane (h = lane>>5, i = lane&31) of column block nt reads the four
    weights w[nt*32 + i][8*c8 + 4*h + j][ky][kx], j = 0..3, as one 16-byte load.  BatchNorm folded in float64."""
    w = _np64(weight)
    cout, cin, K, _ = w.shape
    if cin % 8:
        raise ValueError("pack_conv_mfma: cin must be a multiple of 8")
    if bn is not None:
        g, b, m, v = (_np64(t) for t in bn)
        s = g / np.sqrt(v + eps)
        w = w * s[:, None, None, None]
        shift = b - m * s
    elif bias is not None:
        shift = _np64(bias)
    else:
        shift = np.zeros(cout)
    coutp = (cout + 31) // 32 * 32  # zero rows pad the last column block
    w = np.concatenate([w, np.zeros((coutp - cout, cin, K, K))], 0)
    shift = np.concatenate([shift, np.zeros(coutp - cout)])
    cout = coutp
    # [cout, cin, ky, kx] -> [ky, kx, c8, h, j, nt, i] -> [tap, c8, nt, h, i, j]
    t = w.transpose(2, 3, 1, 0).reshape(K * K, cin // 8, 2, 4, cout // 32, 32)
    t = t.transpose(0, 1, 4, 2, 5, 3).reshape(K * K, cin // 8, cout // 32, 64, 4)
    return np.ascontiguousarray(t.astype(np.float32)), np.ascontiguousarray(shift.astype(np.float32))


def pack_refine_tail(conv3_weight, conv3_bn, res_weight, eps: float = BN_EPS):
    """Refinement.conv3 (Conv2d 16->8 + BatchNorm) and Refinement.res (Conv2d 8->1, no bias) -> (w3 float32 [2,3,3,16,4],
    s3 float32 [8], wr float32 [3,3,8]) for pmn_refine_tail: conv3's output channels split into two halves of four (a half is
    wave-uniform in the kernel), BatchNorm folded in float64."""
    w, s = pack_conv(conv3_weight, bn=conv3_bn, eps=eps)  # [3,3,16,8], [8]
    w3 = np.ascontiguousarray(w.reshape(3, 3, 16, 2, 4).transpose(3, 0, 1, 2, 4))
    wr = np.ascontiguousarray(_np64(res_weight)[0].transpose(1, 2, 0).astype(np.float32))  # [1,8,3,3] -> [3,3,8]
    return w3, s, wr


_WINO_G = np.array([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])


def pack_conv_wino(weight: torch.Tensor, bn=None, bias=None, eps: float = BN_EPS):
    """3x3 Conv2d weight [C,C,3,3] with C in {16,32,64} (+ BatchNorm2d tensors or a conv bias) -> (float32 [C/16, 16, C/16, 64, 4],
    float32 [C]) for pmn_conv3x3_wino: the Winograd F(2x2,3x3) filter transform U = G g G^T (float64, BatchNorm scale folded
    in first), laid out so that lane (j = lane&15, kq = lane>>4) of output-channel block cb reads, for input-channel chunk cc
    and transform position pos = 4p+q, the four values U[p][q][cin = 16cc + 4kq + m][cout = 16cb + j], m = 0..3."""
    w = _np64(weight)
    cout, cin, K, _ = w.shape
    if K != 3 or cin != cout or cin not in (16, 32, 64):
        raise ValueError("pack_conv_wino: 3x3 with cin == cout in {16, 32, 64}")
    if bn is not None:
        g, b, m, v = (_np64(t) for t in bn)
        s = g / np.sqrt(v + eps)
        w = w * s[:, None, None, None]
        shift = b - m * s
    elif bias is not None:
        shift = _np64(bias)
    else:
        shift = np.zeros(cout)
    U = np.einsum("pa,kcab,qb->pqck", _WINO_G, w, _WINO_G).reshape(16, cin, cout)  # [pos][cin][cout]
    # cin = 16cc + 4kq + m, cout = 16cb + j  ->  [cc][pos][cb][kq*16 + j][m]
    t = U.reshape(16, cin // 16, 4, 4, cout // 16, 16).transpose(1, 0, 4, 2, 5, 3).reshape(cin // 16, 16, cout // 16, 64, 4)
    return np.ascontiguousarray(t.astype(np.float32)), np.ascontiguousarray(shift.astype(np.float32))


_WINO_G2 = np.array([[1.0, 0.0], [1.0, 1.0], [0.0, 1.0]])  # F(2,2): m1 = (d0-d1) g0, m2 = d1 (g0+g1), m3 = (d1-d2) g1


def pack_conv5x5s2_wino(weight: torch.Tensor, bn=None, bias=None, eps: float = BN_EPS):
    """5x5 Conv2d weight [cout,cin,5,5] of a stride-2 / padding-2 layer, (cin,cout) in {(8,16),(16,32),(32,64)} (+ BatchNorm2d
    tensors or a conv bias) -> (float32 [cin/8, 49, cout/16, 64, 2], float32 [cout]) for pmn_conv5x5s2_wino.  The kernel splits
    the convolution into the four parity sub-convolutions W_rs[a,b] = w[2a+r, 2b+s]; their filter transforms G_r W_rs G_s^T
    (G = F(2,3) matrix for the even / 3-tap phase, F(2,2) matrix for the odd / 2-tap phase; float64, BatchNorm scale folded in
    first) are enumerated as positions (r, s, p, q) in that nesting order, and lane (j = lane&15, kq = lane>>4) of output-channel
    block cb reads the two values U[pos][cin = 8cc + 2kq + m][cout = 16cb + j], m = 0, 1."""
    w = _np64(weight)
    cout, cin, K, _ = w.shape
    if K != 5 or (cin, cout) not in ((8, 16), (16, 32), (32, 64)):
        raise ValueError("pack_conv5x5s2_wino: 5x5 with (cin, cout) in {(8,16), (16,32), (32,64)}")
    if bn is not None:
        g, b, m, v = (_np64(t) for t in bn)
        s = g / np.sqrt(v + eps)
        w = w * s[:, None, None, None]
        shift = b - m * s
    elif bias is not None:
        shift = _np64(bias)
    else:
        shift = np.zeros(cout)
    gm = {0: _WINO_G, 1: _WINO_G2}
    U = []
    for r in (0, 1):
        for s_ in (0, 1):
            u = np.einsum("pa,kcab,qb->pqck", gm[r], w[:, :, r::2, s_::2], gm[s_])  # [nr, ns, cin, cout]
            U.append(u.reshape(-1, cin, cout))
    U = np.concatenate(U, 0)  # [49, cin, cout]
    assert U.shape[0] == 49
    # cin = 8cc + 2kq + m, cout = 16cb + j  ->  [cc][pos][cb][kq*16 + j][m]
    t = U.reshape(49, cin // 8, 4, 2, cout // 16, 16).transpose(1, 0, 4, 2, 5, 3).reshape(cin // 8, 49, cout // 16, 64, 2)
    return np.ascontiguousarray(t.astype(np.float32)), np.ascontiguousarray(shift.astype(np.float32))


F16S_LO_SCALE = 2048.0  # x = hi + lo / 2048 with hi = fp16(x), lo = fp16((x - hi) * 2048): 22 significant bits, lo in fp16's normal range


def f16s_chunk(cin: int, ksize: int) -> int:
    """Input-channel chunk (channels staged in LDS at a time) of pmn_conv2d_f16s for a layer shape."""
    if ksize == 3:
        return 16
    return 16 if cin == 32 else 8


F16_MAX = 65504.0  # largest finite float16: |x| >= F16_MAX makes hi = +-inf (include/pmn_hip.h, "fp16-split entry points")


class F16DomainError(ValueError):
    """A (BatchNorm-folded) weight lies outside float16's finite range: the fp16-split kernels cannot represent it; the modules
    fall back to the fp32 kernels for the whole network and say so (net.FeatureNet / net.Refinement / patchmatch.PatchMatch)."""


def split_f16(x: np.ndarray):
    """float -> (hi, lo) float16 pair with x ~= hi + lo / F16S_LO_SCALE (relative error 2^-22 for 6.1e-5 <= |x| < 65504; smaller
    magnitudes keep an ABSOLUTE error <= 3e-8).  Raises F16DomainError for |x| >= 65504 or a non-finite value: a tiny BatchNorm
    running_var can fold into such a scale, and the kernels themselves do not check (hi = inf, lo = NaN, and the zero-weight padding
    k-blocks turn 0 * inf into NaN everywhere)."""
    x64 = np.asarray(x, np.float64)
    if x64.size and not (np.isfinite(x64).all() and float(np.abs(x64).max()) < F16_MAX):
        raise F16DomainError(f"weight magnitude {float(np.nanmax(np.abs(x64))):.3e} outside float16's finite range (< {F16_MAX})")
    x32 = x64.astype(np.float32)
    hi = x32.astype(np.float16)
    lo = ((x32 - hi.astype(np.float32)) * np.float32(F16S_LO_SCALE)).astype(np.float16)
    return hi, lo


def _pack_f16s(w: np.ndarray, CC: int) -> np.ndarray:
    """[cout (multiple of 16), cin, K, K] float64 -> float16 [cin/CC, k-steps, cout/16, 2, 64, 8] in B-operand lane order (see
    ``pack_conv_f16s``)."""
    cout, cin, K, _ = w.shape
    ncb, chunks, nt = CC // 8, cin // CC, cout // 16
    nq = K * K * ncb
    ksteps = (nq + 3) // 4
    full = np.zeros((chunks, ksteps * 4, cout, 8), np.float64)  # [chunk][q][cout][e]
    for ch in range(chunks):
        for q in range(nq):
            tap, cb = divmod(q, ncb)
            dy, dx = divmod(tap, K)
            full[ch, q] = w[:, ch * CC + 8 * cb:ch * CC + 8 * cb + 8, dy, dx]
    # [chunk][ks][kb][nt][n][e] -> [chunk][ks][nt][kb][n][e]: lane = 16 kb + n
    full = full.reshape(chunks, ksteps, 4, nt, 16, 8).transpose(0, 1, 3, 2, 4, 5).reshape(chunks, ksteps, nt, 64, 8)
    hi, lo = split_f16(full)
    return np.ascontiguousarray(np.stack((hi, lo), axis=3))  # [chunk][ks][nt][split][lane][8]


def pack_conv_f16s(weight: torch.Tensor, bn=None, bias=None, eps: float = BN_EPS):
    """Conv2d weight [cout,cin,K,K] (3x3 stride 1 or 5x5 stride 2; cin in {8,16,32,64}, cout in {16,32,64}) (+ BatchNorm2d tensors or
    a conv bias) -> (float16 [chunks, ksteps, cout/16, 2, 64, 8], float32 [cout]) for pmn_conv2d_f16s: the B operands of
    v_mfma_f32_16x16x32_f16 in lane order, BatchNorm scale folded in (float64), every weight SPLIT into hi / lo float16 parts
    (``split_f16``).  The GEMM's k axis of one chunk of CC = f16s_chunk(cin, K) input channels is cut into blocks of 8 channels,
    block q = tap * (CC / 8) + cb  (tap = dy * K + dx, channels [8 cb, 8 cb + 8) of the chunk); k-step ks holds blocks 4 ks .. 4 ks + 3 and
    lane l = 16 kb + n of output-channel tile nt reads w[cout = 16 nt + n][chunk * CC + 8 cb + e][dy][dx], e = 0..7, for q = 4 ks + kb
    (zeros for the padding blocks q >= K * K * CC / 8)."""
    w = _np64(weight)
    cout, cin, K, _ = w.shape
    if (K, cin, cout) not in ((5, 8, 16), (3, 16, 16), (5, 16, 32), (3, 32, 32), (5, 32, 64), (3, 64, 64)):
        raise ValueError("pack_conv_f16s: unsupported layer shape")
    if bn is not None:
        g, b, m, v = (_np64(t) for t in bn)
        sc = g / np.sqrt(v + eps)
        w = w * sc[:, None, None, None]
        shift = b - m * sc
    elif bias is not None:
        shift = _np64(bias)
    else:
        shift = np.zeros(cout)
    return _pack_f16s(w, f16s_chunk(cin, K)), np.ascontiguousarray(shift.astype(np.float32))


def pack_offset_heads_f16s(weight: torch.Tensor, bias: torch.Tensor):
    """The row-concatenated 3x3 filters [cout,cin,3,3] and biases of a stage's offset heads (propa_conv rows, then eval_conv:
    reference models/patchmatch.py:288-311) for pmn_offset_heads_f16s: rows zero-padded to a multiple of 16, chunks of 16 input
    channels -> (float16 [cin/16, k-steps, coutp/16, 2, 64, 8], float32 [coutp])."""
    w = _np64(weight)
    cout, cin, K, _ = w.shape
    if K != 3 or cin % 16:
        raise ValueError("pack_offset_heads_f16s: 3x3 filters over a multiple of 16 input channels")
    coutp = (cout + 15) // 16 * 16
    wp = np.zeros((coutp, cin, 3, 3), np.float64)
    wp[:cout] = w
    shift = np.zeros(coutp, np.float64)
    shift[:cout] = _np64(bias)
    return _pack_f16s(wp, 16), np.ascontiguousarray(shift.astype(np.float32))


def pack_stem_conv1_f16s(weight: torch.Tensor, bn, eps: float = BN_EPS):
    """FeatureNet conv1 (8 -> 8, 3x3) for pmn_stem_f16s: the weights as the A operands (rows = output channels) of
    v_mfma_f32_16x16x32_f16 -> (float16 [3 k-steps][2 (hi|lo)][64 lanes][8], float32 shift [8]).  Lane l = 16 kb + row: row = output
    channel (rows 8..15 zero), k-block q = 4 ks + kb = tap dy * 3 + dx (blocks 9..11 zero), the 8 values = input channels 0..7;
    BatchNorm scale folded in float64, every weight split with ``split_f16``."""
    w = _np64(weight)
    if w.shape != (8, 8, 3, 3):
        raise ValueError("pack_stem_conv1_f16s: conv1 is 8 -> 8, 3x3")
    g, b, m, v = (_np64(t) for t in bn)
    sc = g / np.sqrt(v + eps)
    w = w * sc[:, None, None, None]
    shift = b - m * sc
    full = np.zeros((3, 4, 16, 8), np.float64)  # [ks][kb][row][e]
    for q in range(9):
        dy, dx = divmod(q, 3)
        full[q // 4, q % 4, :8, :] = w[:, :, dy, dx]
    hi, lo = split_f16(full.reshape(3, 64, 8))
    return np.ascontiguousarray(np.stack((hi, lo), axis=1)), np.ascontiguousarray(shift.astype(np.float32))


def pack_refine_conv3_f16s(weight: torch.Tensor, bn, eps: float = BN_EPS):
    """Refinement.conv3 (16 -> 8, 3x3 + BatchNorm) for pmn_refine_fused: the weights as the A operands (rows = output channels) of
    v_mfma_f32_16x16x32_f16 -> (float16 [5 k-steps][2 (hi|lo)][64 lanes][8], float32 shift [8]).  Lane l = 16 kb + row (rows 8..15
    zero); k-block q = 4 ks + kb = 2 tap + cb (tap = dy * 3 + dx, cb = half of the 16 input channels; blocks 18, 19 zero); the 8 values =
    input channels 8 cb .. 8 cb + 7; BatchNorm scale folded in float64, every weight split with ``split_f16``."""
    w = _np64(weight)
    if w.shape != (8, 16, 3, 3):
        raise ValueError("pack_refine_conv3_f16s: conv3 is 16 -> 8, 3x3")
    g, b, m, v = (_np64(t) for t in bn)
    sc = g / np.sqrt(v + eps)
    w = w * sc[:, None, None, None]
    shift = b - m * sc
    full = np.zeros((5, 4, 16, 8), np.float64)  # [ks][kb][row][e]
    for q in range(18):
        tap, cb = q >> 1, q & 1
        dy, dx = divmod(tap, 3)
        full[q // 4, q % 4, :8, :] = w[:, 8 * cb:8 * cb + 8, dy, dx]
    hi, lo = split_f16(full.reshape(5, 64, 8))
    return np.ascontiguousarray(np.stack((hi, lo), axis=1)), np.ascontiguousarray(shift.astype(np.float32))


def pack_deconv(weight: torch.Tensor, bn=None, eps: float = BN_EPS):
    """ConvTranspose2d weight [cin,cout,K,K] (+ BatchNorm2d tensors) -> (float32 [K,K,cin,cout], float32 [cout]) for
    pmn_deconv3x3s2; BatchNorm folded in float64."""
    w = _np64(weight)
    cin, cout, K, _ = w.shape
    shift = np.zeros(cout)
    if bn is not None:
        g, b, m, v = (_np64(t) for t in bn)
        s = g / np.sqrt(v + eps)
        w = w * s[None, :, None, None]
        shift = b - m * s
    return (np.ascontiguousarray(w.transpose(2, 3, 0, 1).astype(np.float32)),
            np.ascontiguousarray(shift.astype(np.float32)))
