#!/usr/bin/env python
"""Numerics study (CPU) for DESIGN.md section 9: the three pointwise MLPs (G -> 16 -> 8 -> 1: SimilarityNet, PixelwiseNet, FeatureWeightNet;
~26 M evaluations = ~13 GFLOP of fp32 VALU work per 1600x1200 depth map) as SPLIT-operand fp16 matrix-core products, the way
conv_f16s.hip runs FeatureNet's convolutions:
    x = hi + lo / 2048 (hi = fp16(x), lo = fp16((x - hi) * 2048)),   sum x*w ~= sum hi*hi + (sum hi*lo + sum lo*hi) / 2048, fp32 accumulation.
Inputs: the reference's own similarity tensors of the golden cascade (tests/golden/cascade_96x128_n2.npz) and the released checkpoint's
MLPs with BatchNorm folded in float64 (params.pack_mlp's arithmetic).  Reported against a float64 evaluation: the fp32 fma chain the
kernels run today, and the split form; for PixelwiseNet additionally how many (pixel, view) arg-max-over-D picks move."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import goldenutil as GU

g, P, kw = GU.load_case("default")
EPS = 1e-5


def folded(prefix, last):
    w0 = P[prefix + "conv0.conv.weight"].astype(np.float64).reshape(16, -1)
    w1 = P[prefix + "conv1.conv.weight"].astype(np.float64).reshape(8, 16)
    def bn(name):
        s = P[prefix + name + ".bn.weight"].astype(np.float64) / np.sqrt(P[prefix + name + ".bn.running_var"].astype(np.float64) + EPS)
        return s, P[prefix + name + ".bn.bias"].astype(np.float64) - P[prefix + name + ".bn.running_mean"].astype(np.float64) * s
    s0, t0 = bn("conv0"); s1, t1 = bn("conv1")
    w2 = P[prefix + last + ".weight"].astype(np.float64).reshape(8)
    b2 = float(P[prefix + last + ".bias"].reshape(-1)[0])
    return w0 * s0[:, None], t0, w1 * s1[:, None], t1, w2, b2


def split(x):
    x = x.astype(np.float32)
    hi = x.astype(np.float16)
    lo = ((x - hi.astype(np.float32)) * np.float32(2048)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def mm32(a, b):
    """[n,k] x [k,m] with fp32 accumulation in k order (what a chain of fmas / one MFMA k-block does up to association)."""
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for k in range(a.shape[1]):
        acc = (acc + a[:, k:k + 1].astype(np.float32) * b[k:k + 1, :].astype(np.float32)).astype(np.float32)
    return acc


def mm_split(x, w):
    xh, xl = split(x); wh, wl = split(w)
    return (mm32(xh, wh) + (mm32(xh, wl) + mm32(xl, wh)) * np.float32(1.0 / 2048)).astype(np.float32)


def mlp(x, W, mode):
    w0, t0, w1, t1, w2, b2 = W
    if mode == "f64":
        h = np.maximum(x.astype(np.float64) @ w0.T + t0, 0)
        h = np.maximum(h @ w1.T + t1, 0)
        return h @ w2 + b2
    f = np.float32
    mm = mm32 if mode == "f32" else mm_split
    h = np.maximum(mm(x.astype(f), w0.T.astype(f)) + t0.astype(f), 0).astype(f)
    h = np.maximum(mm(h, w1.T.astype(f)) + t1.astype(f), 0).astype(f)
    return (mm(h, w2.astype(f)[:, None])[:, 0] + f(b2)).astype(f)


def report(name, x, W, sigmoid=False, argmax_shape=None):
    ref = mlp(x, W, "f64")
    outs = {m: mlp(x, W, m).astype(np.float64) for m in ("f32", "split")}
    if sigmoid:
        ref = 1 / (1 + np.exp(-ref)); outs = {m: 1 / (1 + np.exp(-o)) for m, o in outs.items()}
    scale = np.abs(ref).max()
    line = f"{name:34s} n={x.shape[0]:7d} |out| max {scale:8.3f}: "
    for m, o in outs.items():
        e = np.abs(o - ref)
        line += f" {m}: max {e.max() / scale:.2e} rms {np.sqrt((e ** 2).mean()) / scale:.2e} |"
    if argmax_shape is not None:
        r = ref.reshape(argmax_shape).argmax(0)
        line += "  arg-max over D moved: " + ", ".join(f"{m} {int((o.reshape(argmax_shape).argmax(0) != r).sum())}" for m, o in outs.items()) \
                + f" of {r.size}"
    print(line)


for stage, G in ((3, 8), (2, 8), (1, 4)):
    pre = f"patchmatch_{stage}.evaluation."
    for it in range(1, kw["patchmatch_iteration"][stage - 1] + 1):
        sim = g[f"s{stage}_it{it}_similarity"]  # [1,G,D,h,w] aggregated over the views
        x = sim[0].reshape(G, -1).T
        report(f"SimilarityNet s{stage} it{it}", x, folded(pre + "similarity_net.", "similarity"))
v = g["s3_it1_view_similarity_0"][0]  # [G,D,h,w] of view 0
report("PixelwiseNet s3 it1 view 0 (sigmoid)", v.reshape(8, -1).T, folded("patchmatch_3.evaluation.pixel_wise_net.", "conv2"), sigmoid=True,
       argmax_shape=(v.shape[1], v.shape[2] * v.shape[3]))
