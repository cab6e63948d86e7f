"""CPU oracle for the learned-PatchMatch hot path (numpy composition over oracle/pmn_oracle.c).

TEST INFRASTRUCTURE ONLY -- this module is the parity checker for the HIP path.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.  The product
package ``patchmatchnet_amd`` never imports, links or calls anything under ``oracle/``.

Parity status: the reference ships no golden vectors / unit tests for this path, so the
oracle is pinned against outputs of the reference itself: ``tests/golden/make_golden.py`` imports the
reference (read-only, in the authoring container) and dumps every hot-path intermediate for small seeded
inputs; ``tests/test_oracle_golden.py`` checks this restatement against those committed fixtures, and
``tests/test_oracle_vs_reference.py`` re-checks live whenever ``/root/reference`` is present.

Every function cites the reference lines it restates (paths relative to the reference checkout).
All arrays are float32 numpy, NCHW, with an explicit leading batch dimension.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpmn_oracle.so")
_LIB: Optional[ctypes.CDLL] = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)


def build(force: bool = False) -> str:
    """Compile oracle/pmn_oracle.c -> oracle/libpmn_oracle.so with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "pmn_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libpmn_oracle.so"])
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.pmo_num_threads.restype = ctypes.c_int
        L.pmo_set_num_threads.argtypes = [ctypes.c_int]
        L.pmo_warp_positions.argtypes = [_f32p, _f32p, _f32p] + [ctypes.c_int] * 5 + [_f32p, _f32p]
        L.pmo_differentiable_warping.argtypes = [_f32p, _f32p, _f32p, _f32p] + [ctypes.c_int] * 6 + [_f32p]
        L.pmo_warp_similarity.argtypes = [_f32p] * 5 + [ctypes.c_int] * 7 + [_f32p]
        L.pmo_pointwise_mlp.argtypes = [_f32p, ctypes.c_int, ctypes.c_int64, _f32p, _f32p, _f32p, _f32p, _f32p,
                                        ctypes.c_float, ctypes.c_float, ctypes.c_int, _f32p]
        L.pmo_neighbor_positions.argtypes = [_f32p, _i32p] + [ctypes.c_int] * 3 + [_f32p, _f32p]
        L.pmo_neighbor_gather.argtypes = [_f32p, _f32p, _i32p] + [ctypes.c_int] * 4 + [_f32p]
        i64 = ctypes.c_int64
        L.pmo_feature_corr.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, i64, _f32p]
        L.pmo_normalised_inverse_depth.argtypes = [_f32p, i64, ctypes.c_float, ctypes.c_float, _f32p]
        L.pmo_depth_weight.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, i64, ctypes.c_float, _f32p]
        L.pmo_weight_normalise.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, i64]
        L.pmo_weighted_neighbor_sum.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, i64, _f32p]
        L.pmo_view_accumulate.argtypes = [_f32p, _f32p, ctypes.c_int, i64, _f32p, _f32p]
        L.pmo_view_normalise.argtypes = [_f32p, _f32p, ctypes.c_int, i64]
        if "OMP_NUM_THREADS" not in os.environ:
            # default cap: on a 256-thread host the per-call OpenMP regions are 7x SLOWER with all threads than with 8 (they fight
            # torch's own pool; bench.py's cpu_baseline: 2.3 s -> 16 s for one cascade); set_num_threads() overrides
            L.pmo_set_num_threads(min(os.cpu_count() or 1, 32))
        _LIB = L
    return _LIB


def num_threads() -> int:
    return int(lib().pmo_num_threads())


def set_num_threads(n: int) -> None:
    lib().pmo_set_num_threads(int(n))


def _c(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: np.ndarray):
    return a.ctypes.data_as(_f32p)


# --------------------------------------------------------------------------------------------------------
# a2 / a3: warping + group-wise correlation
# --------------------------------------------------------------------------------------------------------

def relative_pose(src_proj: np.ndarray, ref_proj: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """models/module.py:148-150 -- proj = src_proj @ inverse(ref_proj); rot = proj[:3,:3]; trans = proj[:3,3]."""
    proj = np.matmul(src_proj.astype(np.float32), np.linalg.inv(ref_proj.astype(np.float32))).astype(np.float32)
    return _c(proj[:3, :3]), _c(proj[:3, 3])


def differentiable_warping(src_fea: np.ndarray, src_proj: np.ndarray, ref_proj: np.ndarray,
                           depth_samples: np.ndarray) -> np.ndarray:
    """models/module.py:130-181.  src_fea [B,C,Hs,Ws], projs [B,4,4], depth_samples [B,D,H,W] -> [B,C,D,H,W]."""
    B, C, hs, ws = src_fea.shape
    _, D, h, w = depth_samples.shape
    out = np.empty((B, C, D, h, w), np.float32)
    for b in range(B):
        rot, trans = relative_pose(src_proj[b], ref_proj[b])
        s, d = _c(src_fea[b]), _c(depth_samples[b])
        lib().pmo_differentiable_warping(_p(s), _p(rot), _p(trans), _p(d), C, D, h, w, hs, ws, _p(out[b]))
    return out


def warp_similarity(ref_fea: np.ndarray, src_fea: np.ndarray, src_proj: np.ndarray, ref_proj: np.ndarray,
                    depth_samples: np.ndarray, G: int) -> np.ndarray:
    """models/patchmatch.py:199-203 -- (warped * ref).view(B,G,C/G,D,h,w).mean(2) for ONE source view -> [B,G,D,h,w]."""
    B, C, h, w = ref_fea.shape
    hs, ws = src_fea.shape[2:]
    D = depth_samples.shape[1]
    out = np.empty((B, G, D, h, w), np.float32)
    for b in range(B):
        rot, trans = relative_pose(src_proj[b], ref_proj[b])
        r, s, d = _c(ref_fea[b]), _c(src_fea[b]), _c(depth_samples[b])
        lib().pmo_warp_similarity(_p(r), _p(s), _p(rot), _p(trans), _p(d), C, G, D, h, w, hs, ws, _p(out[b]))
    return out


# --------------------------------------------------------------------------------------------------------
# a15: pointwise MLP (ConvBnReLU3D x2 + Conv3d) shared by PixelwiseNet / SimilarityNet / FeatureWeightNet
# --------------------------------------------------------------------------------------------------------

def _mlp_params(params: Dict[str, np.ndarray], prefix: str, last: str):
    def bn(name):
        return _c(np.concatenate([params[f"{prefix}.{name}.bn.weight"], params[f"{prefix}.{name}.bn.bias"],
                                  params[f"{prefix}.{name}.bn.running_mean"],
                                  params[f"{prefix}.{name}.bn.running_var"]]))
    w0 = _c(params[f"{prefix}.conv0.conv.weight"].reshape(16, -1))
    w1 = _c(params[f"{prefix}.conv1.conv.weight"].reshape(8, 16))
    w2 = _c(params[f"{prefix}.{last}.weight"].reshape(8))
    b2 = float(params[f"{prefix}.{last}.bias"].reshape(-1)[0])
    return w0, bn("conv0"), w1, bn("conv1"), w2, b2


def pointwise_mlp(x: np.ndarray, params: Dict[str, np.ndarray], prefix: str, last: str, sigmoid: bool) -> np.ndarray:
    """x [B,G,...] -> [B,...]; models/module.py:43-72 (ConvBnReLU3D), eps = 1e-5 (BatchNorm3d default)."""
    w0, bn0, w1, bn1, w2, b2 = _mlp_params(params, prefix, last)
    B, G = x.shape[:2]
    rest = x.shape[2:]
    M = int(np.prod(rest))
    out = np.empty((B,) + rest, np.float32)
    for b in range(B):
        xb = _c(x[b]).reshape(G, M)
        lib().pmo_pointwise_mlp(_p(xb), G, M, _p(w0), _p(bn0), _p(w1), _p(bn1), _p(w2), b2, 1e-5,
                                1 if sigmoid else 0, _p(out[b]))
    return out


def pixelwise_net(similarity: np.ndarray, params, prefix: str, responses: Optional[list] = None) -> Tuple[np.ndarray, np.ndarray]:
    """models/patchmatch.py:695-702 -- max over D of sigmoid(MLP(sim)).  Returns ([B,1,h,w], argmax [B,h,w]); ``responses`` (a list)
    additionally receives the per-hypothesis responses [B,D,h,w] the max was taken over (the parity tests show with them that every
    arg-max mismatch at full size is a tie)."""
    resp = pointwise_mlp(similarity, params, prefix, "conv2", sigmoid=True)  # [B,D,h,w]
    if responses is not None:
        responses.append(resp)
    return resp.max(axis=1, keepdims=True), resp.argmax(axis=1)


# --------------------------------------------------------------------------------------------------------
# a10: neighbour tables + gathers
# --------------------------------------------------------------------------------------------------------

def propagation_table(neighbors: int, dilation: int) -> np.ndarray:
    """models/patchmatch.py:331-360 -- [K,2] (dy,dx) base offsets of adaptive propagation."""
    d = dilation
    ring = [[-d, -d], [-d, 0], [-d, d], [0, -d], [0, d], [d, -d], [d, 0], [d, d]]
    if neighbors == 4:
        t = [[-d, 0], [0, -d], [0, d], [d, 0]]
    elif neighbors == 8:
        t = ring
    elif neighbors == 16:
        t = ring + [[2 * a, 2 * b] for a, b in ring]
    else:
        raise NotImplementedError
    return np.asarray(t, np.int32)


def evaluation_table(neighbors: int, dilation: int) -> np.ndarray:
    """models/patchmatch.py:361-392 -- [K,2] (dy,dx); evaluation dilation is (propagation range - 1)."""
    d = dilation - 1
    nine = [[-d, -d], [-d, 0], [-d, d], [0, -d], [0, 0], [0, d], [d, -d], [d, 0], [d, d]]
    if neighbors == 9:
        t = nine
    elif neighbors == 17:
        t = nine + [[2 * a, 2 * b] for a, b in nine if a != 0 or b != 0]
    else:
        raise NotImplementedError
    return np.asarray(t, np.int32)


def neighbor_gather(inp: np.ndarray, offsets: np.ndarray, table: np.ndarray) -> np.ndarray:
    """get_grid (patchmatch.py:396-426) + F.grid_sample(bilinear, border, align_corners=False).

    inp [B,Cn,h,w], offsets [B,2K,h,w] (learned, channel 2k -> x, 2k+1 -> y), table [K,2] -> [B,Cn,K,h,w]."""
    B, Cn, h, w = inp.shape
    K = table.shape[0]
    tab = np.ascontiguousarray(table, np.int32)
    out = np.empty((B, Cn, K, h, w), np.float32)
    for b in range(B):
        i, o = _c(inp[b]), _c(offsets[b])
        lib().pmo_neighbor_gather(_p(i), _p(o), tab.ctypes.data_as(_i32p), Cn, K, h, w, _p(out[b]))
    return out


# --------------------------------------------------------------------------------------------------------
# a8 / a9 / a12 / a11 / a6 / a7: the per-iteration pieces
# --------------------------------------------------------------------------------------------------------

def depth_initialization(depth: Optional[np.ndarray], noise: Optional[np.ndarray], depth_min: np.ndarray,
                         depth_max: np.ndarray, num_sample: int, interval_scale: float) -> np.ndarray:
    """models/patchmatch.py:53-94.  depth None/empty -> 48 random bins from ``noise`` [B,48,h,w] (the torch.rand
    draw of :61-62, injected so both sides see the same numbers); else local perturbation around ``depth``."""
    one = np.float32(1.0)
    inv_min = (one / depth_min.astype(np.float32)).reshape(-1, 1, 1, 1)
    inv_max = (one / depth_max.astype(np.float32)).reshape(-1, 1, 1, 1)
    if depth is None or depth.size == 0:
        n = 48
        u = noise.astype(np.float32) + np.arange(n, dtype=np.float32).reshape(1, n, 1, 1)
        inv = inv_max + u / np.float32(n) * (inv_min - inv_max)
        return (one / inv).astype(np.float32)
    if num_sample == 1:
        return depth.astype(np.float32)
    k = np.arange(-num_sample // 2, num_sample // 2, 1).astype(np.float32).reshape(1, num_sample, 1, 1)
    interval = (inv_min - inv_max) * np.float32(interval_scale)
    inv = one / depth.astype(np.float32) + interval * k
    inv = np.clip(inv, inv_max, inv_min)
    return (one / inv).astype(np.float32)


def propagation(depth_sample: np.ndarray, propa_offsets: np.ndarray, table: np.ndarray) -> np.ndarray:
    """models/patchmatch.py:115-124 -- gather channel D//2 at K neighbours, concat, sort ascending along D."""
    D = depth_sample.shape[1]
    centre = depth_sample[:, D // 2:D // 2 + 1]
    nb = neighbor_gather(centre, propa_offsets, table)[:, 0]  # [B,K,h,w]
    return np.sort(np.concatenate([depth_sample, nb], axis=1), axis=1).astype(np.float32)


def _sigmoid(x: np.ndarray) -> np.ndarray:
    return (np.float32(1.0) / (np.float32(1.0) + np.exp(-x, dtype=np.float32))).astype(np.float32)


def depth_weight(depth_sample: np.ndarray, depth_min: np.ndarray, depth_max: np.ndarray, eval_offsets: np.ndarray,
                 table: np.ndarray, interval_scale: float) -> np.ndarray:
    """models/patchmatch.py:650-669 -> [B,D,K,h,w]."""
    one = np.float32(1.0)
    B, D, h, w = depth_sample.shape
    K = table.shape[0]
    ds = _c(depth_sample)
    x = np.empty_like(ds)
    for b in range(B):
        inv_min, inv_max = one / np.float32(depth_min[b]), one / np.float32(depth_max[b])
        lib().pmo_normalised_inverse_depth(_p(ds[b]), D * h * w, float(inv_min), float(inv_max), _p(x[b]))
    x1 = neighbor_gather(x, eval_offsets, table)  # [B,D,K,h,w]
    out = np.empty_like(x1)
    for b in range(B):
        lib().pmo_depth_weight(_p(x[b]), _p(x1[b]), D, K, h * w, float(np.float32(interval_scale)), _p(out[b]))
    return out


def feature_weight_net(ref_feature: np.ndarray, eval_offsets: np.ndarray, table: np.ndarray, G: int, params,
                       prefix: str) -> np.ndarray:
    """models/patchmatch.py:613-624 -> [B,K,h,w]."""
    B, C, h, w = ref_feature.shape
    K = table.shape[0]
    nb = neighbor_gather(ref_feature, eval_offsets, table)  # [B,C,K,h,w]
    ref = _c(ref_feature)
    corr = np.empty((B, G, K, h, w), np.float32)  # (nb * ref).view(B,G,C/G,K,h,w).mean(2)
    for b in range(B):
        lib().pmo_feature_corr(_p(nb[b]), _p(ref[b]), C, G, K, h * w, _p(corr[b]))
    return pointwise_mlp(corr, params, prefix, "similarity", sigmoid=True)


def similarity_net(similarity: np.ndarray, eval_offsets: np.ndarray, table: np.ndarray, weight: np.ndarray, params,
                   prefix: str) -> Tuple[np.ndarray, np.ndarray]:
    """models/patchmatch.py:565-577.  Returns (score_pre_softmax [B,D,h,w], pointwise cost [B,D,h,w])."""
    cost = pointwise_mlp(similarity, params, prefix, "similarity", sigmoid=False)
    nb = neighbor_gather(cost, eval_offsets, table)  # [B,D,K,h,w]
    B, D, K, h, w = nb.shape
    weight = _c(weight)
    score = np.empty((B, D, h, w), np.float32)  # (nb * weight).sum(2)
    for b in range(B):
        lib().pmo_weighted_neighbor_sum(_p(nb[b]), _p(weight[b]), D, K, h * w, _p(score[b]))
    return score, cost


def softmax_over_depth(score: np.ndarray) -> np.ndarray:
    """models/patchmatch.py:142,221 -- exp(log_softmax(score, dim=1))."""
    m = score.max(axis=1, keepdims=True)
    z = score - m
    lse = np.log(np.exp(z, dtype=np.float32).sum(axis=1, keepdims=True, dtype=np.float32))
    return np.exp(z - lse, dtype=np.float32)


def regress_depth(depth_sample: np.ndarray, prob: np.ndarray, is_inverse: bool) -> np.ndarray:
    """models/patchmatch.py:226-237 -> [B,h,w]."""
    D = depth_sample.shape[1]
    if is_inverse:
        idx = (np.arange(D, dtype=np.float32).reshape(1, D, 1, 1) * prob).sum(axis=1, dtype=np.float32)
        inv_min = np.float32(1.0) / depth_sample[:, -1]
        inv_max = np.float32(1.0) / depth_sample[:, 0]
        inv = inv_max + idx / np.float32(D - 1) * (inv_min - inv_max)
        return (np.float32(1.0) / inv).astype(np.float32)
    return (depth_sample * prob).sum(axis=1, dtype=np.float32)


def evaluation(ref_feature, src_features: Sequence[np.ndarray], ref_proj, src_projs: Sequence[np.ndarray],
               depth_sample, eval_offsets, table, weight, view_weights: Optional[np.ndarray], is_inverse: bool, G: int,
               params, prefix: str) -> Dict[str, np.ndarray]:
    """models/patchmatch.py:179-239.  Returns dict with depth [B,h,w], score [B,D,h,w], view_weights [B,N,h,w] and
    the intermediates the parity tests compare (aggregated similarity, pointwise cost, view-weight argmax)."""
    B, C, h, w = ref_feature.shape
    D = depth_sample.shape[1]
    have_vw = view_weights is not None and view_weights.size > 0
    assert len(src_features) == len(src_projs)
    if have_vw:
        assert len(src_features) == view_weights.shape[1]
    weight_sum = np.full((B, h, w), 1e-5, np.float32)
    sim_sum = np.zeros((B, G, D, h, w), np.float32)
    vw_list, arg_list, resp_list = [], [], []
    for i, (src_fea, src_proj) in enumerate(zip(src_features, src_projs)):
        sim = warp_similarity(ref_feature, src_fea, src_proj, ref_proj, depth_sample, G)
        if have_vw:
            vw = view_weights[:, i:i + 1]
        else:
            vw, arg = pixelwise_net(sim, params, f"{prefix}.pixel_wise_net", resp_list)
            vw_list.append(vw)
            arg_list.append(arg)
        vw = _c(vw)
        for b in range(B):  # sim_sum += sim * vw; weight_sum += vw
            lib().pmo_view_accumulate(_p(sim[b]), _p(vw[b]), G * D, h * w, _p(sim_sum[b]), _p(weight_sum[b]))
    for b in range(B):  # similarity = sim_sum / weight_sum
        lib().pmo_view_normalise(_p(sim_sum[b]), _p(weight_sum[b]), G * D, h * w)
    similarity = sim_sum
    score, cost = similarity_net(similarity, eval_offsets, table, weight, params, f"{prefix}.similarity_net")
    prob = softmax_over_depth(score)
    out = {"similarity": similarity, "cost": cost, "score": prob}
    if not have_vw:
        view_weights = np.concatenate(vw_list, axis=1)
        out["view_weight_argmax"] = np.stack(arg_list, axis=1)
        out["view_weight_responses"] = np.stack(resp_list, axis=1)  # [B,N,D,h,w]
    out["view_weights"] = view_weights
    out["depth"] = regress_depth(depth_sample, prob, is_inverse)
    return out


# --------------------------------------------------------------------------------------------------------
# a16: offset heads (3x3 dilated conv with bias) -- stays on MIOpen in the product; numpy here for a full cascade
# --------------------------------------------------------------------------------------------------------

def dilated_conv3x3(x: np.ndarray, weight: np.ndarray, bias: np.ndarray, dilation: int) -> np.ndarray:
    """nn.Conv2d(k=3, padding=dilation, dilation=dilation, bias=True) (patchmatch.py:288-311)."""
    B, C, h, w = x.shape
    d = dilation
    xp = np.pad(x.astype(np.float32), ((0, 0), (0, 0), (d, d), (d, d)))
    out = np.zeros((B, weight.shape[0], h, w), np.float32)
    for ky in range(3):
        for kx in range(3):
            patch = np.ascontiguousarray(xp[:, :, ky * d:ky * d + h, kx * d:kx * d + w]).reshape(B, C, h * w)
            # [O,C] @ [B,C,hw] through BLAS sgemm (threaded); tap order ky, kx as before
            out += np.matmul(weight[:, :, ky, kx].astype(np.float32), patch).reshape(B, -1, h, w)
    return out + bias.astype(np.float32).reshape(1, -1, 1, 1)


# --------------------------------------------------------------------------------------------------------
# a13: one PatchMatch stage, a1: the cascade, a14: the confidence epilogue
# --------------------------------------------------------------------------------------------------------

class StageConfig:
    def __init__(self, stage: int, iterations: int, num_sample: int, interval_scale: float, dilation: int, G: int,
                 propagate_neighbors: int, evaluate_neighbors: int):
        self.stage, self.iterations, self.num_sample = stage, iterations, num_sample
        self.interval_scale, self.dilation, self.G = interval_scale, dilation, G
        self.propagate_neighbors, self.evaluate_neighbors = propagate_neighbors, evaluate_neighbors


def default_stage_configs(interval_scale=(0.005, 0.0125, 0.025), propagation_range=(6, 4, 2), iteration=(1, 2, 2),
                          num_sample=(8, 8, 16), propagate_neighbors=(0, 8, 16), evaluate_neighbors=(9, 9, 9)):
    """models/net.py:150-172 -- lists are indexed [stage1, stage2, stage3]; G = [4, 8, 8]."""
    G = (4, 8, 8)
    return {s: StageConfig(s, iteration[s - 1], num_sample[s - 1], interval_scale[s - 1], propagation_range[s - 1],
                           G[s - 1], propagate_neighbors[s - 1], evaluate_neighbors[s - 1]) for s in (1, 2, 3)}


def patchmatch_stage(cfg: StageConfig, params, ref_feature, src_features, ref_proj, src_projs, depth_min, depth_max,
                     depth: Optional[np.ndarray], view_weights: Optional[np.ndarray], noise: Optional[np.ndarray] = None,
                     propa_offsets: Optional[np.ndarray] = None, eval_offsets: Optional[np.ndarray] = None,
                     trace: Optional[list] = None):
    """models/patchmatch.py:460-529.  Returns (depths list of [B,1,h,w], score [B,D,h,w], view_weights [B,N,h,w]).
    ``propa_offsets`` / ``eval_offsets`` may be passed in (the product computes them with MIOpen); when None they
    are computed here with the numpy conv.  ``trace`` (a list) receives one dict of intermediates per iteration."""
    prefix = f"patchmatch_{cfg.stage}"
    do_propagate_any = cfg.propagate_neighbors > 0 and not (cfg.stage == 1 and cfg.iterations == 1)
    if do_propagate_any and propa_offsets is None:
        propa_offsets = dilated_conv3x3(ref_feature, params[f"{prefix}.propa_conv.weight"],
                                        params[f"{prefix}.propa_conv.bias"], cfg.dilation)
    if eval_offsets is None:
        eval_offsets = dilated_conv3x3(ref_feature, params[f"{prefix}.eval_conv.weight"],
                                       params[f"{prefix}.eval_conv.bias"], cfg.dilation)
    ptab = propagation_table(cfg.propagate_neighbors, cfg.dilation) if do_propagate_any else None
    etab = evaluation_table(cfg.evaluate_neighbors, cfg.dilation)
    feature_weight = feature_weight_net(ref_feature, eval_offsets, etab, cfg.G, params, f"{prefix}.feature_weight_net")
    depth_sample = depth
    depths: List[np.ndarray] = []
    score = None
    for it in range(1, cfg.iterations + 1):
        is_inverse = cfg.stage == 1 and it == cfg.iterations
        depth_sample = depth_initialization(depth_sample, noise, depth_min, depth_max, cfg.num_sample,
                                            cfg.interval_scale)
        if cfg.propagate_neighbors > 0 and not (cfg.stage == 1 and it == cfg.iterations):
            depth_sample = propagation(depth_sample, propa_offsets, ptab)
        weight = depth_weight(depth_sample, depth_min, depth_max, eval_offsets, etab, cfg.interval_scale)
        fw = _c(feature_weight)
        for b in range(weight.shape[0]):  # weight = weight * feature_weight.unsqueeze(1); weight /= weight.sum(2)
            lib().pmo_weight_normalise(_p(weight[b]), _p(fw[b]), weight.shape[1], weight.shape[2], weight.shape[3] * weight.shape[4])
        ev = evaluation(ref_feature, src_features, ref_proj, src_projs, depth_sample, eval_offsets, etab, weight,
                        view_weights, is_inverse, cfg.G, params, f"{prefix}.evaluation")
        if trace is not None:
            rec = dict(ev)
            rec.update(depth_sample=depth_sample, weight=weight, feature_weight=feature_weight,
                       eval_offsets=eval_offsets, propa_offsets=propa_offsets)
            trace.append(rec)
        view_weights = ev["view_weights"]
        score = ev["score"]
        depth_sample = ev["depth"][:, None]
        depths.append(depth_sample)
    return depths, score, view_weights


def stage_projections(intrinsics: np.ndarray, extrinsics: np.ndarray, scale: float) -> np.ndarray:
    """models/net.py:225-229 -> proj [B,N,4,4]."""
    K = intrinsics.astype(np.float32).copy()
    K[:, :, :2] *= np.float32(scale)
    proj = extrinsics.astype(np.float32).copy()
    proj[:, :, :3, :4] = np.matmul(K, extrinsics[:, :, :3, :4].astype(np.float32))
    return proj


def nearest_up2(x: np.ndarray) -> np.ndarray:
    """F.interpolate(scale_factor=2, mode='nearest') (net.py:274-275): out[y,x] = in[y//2, x//2]."""
    return np.repeat(np.repeat(x, 2, axis=-2), 2, axis=-1)


def cascade(params, features: Sequence[Dict[int, np.ndarray]], intrinsics, extrinsics, depth_min, depth_max,
            noise: np.ndarray, configs: Optional[Dict[int, StageConfig]] = None, trace: Optional[dict] = None):
    """models/net.py:210-275 (between FeatureNet and Refinement).  features[i][stage] = [B,C,h,w] (index 0 = ref).
    Returns (stage-1 depth [B,1,H/2,W/2], stage-1 score [B,D,H/2,W/2], depth_patchmatch dict)."""
    configs = configs or default_stage_configs()
    depth_min = depth_min.astype(np.float32)
    depth_max = depth_max.astype(np.float32)
    depth, view_weights, score = None, None, None
    out: Dict[int, List[np.ndarray]] = {}
    scale = 0.125
    for stage in (3, 2, 1):
        proj = stage_projections(intrinsics, extrinsics, scale)
        scale *= 2.0
        tr = [] if trace is not None else None
        depths, score, view_weights = patchmatch_stage(
            configs[stage], params, features[0][stage], [f[stage] for f in features[1:]], proj[:, 0],
            [proj[:, i] for i in range(1, proj.shape[1])], depth_min, depth_max, depth, view_weights,
            noise=noise if stage == 3 else None, trace=tr)
        if trace is not None:
            trace[stage] = tr
        out[stage] = depths
        depth = depths[-1]
        if stage > 1:
            depth = nearest_up2(depth)
            view_weights = nearest_up2(view_weights)
    return depth, score, out


def confidence(score: np.ndarray, out_hw: Tuple[int, int]) -> Tuple[np.ndarray, np.ndarray]:
    """models/net.py:288-299 + module.py:184-196.  score [B,D,h,w] -> (confidence [B,H,W], depth_index [B,h,w])."""
    B, D, h, w = score.shape
    pad = np.pad(score, ((0, 0), (1, 2), (0, 0), (0, 0)))
    # 4 * avg_pool3d((4,1,1), stride 1): mean of the 4-window (sum, then /4), times 4
    win = (pad[:, 0:D] + pad[:, 1:D + 1] + pad[:, 2:D + 2] + pad[:, 3:D + 3]).astype(np.float32)
    sum4 = np.float32(4.0) * (win / np.float32(4.0))
    idx = (score * np.arange(D, dtype=np.float32).reshape(1, D, 1, 1)).sum(axis=1, dtype=np.float32)
    idx = np.clip(idx.astype(np.int64), 0, D - 1)  # .long() truncates toward zero; values are >= 0
    conf = np.take_along_axis(sum4, idx[:, None], axis=1)[:, 0]
    H, W = out_hw
    ys = np.minimum((np.arange(H) * (h / H)).astype(np.int64), h - 1)  # F.interpolate(mode='nearest') index rule
    xs = np.minimum((np.arange(W) * (w / W)).astype(np.int64), w - 1)
    return conf[:, ys][:, :, xs].astype(np.float32), idx
