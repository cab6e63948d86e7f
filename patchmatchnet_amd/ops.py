"""Tensor-level entry points of the HIP path: thin, checked wrappers over the C ABI (include/pmn_hip.h).

PyTorch is used here only as the owner of device memory and of the current HIP stream.  Every function requires
float32, contiguous tensors on a ROCm device and raises otherwise -- there is no CPU or eager fallback.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import PmnError, check


# ---- optional per-launch timing of the hot kernel (bench.py's roofline leg) --------------------------------------
# When enabled, every pmn_warp_correlate launch is bracketed by HIP events recorded on the launch stream (torch's
# current stream) and tagged with its ALGORITHMIC byte count B_alg = 4*h*w*[(1+N)*C + D + N + G*D].
_TIMING = None
_TIMING_ON = True


def enable_kernel_timing() -> None:
    global _TIMING, _TIMING_ON
    _TIMING, _TIMING_ON = [], True


def pause_kernel_timing(paused: bool) -> None:
    """Suspends / resumes recording without dropping what was recorded (an event pair costs ~10 us of stream time on
    ROCm, so bench.py samples every few steps instead of bracketing every launch)."""
    global _TIMING_ON
    _TIMING_ON = not paused


def disable_kernel_timing():
    """Stops recording and returns [(milliseconds, algorithmic_bytes, tag), ...] (synchronises once)."""
    global _TIMING
    rec, _TIMING = _TIMING, None
    if not rec:
        return []
    torch.cuda.synchronize()
    return [(s.elapsed_time(e), nbytes, tag) for s, e, nbytes, tag in rec]


TUNE_WINDOW_BYTES, TUNE_FLAGS, TUNE_WINDOW_BYTES_PIXELWISE, TUNE_ABLATE = 0, 1, 2, 3
TUNE_LANE_WINDOW_BYTES, TUNE_LANE_ABLATE, TUNE_LANE_WAVES_PER_SIMD = 4, 5, 6
TUNE_TILE_WINDOW_BYTES = 10
FLAG_WINDOWED, FLAG_NO_ROTATION, FLAG_STREAM_PIXELWISE, FLAG_STREAM_VIEWS, FLAG_WIN_V1, FLAG_TILE, FLAG_MFMA = 1, 2, 4, 8, 16, 32, 64
DEFAULT_FLAGS = 0  # the library default: streaming kernels for every launch (the fastest measured, profiles/README.md)


def set_tuning(key: int, value: int) -> None:
    """pmn_set_tuning of the EXPERIMENTAL build only (include/pmn_hip_experimental.h; PMN_EXPERIMENTAL=1): selects one of the
    research kernel families of pmn_warp_correlate / their window sizes; all families give bit-identical results
    (tests/test_gather_win.py).  The product library has neither the kernels nor the symbol."""
    if not _lib.experimental():
        raise PmnError("pmn_set_tuning exists only in libpmn_hip_experimental.so: build with `make -C patchmatchnet_amd/csrc "
                       "EXPERIMENTAL=1` and run with PMN_EXPERIMENTAL=1")
    check(_lib.lib().pmn_set_tuning(int(key), int(value)), "pmn_set_tuning")


# ---- opt-in activation-range check of the fp16-split entry points (PMN_CHECK_F16_DOMAIN=1; include/pmn_hip.h) --------------------------
import os as _os

F16_DOMAIN_CHECK = _os.environ.get("PMN_CHECK_F16_DOMAIN", "") == "1"
_F16_FLAGS = {}


def _f16_domain_probe(*tensors: Optional[torch.Tensor]) -> None:
    """PMN_CHECK_F16_DOMAIN=1: one pmn_check_f16_domain launch per input of an _f16s entry point (a pass over the tensor: a debugging
    aid for checkpoints / inputs other than the ones this was built on, not a default)."""
    if not F16_DOMAIN_CHECK:
        return
    for t in tensors:
        if t is None or not t.is_cuda or t.numel() == 0:
            continue
        flag = _F16_FLAGS.get(t.device)
        if flag is None:
            flag = _F16_FLAGS[t.device] = torch.zeros(1, dtype=torch.int32, device=t.device)
        with torch.cuda.device(t.device):
            check(_lib.lib().pmn_check_f16_domain(t.data_ptr(), t.numel(), flag.data_ptr(), _stream(t)), "pmn_check_f16_domain")


def f16_domain_check(reset: bool = True) -> None:
    """Raises PmnError when an activation handed to an fp16-split kernel since the last call was not finite or >= 65504 in magnitude
    (synchronises; only meaningful with PMN_CHECK_F16_DOMAIN=1)."""
    bad = [str(d) for d, f in _F16_FLAGS.items() if int(f.item()) != 0]
    if reset:
        for f in _F16_FLAGS.values():
            f.zero_()
    if bad:
        raise PmnError("an activation outside the fp16-split kernels' domain (|x| >= 65504 or not finite) reached pmn_*_f16s on " +
                       ", ".join(bad) + ": the outputs contain inf / NaN where an fp32 convolution is finite -- set f16_split = False on "
                       "FeatureNet / Refinement / PatchMatch for this checkpoint or input range (include/pmn_hip.h)")


def _dev(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise PmnError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise PmnError(f"{name}: tensor is on {t.device}; patchmatchnet_amd runs only on a ROCm GPU (no CPU fallback)")
    if t.dtype != torch.float32:
        raise PmnError(f"{name}: expected float32, got {t.dtype}")
    if not t.is_contiguous():
        raise PmnError(f"{name}: tensor must be contiguous")
    return t


def _dlast(t: torch.Tensor, name: str) -> torch.Tensor:
    """[B,D,h,w] tensor whose STORAGE is hypothesis-last [B,h,w,D] (what pmn_init_hypotheses / pmn_warp_correlate write and
    pmn_aggregate_regress reads); a planar tensor (tests, oracle data) is re-laid-out here."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float32:
        raise PmnError(f"{name}: expected a float32 tensor on a ROCm GPU (no CPU fallback)")
    if t.permute(0, 2, 3, 1).is_contiguous():
        return t
    return t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _host_f32(a: np.ndarray, n: int, name: str):
    a = np.ascontiguousarray(a, np.float32)
    if a.size != n:
        raise PmnError(f"{name}: expected {n} floats, got {a.size}")
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _host_i32(a: np.ndarray, n: int, name: str):
    a = np.ascontiguousarray(a, np.int32)
    if a.size != n:
        raise PmnError(f"{name}: expected {n} ints, got {a.size}")
    return a, a.ctypes.data_as(ctypes.c_void_p)


def nchw_to_nhwc(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[B,C,h,w] -> [B,h,w,C] (fresh tensor, or into ``out`` which may be a slice of a stacked buffer).  An input that is
    already channels-last in memory (an NCHW-shaped view of NHWC storage) is returned as a view, without a copy."""
    if out is None and isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and not x.is_contiguous():
        v = x.permute(0, 2, 3, 1)
        if v.is_contiguous():
            return v
        x = x.contiguous()
    _dev(x, "x")
    B, C, h, w = x.shape
    if out is None:
        out = torch.empty((B, h, w, C), dtype=torch.float32, device=x.device)
    else:
        _dev(out, "out")
        if tuple(out.shape) != (B, h, w, C):
            raise PmnError("nchw_to_nhwc: bad output shape")
    with torch.cuda.device(x.device):
        check(_lib.lib().pmn_nchw_to_nhwc(x.data_ptr(), out.data_ptr(), B, C, h, w, _stream(x)), "pmn_nchw_to_nhwc")
    return out


def stack_sources_nhwc(src_features: Sequence[torch.Tensor]) -> torch.Tensor:
    """List of N source feature maps [B,C,hs,ws] -> one channels-last buffer [N,B,hs,ws,C] (all maps one size; see
    ``stack_sources_padded`` for source views of different sizes)."""
    buf, sizes = stack_sources_padded(src_features)
    if any(sz != sizes[0] for sz in sizes):
        raise PmnError("all source feature maps of one stage must share a shape (stack_sources_padded handles mixed sizes)")
    return buf


def stack_sources_padded(src_features: Sequence[torch.Tensor]):
    """List of N source feature maps [B,C,hs_v,ws_v] -> (channels-last buffer [N,B,HS,WS,C] with HS x WS the LARGEST map and every
    smaller map in the top-left corner of its slice, zeros elsewhere; [(hs_v, ws_v)]).  The reference warps every source view at
    its own size (models/module.py:130-181: F.grid_sample un-normalises with the source map's size and pads with zeros), so a
    sample whose images differ in size is legal there; pmn_warp_correlate reads one size per launch, and zero padding IS
    grid_sample's padding -- the per-view scale is restored on the projection (``rescale_projection_rows``)."""
    if len(src_features) == 0:
        raise PmnError("at least one source view is required")
    B, C = src_features[0].shape[:2]
    sizes = [(int(f.shape[2]), int(f.shape[3])) for f in src_features]
    if any(tuple(f.shape[:2]) != (B, C) for f in src_features):
        raise PmnError("source feature maps of one stage must share batch size and channels")
    hs, ws = max(h for h, _ in sizes), max(w for _, w in sizes)
    mixed = any(sz != (hs, ws) for sz in sizes)
    alloc = torch.zeros if mixed else torch.empty
    buf = alloc((len(src_features), B, hs, ws, C), dtype=torch.float32, device=src_features[0].device)
    for i, f in enumerate(src_features):
        v = f.permute(0, 2, 3, 1)
        if sizes[i] != (hs, ws):
            _dev(f.contiguous(), "src_feature")
            buf[i][:, :sizes[i][0], :sizes[i][1], :].copy_(v)
        elif f.is_cuda and f.dtype == torch.float32 and v.is_contiguous():
            buf[i].copy_(v)  # an NCHW-shaped view of channels-last storage (the HIP FeatureNet's output): one plain copy, not a
            #                  transpose to NCHW and back (eval.py's encode-once path hands such views over, 5 per stage)
        else:
            nchw_to_nhwc(f.contiguous(), buf[i])
    return buf, sizes


def rescale_projection_rows(rel_proj: torch.Tensor, sizes, padded_hw) -> torch.Tensor:
    """Relative projections [B,N,4,4] for source maps that sit zero-padded in an HS x WS buffer: the kernel scales the projected
    position by (WS-1)/(w-1), the reference by (ws_v-1)/(w-1) (normalise with the reference map, un-normalise with the source map:
    models/module.py:170-181) -- rows 0 / 1 of view v are multiplied by (ws_v-1)/(WS-1) / (hs_v-1)/(HS-1).  Identity (same tensor)
    when every map has the padded size."""
    HS, WS = padded_hw
    if all(sz == (HS, WS) for sz in sizes):
        return rel_proj
    out = rel_proj.clone()
    for v, (h_v, w_v) in enumerate(sizes):
        if (h_v, w_v) != (HS, WS):
            out[:, v, 0, :] *= (w_v - 1) / (WS - 1)
            out[:, v, 1, :] *= (h_v - 1) / (HS - 1)
    return out.contiguous()


def _mlp_dev(t: torch.Tensor, name: str) -> torch.Tensor:
    _dev(t, name)
    if t.numel() != _lib.MLP_FLOATS:
        raise PmnError(f"{name}: expected a packed MLP block of {_lib.MLP_FLOATS} floats")
    return t


def feature_weight(ref_nhwc: torch.Tensor, eval_offsets: torch.Tensor, table: np.ndarray, mlp: torch.Tensor,
                   G: int) -> torch.Tensor:
    """FeatureWeightNet (reference models/patchmatch.py:603-624) -> [B,K,h,w]."""
    _dev(ref_nhwc, "ref_nhwc")
    _dev(eval_offsets, "eval_offsets")
    B, h, w, C = ref_nhwc.shape
    K = eval_offsets.shape[1] // 2
    if tuple(eval_offsets.shape) != (B, 2 * K, h, w):
        raise PmnError("feature_weight: eval_offsets must be [B,2K,h,w]")
    tab, tab_p = _host_i32(table, 2 * K, "table")
    _mlp_dev(mlp, "mlp")
    out = torch.empty((B, K, h, w), dtype=torch.float32, device=ref_nhwc.device)
    with torch.cuda.device(ref_nhwc.device):
        check(_lib.lib().pmn_feature_weight(ref_nhwc.data_ptr(), eval_offsets.data_ptr(), tab_p, mlp.data_ptr(), B, C, G,
                                            K, h, w,
                                            out.data_ptr(), _stream(out)), "pmn_feature_weight")
    return out


def init_hypotheses(noise: Optional[torch.Tensor], depth: Optional[torch.Tensor], depth_shift: int,
                    depth_min: torch.Tensor, depth_max: torch.Tensor, num_sample: int, interval_scale: float,
                    propa_offsets: Optional[torch.Tensor], table: Optional[np.ndarray], h: int, w: int
                    ) -> Tuple[torch.Tensor, torch.Tensor]:
    """DepthInitialization + Propagation (reference models/patchmatch.py:53-94, 115-124).

    Returns (depth_sample [B,D,h,w], xnorm [B,D,h,w] -- a hypothesis-last view: storage [B,h,w,D])."""
    _dev(depth_min, "depth_min")
    _dev(depth_max, "depth_max")
    B = depth_min.shape[0]
    dev = depth_min.device
    if noise is not None:
        _dev(noise, "noise")
        if tuple(noise.shape) != (B, 48, h, w):
            raise PmnError("init_hypotheses: noise must be [B,48,h,w]")
        D0 = 48
    else:
        if depth is None:
            raise PmnError("init_hypotheses: need noise or depth")
        _dev(depth, "depth")
        if tuple(depth.shape) != (B, 1, h >> depth_shift, w >> depth_shift):
            raise PmnError(f"init_hypotheses: depth must be [B,1,{h >> depth_shift},{w >> depth_shift}], "
                           f"got {tuple(depth.shape)}")
        D0 = num_sample
    K = 0
    tab_p = None
    tab = None
    if propa_offsets is not None:
        _dev(propa_offsets, "propa_offsets")
        K = propa_offsets.shape[1] // 2
        if tuple(propa_offsets.shape) != (B, 2 * K, h, w):
            raise PmnError("init_hypotheses: propa_offsets must be [B,2K,h,w]")
        tab, tab_p = _host_i32(table, 2 * K, "table")
    D = D0 + K
    ds = torch.empty((B, D, h, w), dtype=torch.float32, device=dev)
    xn = torch.empty((B, h, w, D), dtype=torch.float32, device=dev).permute(0, 3, 1, 2)
    with torch.cuda.device(dev):
        check(_lib.lib().pmn_init_hypotheses(_ptr(noise), _ptr(depth) if noise is None else None, depth_shift,
                                             depth_min.data_ptr(), depth_max.data_ptr(), num_sample,
                                             float(interval_scale), _ptr(propa_offsets), tab_p, K, B, h, w,
                                             ds.data_ptr(), xn.data_ptr(), _stream(ds)), "pmn_init_hypotheses")
    del tab
    return ds, xn


class SourceTable:
    """The source views of a sample as a DEVICE table of per-view addresses (pmn_warp_correlate_views) instead of one stacked
    [N,B,hs,ws,C] tensor: ``table`` int64 [N] on the device, entry v = data_ptr of view v's channels-last [B,hs,ws,C] map.  Duck-types
    the stacked tensor where only its shape / device are read.  The maps themselves must outlive the launch (the caller holds them)."""

    def __init__(self, table: torch.Tensor, shape) -> None:
        if table.dtype != torch.int64 or not table.is_cuda or table.numel() != int(shape[0]):
            raise PmnError("SourceTable: int64 device tensor with one address per source view")
        self.table, self.shape, self.device = table, tuple(int(x) for x in shape), table.device

    @staticmethod
    def image_in_place(im: torch.Tensor) -> bool:
        """True when pmn_stem_f16s_views can read this [B,3,H,W] image where it is (dense float32, 16-byte aligned)."""
        return bool(im.is_cuda and im.dtype == torch.float32 and im.is_contiguous() and im.data_ptr() % 16 == 0)

    @staticmethod
    def addresses(maps) -> list:
        """data_ptr of every view's map after checking it is a dense channels-last [B,hs,ws,C] float32 device tensor."""
        out = []
        for m in maps:
            if not (m.is_cuda and m.dtype == torch.float32 and m.is_contiguous()):
                raise PmnError("SourceTable: dense float32 channels-last device maps")
            out.append(m.data_ptr())
        return out


def warp_correlate(ref_nhwc: torch.Tensor, src_nhwc, rel_proj: torch.Tensor, depth_sample: torch.Tensor,
                   view_weights: Optional[torch.Tensor], vw_shift: int, similarity_mlp: torch.Tensor,
                   pixelwise_mlp: Optional[torch.Tensor], G: int, want_similarity: bool = False,
                   want_argmax: bool = False):
    """The fused warp + gather + group-correlation + view aggregation + SimilarityNet-MLP kernel.

    Returns (cost [B,D,h,w] (a hypothesis-last view: storage [B,h,w,D]), view_weights [B,N,h,w] (input passed through, or computed), argmax or None,
    aggregated similarity [B,G,D,h,w] or None)."""
    _dev(ref_nhwc, "ref_nhwc")
    table = src_nhwc if isinstance(src_nhwc, SourceTable) else None
    if table is None:
        _dev(src_nhwc, "src_nhwc")
    _dev(rel_proj, "rel_proj")
    _dev(depth_sample, "depth_sample")
    B, h, w, C = ref_nhwc.shape
    N, Bs, hs, ws, Cs = src_nhwc.shape
    D = depth_sample.shape[1]
    if Bs != B or Cs != C or tuple(depth_sample.shape) != (B, D, h, w) or tuple(rel_proj.shape) != (B, N, 4, 4):
        raise PmnError("warp_correlate: inconsistent shapes")
    dev = ref_nhwc.device
    sim_p = _mlp_dev(similarity_mlp, "similarity_mlp").data_ptr()
    pix_p, vw_out, argmax = None, None, None
    if view_weights is not None:
        _dev(view_weights, "view_weights")
        if tuple(view_weights.shape) != (B, N, h >> vw_shift, w >> vw_shift):
            raise PmnError("Patchmatch Evaluation: Different number of images and view weights")
    else:
        if pixelwise_mlp is None:
            raise PmnError("warp_correlate: pixelwise_mlp required when view_weights is None")
        pix_p = _mlp_dev(pixelwise_mlp, "pixelwise_mlp").data_ptr()
        vw_out = torch.empty((B, N, h, w), dtype=torch.float32, device=dev)
        if want_argmax:
            argmax = torch.empty((B, N, h, w), dtype=torch.int32, device=dev)
    cost = torch.empty((B, h, w, D), dtype=torch.float32, device=dev).permute(0, 3, 1, 2)  # hypothesis-last storage
    sim = torch.empty((B, G, D, h, w), dtype=torch.float32, device=dev) if want_similarity else None
    with torch.cuda.device(dev):
        if _TIMING is not None and _TIMING_ON:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        fn = _lib.lib().pmn_warp_correlate if table is None else _lib.lib().pmn_warp_correlate_views
        check(fn(ref_nhwc.data_ptr(), (src_nhwc if table is None else table.table).data_ptr(), rel_proj.data_ptr(),
                 depth_sample.data_ptr(), _ptr(view_weights), vw_shift, sim_p, pix_p, B, N, C,
                 G, D, h, w, hs, ws, cost.data_ptr(), _ptr(vw_out), _ptr(argmax), _ptr(sim),
                 _stream(cost)), "pmn_warp_correlate")
        if _TIMING is not None and _TIMING_ON:
            ev1.record()
            _TIMING.append((ev0, ev1, 4 * B * h * w * ((1 + N) * C + D + N + G * D),
                            f"C{C}_D{D}_{h}x{w}_N{N}_{'vw' if view_weights is not None else 'pixelwise'}"))
    return cost, (view_weights if view_weights is not None else vw_out), argmax, sim


def aggregate_regress(cost: torch.Tensor, depth_sample: torch.Tensor, xnorm: torch.Tensor, feature_weight_: torch.Tensor,
                      eval_offsets: torch.Tensor, table: np.ndarray, interval_scale: float, is_inverse: bool
                      ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Adaptive spatial aggregation + softmax + regression -> (score [B,D,h,w], depth [B,h,w])."""
    for n, t in (("depth_sample", depth_sample), ("feature_weight", feature_weight_), ("eval_offsets", eval_offsets)):
        _dev(t, n)
    cost, xnorm = _dlast(cost, "cost"), _dlast(xnorm, "xnorm")
    B, D, h, w = cost.shape
    K = feature_weight_.shape[1]
    if tuple(depth_sample.shape) != (B, D, h, w) or tuple(xnorm.shape) != (B, D, h, w) or \
            tuple(feature_weight_.shape) != (B, K, h, w) or tuple(eval_offsets.shape) != (B, 2 * K, h, w):
        raise PmnError("aggregate_regress: inconsistent shapes")
    tab, tab_p = _host_i32(table, 2 * K, "table")
    score = torch.empty((B, D, h, w), dtype=torch.float32, device=cost.device)
    depth = torch.empty((B, h, w), dtype=torch.float32, device=cost.device)
    with torch.cuda.device(cost.device):
        check(_lib.lib().pmn_aggregate_regress(cost.data_ptr(), depth_sample.data_ptr(), xnorm.data_ptr(),
                                               feature_weight_.data_ptr(), eval_offsets.data_ptr(), tab_p, K,
                                               float(interval_scale), 1 if is_inverse else 0, B, D, h, w,
                                               score.data_ptr(), depth.data_ptr(), _stream(cost)),
              "pmn_aggregate_regress")
    return score, depth


def confidence(score: torch.Tensor, H: int, W: int, want_index: bool = False):
    """Photometric confidence (reference models/net.py:288-299) -> ([B,H,W], depth_index [B,h,w] int32 or None)."""
    _dev(score, "score")
    B, D, h, w = score.shape
    conf = torch.empty((B, H, W), dtype=torch.float32, device=score.device)
    idx = torch.empty((B, h, w), dtype=torch.int32, device=score.device) if want_index else None
    with torch.cuda.device(score.device):
        check(_lib.lib().pmn_confidence(score.data_ptr(), B, D, h, w, H, W, conf.data_ptr(), _ptr(idx), _stream(score)),
              "pmn_confidence")
    return conf, idx


def relative_projection(src_projs: Sequence[torch.Tensor], ref_proj: torch.Tensor) -> torch.Tensor:
    """src_proj @ inverse(ref_proj) for every source view (reference models/module.py:148) -> [B,N,4,4]."""
    inv = torch.inverse(ref_proj)
    return torch.matmul(torch.stack(list(src_projs), dim=1), inv.unsqueeze(1)).contiguous()


def differentiable_warping(src_fea: torch.Tensor, src_proj: torch.Tensor, ref_proj: torch.Tensor,
                           depth_samples: torch.Tensor) -> torch.Tensor:
    """Drop-in for reference models/module.py:130-181 (inference): [B,C,Hs,Ws],[B,4,4],[B,4,4],[B,D,H,W] -> [B,C,D,H,W]."""
    src_fea = _dev(src_fea.contiguous(), "src_fea")
    depth_samples = _dev(depth_samples.contiguous(), "depth_samples")
    B, C, hs, ws = src_fea.shape
    _, D, h, w = depth_samples.shape
    proj = torch.matmul(src_proj, torch.inverse(ref_proj)).contiguous()
    _dev(proj, "proj")
    out = torch.empty((B, C, D, h, w), dtype=torch.float32, device=src_fea.device)
    with torch.cuda.device(src_fea.device):
        check(_lib.lib().pmn_differentiable_warping(src_fea.data_ptr(), proj.data_ptr(), depth_samples.data_ptr(), B, C,
                                                    D, h, w, hs, ws, out.data_ptr(), _stream(out)),
              "pmn_differentiable_warping")
    return out


def stage_projections(intrinsics: torch.Tensor, extrinsics: torch.Tensor, nstages: int = 3, scale0: float = 0.125
                      ) -> torch.Tensor:
    """pmn_stage_projections: intrinsics [B,V,3,3], extrinsics [B,V,4,4] -> relative projections [nstages,B,V-1,4,4]
    (stage index 0 = coarsest, scale0; reference models/net.py:221-232 + models/module.py:148)."""
    intrinsics = _dev(intrinsics.float().contiguous(), "intrinsics")
    extrinsics = _dev(extrinsics.float().contiguous(), "extrinsics")
    B, V = intrinsics.shape[:2]
    rel = torch.empty((nstages, B, V - 1, 4, 4), dtype=torch.float32, device=intrinsics.device)
    with torch.cuda.device(rel.device):
        check(_lib.lib().pmn_stage_projections(intrinsics.data_ptr(), extrinsics.data_ptr(), B, V, nstages, float(scale0),
                                               rel.data_ptr(), _stream(rel)), "pmn_stage_projections")
    return rel


def normalize_depth(depth: torch.Tensor, depth_min: torch.Tensor, depth_max: torch.Tensor) -> torch.Tensor:
    """pmn_normalize_depth: (depth - depth_min[b]) / (depth_max[b] - depth_min[b]) for depth [B, ...] (reference models/net.py:104-106),
    the bits of the torch expression; with it the forward launches nothing but this library's kernels (launch plans, plan.py)."""
    depth = _dev(depth.contiguous(), "depth")
    lo, hi = _dev(depth_min.float().contiguous(), "depth_min"), _dev(depth_max.float().contiguous(), "depth_max")
    B = depth.shape[0]
    if lo.numel() != B or hi.numel() != B:
        raise PmnError("normalize_depth: depth_min / depth_max must hold one value per batch element")
    out = torch.empty_like(depth)
    with torch.cuda.device(depth.device):
        check(_lib.lib().pmn_normalize_depth(depth.data_ptr(), lo.data_ptr(), hi.data_ptr(), B, depth.numel() // B, out.data_ptr(),
                                             _stream(out)), "pmn_normalize_depth")
    return out


def conv2d(x: torch.Tensor, weights: torch.Tensor, shift: torch.Tensor, cout: int, K: int, stride: int = 1, pad: int = 0,
           dil: int = 1, relu: bool = False, up: Optional[torch.Tensor] = None, in_nchw: bool = False,
           out_nchw: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pmn_conv2d: x [N,H,W,cin] (or [N,cin,H,W] with in_nchw) -> [N,Ho,Wo,cout] (or [N,cout,Ho,Wo] with out_nchw);
    ``weights`` / ``shift`` from params.pack_conv (device tensors); ``up`` [N,Ho/2,Wo/2,cout] is up-sampled x2 and added."""
    _dev(x, "x")
    _dev(weights, "weights")
    _dev(shift, "shift")
    if in_nchw:
        N, cin, H, W = x.shape
    else:
        N, H, W, cin = x.shape
    if tuple(weights.shape[:3]) != (K, K, cin) or shift.numel() != weights.shape[3]:
        raise PmnError("conv2d: packed weights do not match the input")
    Ho = (H + 2 * pad - dil * (K - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (K - 1) - 1) // stride + 1
    shape = (N, cout, Ho, Wo) if out_nchw else (N, Ho, Wo, cout)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
    else:
        _dev(out, "out")
        if tuple(out.shape) != shape:
            raise PmnError("conv2d: bad `out` shape")
    up_h = up_w = 0
    if up is not None:
        _dev(up, "up")
        if tuple(up.shape) != (N, Ho // 2, Wo // 2, cout) or Ho % 2 or Wo % 2:
            raise PmnError("conv2d: `up` must be [N,Ho/2,Wo/2,cout]")
        up_h, up_w = up.shape[1], up.shape[2]
    with torch.cuda.device(x.device):
        check(_lib.lib().pmn_conv2d(x.data_ptr(), weights.data_ptr(), shift.data_ptr(), _ptr(up), out.data_ptr(), N, H, W,
                                    cin, cout, K, stride, pad, dil, 1 if relu else 0, 1 if in_nchw else 0,
                                    1 if out_nchw else 0, up_h, up_w, _stream(x)), "pmn_conv2d")
    return out


def fpn_tail(x: torch.Tensor, up: torch.Tensor, w_in: torch.Tensor, b_in: torch.Tensor, w_out: torch.Tensor) -> torch.Tensor:
    """pmn_fpn_tail: output3(bilinear_x2(up) + inner2(x)) (reference models/net.py:64-67); x [N,H,W,16], up [N,H/2,W/2,64]
    -> [N,H,W,16]; weights in params.pack_conv layout."""
    for n_, t_ in (("x", x), ("up", up), ("w_in", w_in), ("b_in", b_in), ("w_out", w_out)):
        _dev(t_, n_)
    N, H, W, cin = x.shape
    cmid, cout = up.shape[3], w_out.shape[3]
    if tuple(up.shape) != (N, H // 2, W // 2, cmid) or tuple(w_in.shape) != (1, 1, cin, cmid) or \
            tuple(w_out.shape) != (1, 1, cmid, cout):
        raise PmnError("fpn_tail: inconsistent shapes")
    out = torch.empty((N, H, W, cout), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.lib().pmn_fpn_tail(x.data_ptr(), up.data_ptr(), w_in.data_ptr(), b_in.data_ptr(), w_out.data_ptr(),
                                      out.data_ptr(), N, H, W, cin, cmid, cout, _stream(x)), "pmn_fpn_tail")
    return out


MFMA_CONV_SHAPES = {(64, 64, 3, 1), (32, 32, 3, 1), (32, 64, 5, 2), (16, 32, 5, 2)}  # (cin, cout, K, stride)
MFMA_HEAD_SHAPES = {(64, 2), (32, 4), (16, 6)}  # (cin, dilation) of the planar offset-head form, cout <= 64


def conv2d_mfma(x: torch.Tensor, weights: torch.Tensor, shift: torch.Tensor, K: int, stride: int = 1, pad: int = 0,
                relu: bool = True) -> torch.Tensor:
    """pmn_conv2d_mfma (planar = 0): conv + folded-BN shift + ReLU as fp32 implicit GEMM on the matrix cores; x [N,H,W,cin]
    channels-last, weights from params.pack_conv_mfma ([K*K, cin/8, cout/32, 64, 4]) -> [N,Ho,Wo,cout]."""
    for n_, t_ in (("x", x), ("weights", weights), ("shift", shift)):
        _dev(t_, n_)
    N, H, W, cin = x.shape
    cout = shift.shape[0]
    if tuple(weights.shape) != (K * K, cin // 8, cout // 32, 64, 4):
        raise PmnError("conv2d_mfma: weights are not in pack_conv_mfma layout for this input")
    Ho, Wo = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    out = torch.empty((N, Ho, Wo, cout), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.lib().pmn_conv2d_mfma(x.data_ptr(), weights.data_ptr(), shift.data_ptr(), out.data_ptr(), None, N, H, W,
                                         cin, cout, cout, K, stride, pad, 1, 1 if relu else 0, 0, _stream(x)),
              "pmn_conv2d_mfma")
    return out


def conv3x3_wino(x: torch.Tensor, weights: torch.Tensor, shift: torch.Tensor, relu: bool = True) -> torch.Tensor:
    """pmn_conv3x3_wino: 3x3 / stride 1 / padding 1 convolution with cin == cout in {16,32,64} + folded-BN shift + ReLU in
    Winograd F(2x2,3x3) form on the matrix cores; x [N,H,W,C] channels-last, weights from params.pack_conv_wino."""
    if not _lib.experimental():
        raise PmnError("conv3x3_wino: research build only (make -C patchmatchnet_amd/csrc EXPERIMENTAL=1, PMN_EXPERIMENTAL=1)")
    for n_, t_ in (("x", x), ("weights", weights), ("shift", shift)):
        _dev(t_, n_)
    N, H, W, C = x.shape
    if tuple(weights.shape) != (C // 16, 16, C // 16, 64, 4) or tuple(shift.shape) != (C,):
        raise PmnError("conv3x3_wino: weights are not in pack_conv_wino layout for this input")
    out = torch.empty((N, H, W, C), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.lib().pmn_conv3x3_wino(x.data_ptr(), weights.data_ptr(), shift.data_ptr(), out.data_ptr(), N, H, W, C,
                                          1 if relu else 0, _stream(x)), "pmn_conv3x3_wino")
    return out


def conv5x5s2_wino(x: torch.Tensor, weights: torch.Tensor, shift: torch.Tensor, relu: bool = True) -> torch.Tensor:
    """pmn_conv5x5s2_wino: 5x5 / stride 2 / padding 2 convolution + folded-BN shift + ReLU in phase-decomposed Winograd form on
    the matrix cores; x [N,H,W,cin] channels-last, weights from params.pack_conv5x5s2_wino -> [N,(H-1)//2+1,(W-1)//2+1,cout]."""
    if not _lib.experimental():
        raise PmnError("conv5x5s2_wino: research build only (make -C patchmatchnet_amd/csrc EXPERIMENTAL=1, PMN_EXPERIMENTAL=1)")
    for n_, t_ in (("x", x), ("weights", weights), ("shift", shift)):
        _dev(t_, n_)
    N, H, W, cin = x.shape
    cout = shift.shape[0]
    if tuple(weights.shape) != (cin // 8, 49, cout // 16, 64, 2):
        raise PmnError("conv5x5s2_wino: weights are not in pack_conv5x5s2_wino layout for this input")
    out = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, cout), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.lib().pmn_conv5x5s2_wino(x.data_ptr(), weights.data_ptr(), shift.data_ptr(), out.data_ptr(), N, H, W, cin,
                                            cout, 1 if relu else 0, _stream(x)), "pmn_conv5x5s2_wino")
    return out


F16S_SHAPES = {(3, 1, 16, 16), (3, 1, 32, 32), (3, 1, 64, 64), (5, 2, 8, 16), (5, 2, 16, 32), (5, 2, 32, 64)}  # (k, stride, cin, cout)


def conv2d_f16s(x: torch.Tensor, weights: torch.Tensor, shift: torch.Tensor, k: int, stride: int, relu: bool = True) -> torch.Tensor:
    """pmn_conv2d_f16s: convolution (padding k // 2) + folded-BN shift + ReLU on the FP16 matrix cores with split (hi + lo/2048)
    operands -- fp32-convolution accuracy; x [N,H,W,cin] channels-last float32, weights float16 from params.pack_conv_f16s ->
    [N,(H-1)//stride+1,(W-1)//stride+1,cout] float32."""
    _dev(x, "x")
    _f16_domain_probe(x)
    _dev(shift, "shift")
    if not isinstance(weights, torch.Tensor) or not weights.is_cuda or weights.dtype != torch.float16 or not weights.is_contiguous():
        raise PmnError("conv2d_f16s: weights must be a contiguous float16 tensor on a ROCm GPU (params.pack_conv_f16s)")
    N, H, W, cin = x.shape
    cout = shift.shape[0]
    if (k, stride, cin, cout) not in F16S_SHAPES:
        raise PmnError(f"conv2d_f16s: unsupported layer (k={k}, stride={stride}, cin={cin}, cout={cout})")
    from . import params as _params
    cc = _params.f16s_chunk(cin, k)
    if tuple(weights.shape) != (cin // cc, (k * k * (cc // 8) + 3) // 4, cout // 16, 2, 64, 8):
        raise PmnError("conv2d_f16s: weights are not in pack_conv_f16s layout for this input")
    out = torch.empty((N, (H - 1) // stride + 1, (W - 1) // stride + 1, cout), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.lib().pmn_conv2d_f16s(x.data_ptr(), weights.data_ptr(), shift.data_ptr(), out.data_ptr(), N, H, W, cin, cout, k,
                                         stride, 1 if relu else 0, _stream(x)), "pmn_conv2d_f16s")
    return out


def conv2d_f16s_pair(x: torch.Tensor, weights_a: torch.Tensor, shift_a: torch.Tensor, weights_b: torch.Tensor, shift_b: torch.Tensor,
                     relu: bool = True) -> torch.Tensor:
    """pmn_conv2d_f16s_pair: two consecutive 3x3 / stride-1 / 16 -> 16 ConvBnReLU layers in one launch (FeatureNet conv3 + conv4), the
    intermediate map kept in LDS; x [N,H,W,16] channels-last float32, (weights, shift) pairs from params.pack_conv_f16s.  Bit-identical
    to conv2d_f16s(conv2d_f16s(x, a...), b...)."""
    _dev(x, "x")
    _f16_domain_probe(x)
    N, H, W, C = x.shape
    for w_, s_ in ((weights_a, shift_a), (weights_b, shift_b)):
        _dev(s_, "shift")
        if not isinstance(w_, torch.Tensor) or not w_.is_cuda or w_.dtype != torch.float16 or not w_.is_contiguous() or \
                tuple(w_.shape) != (1, 5, 1, 2, 64, 8) or s_.numel() != 16:
            raise PmnError("conv2d_f16s_pair: weights must be params.pack_conv_f16s of a (3, 1, 16, 16) layer on a ROCm GPU")
    if C != 16:
        raise PmnError("conv2d_f16s_pair: 16-channel layers only")
    out = torch.empty((N, H, W, 16), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.lib().pmn_conv2d_f16s_pair(x.data_ptr(), weights_a.data_ptr(), shift_a.data_ptr(), weights_b.data_ptr(), shift_b.data_ptr(),
                                              out.data_ptr(), N, H, W, 16, 1 if relu else 0, _stream(x)), "pmn_conv2d_f16s_pair")
    return out


def pointwise_split_mfma(x: torch.Tensor, weights: torch.Tensor, shift: torch.Tensor, cout: int, ca: int):
    """pmn_conv2d_mfma, 1x1 form: out = x @ W + shift on the matrix cores with the output channels split between two
    channels-last tensors (the 1/8-resolution level of the folded FPN head); x [N,H,W,64], weights from
    params.pack_conv_mfma of the [cout,64,1,1] filter -> ([N,H,W,ca], [N,H,W,cout-ca])."""
    for n_, t_ in (("x", x), ("weights", weights), ("shift", shift)):
        _dev(t_, n_)
    N, H, W, cin = x.shape
    coutp = shift.shape[0]
    if tuple(weights.shape) != (1, cin // 8, coutp // 32, 64, 4) or not 0 < ca < cout <= coutp:
        raise PmnError("pointwise_split_mfma: weights are not in pack_conv_mfma layout for this input")
    out_a = torch.empty((N, H, W, ca), dtype=torch.float32, device=x.device)
    out_b = torch.empty((N, H, W, cout - ca), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.lib().pmn_conv2d_mfma(x.data_ptr(), weights.data_ptr(), shift.data_ptr(), out_a.data_ptr(), out_b.data_ptr(),
                                         N, H, W, cin, cout, ca, 1, 1, 0, 1, 0, 0, _stream(x)), "pmn_conv2d_mfma")
    return out_a, out_b


def offset_heads_mfma(x: torch.Tensor, weights: torch.Tensor, shift: torch.Tensor, cout: int, ca: int, dil: int):
    """pmn_conv2d_mfma (planar = 1): the offset heads of one stage (propa_conv rows first, then eval_conv; reference
    models/patchmatch.py:288-311) as one dilated 3x3 convolution with bias; x [N,H,W,cin] channels-last, weights from
    params.pack_conv_mfma of the row-concatenated filters -> ([N,ca,H,W], [N,cout-ca,H,W] or None) planar."""
    for n_, t_ in (("x", x), ("weights", weights), ("shift", shift)):
        _dev(t_, n_)
    N, H, W, cin = x.shape
    coutp = shift.shape[0]
    if tuple(weights.shape) != (9, cin // 8, coutp // 32, 64, 4) or not 0 < ca <= cout <= coutp:
        raise PmnError("offset_heads_mfma: weights are not in pack_conv_mfma layout for this input")
    out_a = torch.empty((N, ca, H, W), dtype=torch.float32, device=x.device)
    out_b = torch.empty((N, cout - ca, H, W), dtype=torch.float32, device=x.device) if ca < cout else None
    with torch.cuda.device(x.device):
        check(_lib.lib().pmn_conv2d_mfma(x.data_ptr(), weights.data_ptr(), shift.data_ptr(), out_a.data_ptr(), _ptr(out_b), N,
                                         H, W, cin, cout, ca, 3, 1, dil, dil, 0, 1, _stream(x)), "pmn_conv2d_mfma")
    return out_a, out_b


F16S_HEAD_SHAPES = {(64, 2), (32, 4), (16, 6)}  # (input channels, dilation) of the reference's stages 3, 2, 1


def offset_heads_f16s(x: torch.Tensor, weights: torch.Tensor, shift: torch.Tensor, cout: int, ca: int, dil: int):
    """pmn_offset_heads_f16s: the offset heads of one stage (propa_conv rows first, then eval_conv; reference
    models/patchmatch.py:288-311) as one dilated 3x3 convolution with bias on the fp16 matrix cores (split operands);
    x [N,H,W,cin] channels-last, weights / shift from params.pack_offset_heads_f16s -> ([N,ca,H,W], [N,cout-ca,H,W] or None) planar."""
    _dev(x, "x")
    _f16_domain_probe(x)
    _dev(shift, "shift")
    N, H, W, cin = x.shape
    coutp = shift.shape[0]
    if not isinstance(weights, torch.Tensor) or not weights.is_cuda or weights.dtype != torch.float16 or not weights.is_contiguous() \
            or tuple(weights.shape) != (cin // 16, (9 * 2 + 3) // 4, coutp // 16, 2, 64, 8) or not 0 < ca <= cout <= coutp \
            or coutp != (cout + 15) // 16 * 16:
        raise PmnError("offset_heads_f16s: weights are not in pack_offset_heads_f16s layout for this input")
    out_a = torch.empty((N, ca, H, W), dtype=torch.float32, device=x.device)
    out_b = torch.empty((N, cout - ca, H, W), dtype=torch.float32, device=x.device) if ca < cout else None
    with torch.cuda.device(x.device):
        check(_lib.lib().pmn_offset_heads_f16s(x.data_ptr(), weights.data_ptr(), shift.data_ptr(), out_a.data_ptr(), _ptr(out_b), N,
                                               H, W, cin, cout, ca, dil, _stream(x)), "pmn_offset_heads_f16s")
    return out_a, out_b


def fpn_level(x: torch.Tensor, u: Optional[torch.Tensor], w: torch.Tensor, b: torch.Tensor, ca: int):
    """pmn_fpn_level: one level of the folded FPN head, out = bilinear_x2(u) + b + x @ w (reference models/net.py:57-67 with
    the 1x1 convolutions composed, params.fold_fpn).  x [N,H,W,cin], u [N,H/2,W/2,cout] or None, w [cin,cout], b [cout]
    -> (out_a [N,H,W,ca], out_b [N,H,W,cout-ca] or None)."""
    for n_, t_ in (("x", x), ("w", w), ("b", b)):
        _dev(t_, n_)
    N, H, W, cin = x.shape
    cout = w.shape[1]
    if tuple(w.shape) != (cin, cout) or tuple(b.shape) != (cout,) or not 0 < ca <= cout:
        raise PmnError("fpn_level: inconsistent shapes")
    if u is not None:
        _dev(u, "u")
        if tuple(u.shape) != (N, H // 2, W // 2, cout) or H % 2 or W % 2:
            raise PmnError("fpn_level: `u` must be [N,H/2,W/2,cout]")
    out_a = torch.empty((N, H, W, ca), dtype=torch.float32, device=x.device)
    out_b = torch.empty((N, H, W, cout - ca), dtype=torch.float32, device=x.device) if ca < cout else None
    with torch.cuda.device(x.device):
        check(_lib.lib().pmn_fpn_level(x.data_ptr(), _ptr(u), w.data_ptr(), b.data_ptr(), out_a.data_ptr(), _ptr(out_b),
                                       N, H, W, cin, cout, ca, _stream(x)), "pmn_fpn_level")
    return out_a, out_b


def refine_front(img: torch.Tensor, t2: torch.Tensor, w0: torch.Tensor, s0: torch.Tensor, wd: torch.Tensor,
                 sd: torch.Tensor) -> torch.Tensor:
    """pmn_refine_front: cat(relu(bn(deconv(t2))), conv0(img)) of Refinement (reference models/net.py:110-117); img [B,3,H,W],
    t2 [B,H/2,W/2,8] channels-last -> [B,H,W,16]."""
    for n_, t_ in (("img", img), ("t2", t2), ("w0", w0), ("s0", s0), ("wd", wd), ("sd", sd)):
        _dev(t_, n_)
    B, c, H, W = img.shape
    if c != 3 or H % 2 or W % 2 or tuple(t2.shape) != (B, H // 2, W // 2, 8) or tuple(w0.shape) != (3, 3, 3, 8) or \
            tuple(wd.shape) != (3, 3, 8, 8):
        raise PmnError("refine_front: inconsistent shapes")
    out = torch.empty((B, H, W, 16), dtype=torch.float32, device=img.device)
    with torch.cuda.device(img.device):
        check(_lib.lib().pmn_refine_front(img.data_ptr(), t2.data_ptr(), w0.data_ptr(), s0.data_ptr(), wd.data_ptr(),
                                          sd.data_ptr(), out.data_ptr(), B, H, W, _stream(img)), "pmn_refine_front")
    return out


def refine_tail(x16: torch.Tensor, w3: torch.Tensor, s3: torch.Tensor, wr: torch.Tensor, dnorm: torch.Tensor,
                depth_min: torch.Tensor, depth_max: torch.Tensor) -> torch.Tensor:
    """pmn_refine_tail: (nearest_x2(dnorm) + res(conv3(x16))) * (depth_max - depth_min) + depth_min (reference
    models/net.py:117-122); x16 [B,H,W,16], dnorm [B,1,H/2,W/2], depth_min / depth_max [B] -> [B,1,H,W]."""
    for n_, t_ in (("x16", x16), ("w3", w3), ("s3", s3), ("wr", wr), ("dnorm", dnorm), ("depth_min", depth_min),
                   ("depth_max", depth_max)):
        _dev(t_, n_)
    B, H, W, c = x16.shape
    if c != 16 or H % 2 or W % 2 or tuple(dnorm.shape) != (B, 1, H // 2, W // 2) or tuple(w3.shape) != (2, 3, 3, 16, 4) or \
            tuple(wr.shape) != (3, 3, 8) or depth_min.numel() != B or depth_max.numel() != B:
        raise PmnError("refine_tail: inconsistent shapes")
    out = torch.empty((B, 1, H, W), dtype=torch.float32, device=x16.device)
    with torch.cuda.device(x16.device):
        check(_lib.lib().pmn_refine_tail(x16.data_ptr(), w3.data_ptr(), s3.data_ptr(), wr.data_ptr(), dnorm.data_ptr(),
                                         depth_min.data_ptr(), depth_max.data_ptr(), out.data_ptr(), B, H, W, _stream(x16)),
              "pmn_refine_tail")
    return out


def refine_fused(img: torch.Tensor, t2: torch.Tensor, w0: torch.Tensor, s0: torch.Tensor, wd: torch.Tensor, sd: torch.Tensor,
                 w3a: torch.Tensor, s3: torch.Tensor, wr: torch.Tensor, dnorm: torch.Tensor, depth_min: torch.Tensor,
                 depth_max: torch.Tensor) -> torch.Tensor:
    """pmn_refine_fused: refine_front + refine_tail in one launch, conv3 on the fp16 matrix cores with split operands (reference
    models/net.py:110-122); img [B,3,H,W], t2 [B,H/2,W/2,8], w3a = params.pack_refine_conv3_f16s (float16 [5,2,64,8]), dnorm
    [B,1,H/2,W/2], depth_min / depth_max [B] -> [B,1,H,W]."""
    for n_, t_ in (("img", img), ("t2", t2), ("w0", w0), ("s0", s0), ("wd", wd), ("sd", sd), ("s3", s3), ("wr", wr), ("dnorm", dnorm),
                   ("depth_min", depth_min), ("depth_max", depth_max)):
        _dev(t_, n_)
    if not isinstance(w3a, torch.Tensor) or not w3a.is_cuda or w3a.dtype != torch.float16 or tuple(w3a.shape) != (5, 2, 64, 8) \
            or not w3a.is_contiguous():
        raise PmnError("refine_fused: w3a must be the float16 [5,2,64,8] tensor of params.pack_refine_conv3_f16s on a ROCm GPU")
    B, c, H, W = img.shape
    if c != 3 or H % 2 or W % 2 or tuple(t2.shape) != (B, H // 2, W // 2, 8) or tuple(dnorm.shape) != (B, 1, H // 2, W // 2) or \
            tuple(w0.shape) != (3, 3, 3, 8) or tuple(wd.shape) != (3, 3, 8, 8) or tuple(wr.shape) != (3, 3, 8) or s3.numel() != 8 or \
            depth_min.numel() != B or depth_max.numel() != B:
        raise PmnError("refine_fused: inconsistent shapes")
    _f16_domain_probe(img, t2)
    out = torch.empty((B, 1, H, W), dtype=torch.float32, device=img.device)
    with torch.cuda.device(img.device):
        check(_lib.lib().pmn_refine_fused(img.data_ptr(), t2.data_ptr(), w0.data_ptr(), s0.data_ptr(), wd.data_ptr(), sd.data_ptr(),
                                          w3a.data_ptr(), s3.data_ptr(), wr.data_ptr(), dnorm.data_ptr(), depth_min.data_ptr(),
                                          depth_max.data_ptr(), out.data_ptr(), B, H, W, _stream(img)), "pmn_refine_fused")
    return out


def deconv3x3s2(x: torch.Tensor, weights: torch.Tensor, shift: torch.Tensor, relu: bool = True) -> torch.Tensor:
    """pmn_deconv3x3s2: ConvTranspose2d(k3,s2,p1,op1) + folded BN + ReLU; x [N,Hi,Wi,8] -> [N,2Hi,2Wi,8]."""
    _dev(x, "x")
    _dev(weights, "weights")
    _dev(shift, "shift")
    N, Hi, Wi, cin = x.shape
    cout = weights.shape[3]
    out = torch.empty((N, 2 * Hi, 2 * Wi, cout), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.lib().pmn_deconv3x3s2(x.data_ptr(), weights.data_ptr(), shift.data_ptr(), out.data_ptr(), N, Hi, Wi, cin,
                                         cout, 1 if relu else 0, _stream(x)), "pmn_deconv3x3s2")
    return out


def stem(img: torch.Tensor, w0: torch.Tensor, s0: torch.Tensor, w1: torch.Tensor, s1: torch.Tensor,
         out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pmn_stem: FeatureNet conv0 + conv1 fused; img [N,3,H,W] -> [N,H,W,8] (optionally into ``out``)."""
    for n_, t_ in (("img", img), ("w0", w0), ("s0", s0), ("w1", w1), ("s1", s1)):
        _dev(t_, n_)
    N, c, H, W = img.shape
    if c != 3 or tuple(w0.shape) != (3, 3, 3, 8) or tuple(w1.shape) != (3, 3, 8, 8):
        raise PmnError("stem: expects a 3-channel image and 3->8->8 weights")
    if out is None:
        out = torch.empty((N, H, W, 8), dtype=torch.float32, device=img.device)
    else:
        _dev(out, "out")
        if tuple(out.shape) != (N, H, W, 8):
            raise PmnError("stem: bad `out` shape")
    with torch.cuda.device(img.device):
        check(_lib.lib().pmn_stem(img.data_ptr(), w0.data_ptr(), s0.data_ptr(), w1.data_ptr(), s1.data_ptr(), out.data_ptr(),
                                  N, H, W, _stream(img)), "pmn_stem")
    return out


def stem_f16s(img: torch.Tensor, w0: torch.Tensor, s0: torch.Tensor, w1a: torch.Tensor, s1: torch.Tensor,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pmn_stem_f16s: FeatureNet conv0 (fp32 VALU) + conv1 (fp16 matrix cores, split operands) fused; img [N,3,H,W] -> [N,H,W,8]
    (optionally into ``out``); w1a = params.pack_stem_conv1_f16s (float16 [3,2,64,8])."""
    for n_, t_ in (("img", img), ("w0", w0), ("s0", s0), ("s1", s1)):
        _dev(t_, n_)
    if not isinstance(w1a, torch.Tensor) or not w1a.is_cuda or w1a.dtype != torch.float16 or tuple(w1a.shape) != (3, 2, 64, 8) \
            or not w1a.is_contiguous():
        raise PmnError("stem_f16s: w1a must be the float16 [3,2,64,8] tensor of params.pack_stem_conv1_f16s on a ROCm GPU")
    N, c, H, W = img.shape
    if c != 3 or tuple(w0.shape) != (3, 3, 3, 8):
        raise PmnError("stem_f16s: expects a 3-channel image and 3->8 conv0 weights")
    _f16_domain_probe(img)
    if out is None:
        out = torch.empty((N, H, W, 8), dtype=torch.float32, device=img.device)
    else:
        _dev(out, "out")
        if tuple(out.shape) != (N, H, W, 8):
            raise PmnError("stem_f16s: bad `out` shape")
    with torch.cuda.device(img.device):
        check(_lib.lib().pmn_stem_f16s(img.data_ptr(), w0.data_ptr(), s0.data_ptr(), w1a.data_ptr(), s1.data_ptr(), out.data_ptr(),
                                       N, H, W, _stream(img)), "pmn_stem_f16s")
    return out


def stem_f16s_views(images: "SourceTable", w0: torch.Tensor, s0: torch.Tensor, w1a: torch.Tensor, s1: torch.Tensor) -> torch.Tensor:
    """pmn_stem_f16s_views: the fused stem over ``views`` separately allocated [B,3,H,W] images found through a device table of
    addresses (``images``: a SourceTable with shape (views, B, 3, H, W)) -> [views*B,H,W,8] view-major.  A captured launch reads
    whatever the table holds at replay time (graph.GraphedForward(inputs_in_place=True))."""
    for n_, t_ in (("w0", w0), ("s0", s0), ("s1", s1)):
        _dev(t_, n_)
    if not isinstance(images, SourceTable) or len(images.shape) != 5 or images.shape[2] != 3:
        raise PmnError("stem_f16s_views: a SourceTable of shape (views, B, 3, H, W)")
    if not isinstance(w1a, torch.Tensor) or not w1a.is_cuda or w1a.dtype != torch.float16 or tuple(w1a.shape) != (3, 2, 64, 8) \
            or not w1a.is_contiguous() or tuple(w0.shape) != (3, 3, 3, 8):
        raise PmnError("stem_f16s_views: 3->8 conv0 weights and the float16 [3,2,64,8] tensor of params.pack_stem_conv1_f16s")
    V, B, _, H, W = images.shape
    out = torch.empty((V * B, H, W, 8), dtype=torch.float32, device=images.device)
    with torch.cuda.device(images.device):
        check(_lib.lib().pmn_stem_f16s_views(images.table.data_ptr(), V, w0.data_ptr(), s0.data_ptr(), w1a.data_ptr(), s1.data_ptr(),
                                             out.data_ptr(), B, H, W, _stream(out)), "pmn_stem_f16s_views")
    return out


def fuse_view(maps: torch.Tensor, ref_slot: int, src_slots: Sequence[int], mats: torch.Tensor, geo_pixel_thres: float,
              geo_depth_thres: float, geo_mask_thres: int, photo_thres: float, want_depth_avg: bool = False,
              want_geo_sum: bool = False, sizes: Optional[Sequence[Tuple[int, int]]] = None):
    """pmn_fuse_view: consistency filtering + fusion of ONE reference view (reference eval.py:86-190, :207-281).

    maps [V,2,H,W] (depth, confidence of every view of the scan, e.g. the all-gathered buffer), or -- views of different sizes
    -- [V,F] flat slots with ``sizes[v] = (h_v, w_v)``: slot v holds depth [h_v,w_v] then confidence [h_v,w_v] packed at its
    start (F >= 2*h*w of the largest view).  mats = fusion.camera_block(...) (device float32).  Returns (masks [3,H,W] uint8 =
    photo / geo / final, xyz [H,W,3] float32 world points, depth_avg [H,W] float64 or None, geo_sum [H,W] int32 or None), H x W
    = the reference view's size."""
    _dev(maps, "maps")
    _dev(mats, "mats")
    if sizes is None:
        if maps.dim() != 4 or maps.shape[1] != 2:
            raise PmnError("fuse_view: maps must be [V,2,H,W] (or [V,F] flat slots with sizes=)")
        V, _, H, W = maps.shape
        stride = 2 * H * W
        sizes = [(H, W)] * V
    else:
        if maps.dim() != 2 or len(sizes) != maps.shape[0]:
            raise PmnError("fuse_view: with sizes= maps must be [V,F] flat slots and sizes must have V entries")
        V, stride = maps.shape
        if any(h < 1 or w < 1 or 2 * h * w > stride for h, w in sizes):
            raise PmnError("fuse_view: a view's maps do not fit its slot")
    n = len(src_slots)
    if n > _lib.MAX_FUSE_SRC:
        raise PmnError(f"fuse_view: at most {_lib.MAX_FUSE_SRC} source views per reference view")
    if not 0 <= ref_slot < V or any(not 0 <= s < V for s in src_slots):
        raise PmnError("fuse_view: slot out of range")
    if mats.numel() != 48 + 64 * n:
        raise PmnError("fuse_view: mats must hold 48 + 64 * n_src floats (fusion.camera_block)")
    H, W = sizes[ref_slot]
    slots, slots_p = _host_i32(np.asarray(list(src_slots) if n else [0], np.int32), max(n, 1), "src_slots")
    hw, hw_p = _host_i32(np.asarray([x for s in src_slots for x in sizes[s]] if n else [0, 0], np.int32), max(2 * n, 2), "src_hw")
    masks = torch.empty((3, H, W), dtype=torch.uint8, device=maps.device)
    xyz = torch.empty((H, W, 3), dtype=torch.float32, device=maps.device)
    davg = torch.empty((H, W), dtype=torch.float64, device=maps.device) if want_depth_avg else None
    gsum = torch.empty((H, W), dtype=torch.int32, device=maps.device) if want_geo_sum else None
    with torch.cuda.device(maps.device):
        check(_lib.lib().pmn_fuse_view(maps.data_ptr(), int(stride), int(ref_slot), slots_p, hw_p, n, mats.data_ptr(), H, W,
                                       float(geo_pixel_thres), float(geo_depth_thres), int(geo_mask_thres), float(photo_thres),
                                       masks.data_ptr(), xyz.data_ptr(), _ptr(davg), _ptr(gsum), _stream(maps)),
              "pmn_fuse_view")
    del slots, hw
    return masks, xyz, davg, gsum


class PointPacker:
    """pmn_pack_points: the PLY vertex records of a scan's fused reference views, packed on the device view after view into ONE
    record buffer (reference eval.py:270-297).  ``capacity`` = room in points (a view can keep at most H*W).

        packer = PointPacker(capacity, device)
        for every reference view:  packer.append(masks[2], xyz, image_hwc)        # three launches on the current stream
        counts = packer.counts()                                                   # synchronises the current stream
        body = packer.records[:15 * sum(counts)]                                   # device uint8: the PLY body in pair-file order
    """

    def __init__(self, capacity: int, device, max_views: int = 1024) -> None:
        self.capacity = int(capacity)
        self.records = torch.empty((15 * self.capacity,), dtype=torch.uint8, device=device)
        self.cursor = torch.zeros((1,), dtype=torch.int64, device=device)
        self.view_counts = torch.zeros((max_views,), dtype=torch.int32, device=device)
        self.scratch = None
        self.n = 0

    def reset(self) -> None:
        self.cursor.zero_()
        self.n = 0

    def append(self, final_mask: torch.Tensor, xyz: torch.Tensor, image_hwc: torch.Tensor) -> None:
        _dev(xyz, "xyz")
        for t, name in ((final_mask, "final_mask"), (image_hwc, "image_hwc")):
            if not isinstance(t, torch.Tensor) or t.device != self.records.device:
                raise PmnError(f"pack_points: {name} must be a tensor on {self.records.device} (no CPU fallback)")
        H, W = final_mask.shape
        if final_mask.dtype != torch.uint8 or not final_mask.is_contiguous():
            raise PmnError("pack_points: final_mask must be contiguous uint8 [H,W]")
        if tuple(xyz.shape) != (H, W, 3) or xyz.dtype != torch.float32 or not xyz.is_contiguous():
            raise PmnError("pack_points: xyz must be contiguous float32 [H,W,3]")
        if tuple(image_hwc.shape) != (H, W, 3) or image_hwc.dtype not in (torch.uint8, torch.float32) or not image_hwc.is_contiguous():
            raise PmnError("pack_points: image must be contiguous uint8 or float32 [H,W,3]")
        if self.n >= self.view_counts.numel():
            raise PmnError("pack_points: more views than the packer was sized for")
        nb = (H * W + 1023) // 1024
        if self.scratch is None or self.scratch.numel() < nb:
            self.scratch = torch.empty((nb,), dtype=torch.int64, device=self.records.device)
        with torch.cuda.device(self.records.device):
            check(_lib.lib().pmn_pack_points(final_mask.data_ptr(), xyz.data_ptr(), image_hwc.data_ptr(),
                                             1 if image_hwc.dtype == torch.float32 else 0, H, W, self.records.data_ptr(),
                                             self.capacity, self.cursor.data_ptr(), self.view_counts[self.n:].data_ptr(),
                                             self.scratch.data_ptr(), _stream(self.records)), "pmn_pack_points")
        self.n += 1

    def counts(self):
        """Per-view point counts (host list); raises if a view did not fit."""
        c = self.view_counts[:self.n].cpu().tolist()
        if any(x < 0 for x in c):
            raise PmnError("pack_points: the record buffer is too small for this scan")
        return c
