"""Learned PatchMatch on MI355X: host-side mirror of the reference's ``models/patchmatch.py`` interface.

``PatchMatch.forward`` keeps the reference signature and return values (models/patchmatch.py:428-529) and the module
tree keeps the reference's parameter names, so ``params_000007.ckpt`` loads unchanged.  The arithmetic runs in four
HIP kernels per iteration-independent / per-iteration step (see patchmatchnet_amd/csrc):

    once per stage   propa_conv / eval_conv: both offset heads as ONE dilated 3x3 convolution on the matrix cores
                     (pmn_conv2d_mfma, planar form; ``hip_offset_heads = False`` runs the nn.Conv2d modules on MIOpen)
                     pmn_nchw_to_nhwc       feature maps -> channels-last
                     pmn_feature_weight     FeatureWeightNet (+ get_grid)
    per iteration    pmn_init_hypotheses    DepthInitialization + Propagation (+ per-pixel sort)
                     pmn_warp_correlate     differentiable_warping + group correlation + PixelwiseNet / view
                                            aggregation + SimilarityNet MLP               <- the hot kernel
                     pmn_aggregate_regress  depth_weight + adaptive aggregation + softmax + regression
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

import warnings

from . import _lib, ops, params
from ._lib import PmnError
from .module import ConvBnReLU3D, is_empty


class _PointwiseMLP(nn.Module):
    """G -> 16 -> 8 -> 1 pointwise network; ``_last`` names the final Conv3d attribute."""

    _last = "similarity"

    def __init__(self, G: int) -> None:
        super().__init__()
        self.G = G
        self.conv0 = ConvBnReLU3D(in_channels=G, out_channels=16, kernel_size=1, stride=1, pad=0)
        self.conv1 = ConvBnReLU3D(in_channels=16, out_channels=8, kernel_size=1, stride=1, pad=0)
        self._packed: Optional[np.ndarray] = None
        self._packed_key = None
        self._packed_dev: Optional[torch.Tensor] = None

    def _sources(self) -> List[torch.Tensor]:
        last = getattr(self, self._last)
        return [self.conv0.conv.weight, *self.conv0.bn_tensors(), self.conv1.conv.weight, *self.conv1.bn_tensors(),
                last.weight, last.bias]

    def packed(self) -> np.ndarray:
        """BN-folded float32[340] block for the kernel-argument segment (cached until a parameter changes)."""
        key = params.versions(self._sources())
        if self._packed is None or key != self._packed_key:
            last = getattr(self, self._last)
            self._packed = params.pack_mlp(self.conv0.conv.weight, self.conv0.bn_tensors(), self.conv1.conv.weight,
                                           self.conv1.bn_tensors(), last.weight, last.bias)
            self._packed_key = key
            self._packed_dev = None
        return self._packed

    def packed_device(self) -> torch.Tensor:
        """The packed block as a device tensor on the parameters' device (what the kernels read)."""
        blk = self.packed()
        dev = self.conv0.conv.weight.device
        if self._packed_dev is None or self._packed_dev.device != dev:
            self._packed_dev = torch.from_numpy(blk).to(dev)
        return self._packed_dev


class PixelwiseNet(_PointwiseMLP):
    """Pixel-wise view weight network (reference models/patchmatch.py:672-702); evaluated inside pmn_warp_correlate."""

    _last = "conv2"

    def __init__(self, G: int) -> None:
        super().__init__(G)
        self.conv2 = nn.Conv3d(in_channels=8, out_channels=1, kernel_size=1, stride=1, padding=0)
        self.output = nn.Sigmoid()


class SimilarityNet(_PointwiseMLP):
    """Similarity network (reference models/patchmatch.py:532-577): MLP inside pmn_warp_correlate, neighbour
    aggregation inside pmn_aggregate_regress."""

    def __init__(self, G: int) -> None:
        super().__init__(G)
        self.similarity = nn.Conv3d(in_channels=8, out_channels=1, kernel_size=1, stride=1, padding=0)


class FeatureWeightNet(_PointwiseMLP):
    """Feature weight network (reference models/patchmatch.py:580-624) -> pmn_feature_weight."""

    def __init__(self, neighbors: int = 9, G: int = 8) -> None:
        super().__init__(G)
        self.neighbors = neighbors
        self.similarity = nn.Conv3d(in_channels=8, out_channels=1, kernel_size=1, stride=1, padding=0)
        self.output = nn.Sigmoid()

    def forward(self, ref_nhwc: torch.Tensor, eval_offsets: torch.Tensor, table: np.ndarray) -> torch.Tensor:
        """ref_nhwc [B,h,w,C], eval_offsets [B,2K,h,w] (raw eval_conv output) -> weights [B,K,h,w]."""
        return ops.feature_weight(ref_nhwc, eval_offsets, table, self.packed_device(), self.G)


class DepthInitialization(nn.Module):
    """Hypothesis generation (reference models/patchmatch.py:17-94), fused with Propagation in pmn_init_hypotheses."""

    def __init__(self, patchmatch_num_sample: int = 1) -> None:
        super().__init__()
        self.patchmatch_num_sample = patchmatch_num_sample

    def forward(self, min_depth: torch.Tensor, max_depth: torch.Tensor, height: int, width: int,
                depth_interval_scale: float, device: torch.device, depth: torch.Tensor) -> torch.Tensor:
        """Same signature as the reference; returns depth_sample [B,D,H,W] (no propagation)."""
        noise = None
        if is_empty(depth):
            noise = torch.rand(size=(min_depth.size()[0], 48, height, width), device=device)
        ds, _ = ops.init_hypotheses(noise, None if noise is not None else depth.detach().contiguous(), 0,
                                    min_depth.float().contiguous(), max_depth.float().contiguous(),
                                    self.patchmatch_num_sample, depth_interval_scale, None, None, height, width)
        return ds


class Propagation(nn.Module):
    """Adaptive propagation (reference models/patchmatch.py:97-124); executed inside pmn_init_hypotheses."""

    def __init__(self) -> None:
        super().__init__()


class Evaluation(nn.Module):
    """Adaptive evaluation (reference models/patchmatch.py:127-239): pmn_warp_correlate + pmn_aggregate_regress."""

    def __init__(self, G: int = 8) -> None:
        super().__init__()
        self.G = G
        self.pixel_wise_net = PixelwiseNet(self.G)
        self.softmax = nn.LogSoftmax(dim=1)
        self.similarity_net = SimilarityNet(self.G)

    def forward(self, ref_feature: torch.Tensor, src_features: List[torch.Tensor], ref_proj: torch.Tensor,
                src_projs: List[torch.Tensor], depth_sample: torch.Tensor, grid: Optional[torch.Tensor],
                weight: Optional[torch.Tensor], view_weights: torch.Tensor, is_inverse: bool, *,
                fused: Optional[dict] = None, debug: Optional[dict] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """The reference's signature and return values (models/patchmatch.py:145-239): ref_feature [B,C,h,w], src_features
        N x [B,C,h,w], ref_proj [B,4,4], src_projs N x [B,4,4], depth_sample [B,D,h,w], grid [B,K*h,w,2] (get_grid's normalised
        neighbour positions), weight [B,D,K,h,w] (normalised aggregation weights), view_weights [B,N,h,w] or empty, is_inverse
        -> (depth [B,h,w], score [B,D,h,w], view_weights [B,N,h,w]).

        warp + group correlation + PixelwiseNet / view aggregation + the SimilarityNet MLP always run in pmn_warp_correlate.
        What follows depends on what the caller hands over:
          * ``grid`` / ``weight`` (a maintainer swapping this class into the reference's PatchMatch.forward): the reference's
            own tail on the device -- grid_sample of the pointwise cost at ``grid`` (bilinear, border, align_corners=False),
            sum over the K neighbours with ``weight``, softmax, regression (:569-577, :221-237);
          * ``fused`` (this package's PatchMatch.forward): {xnorm, eval_offsets, table, feature_weight, interval_scale,
            vw_shift, ref_nhwc, src_nhwc, rel_proj}: pmn_aggregate_regress recomputes grid and weight in registers from the raw
            conv offsets instead of reading the materialised [B,D,K,h,w] tensors; grid / weight may then be None."""
        N = len(src_features) if src_features is not None else fused["src_nhwc"].shape[0]
        if src_features is not None and src_projs is not None and len(src_features) != len(src_projs):
            raise AssertionError("Patchmatch Evaluation: Different number of images and projection matrices")
        have_vw = not is_empty(view_weights)
        if have_vw and view_weights.size()[1] != N:
            raise AssertionError("Patchmatch Evaluation: Different number of images and view weights")
        f = fused or {}
        ref_nhwc = f.get("ref_nhwc")
        src_nhwc = f.get("src_nhwc")
        rel_proj = f.get("rel_proj")
        if ref_nhwc is None:
            ref_nhwc = ops.nchw_to_nhwc(ref_feature.detach())
        if rel_proj is None:
            rel_proj = ops.relative_projection(src_projs, ref_proj)
        if src_nhwc is None:  # source views of different sizes (legal in the reference): zero-padded + projection rows rescaled
            src_nhwc, sizes = ops.stack_sources_padded([s.detach() for s in src_features])
            rel_proj = ops.rescale_projection_rows(rel_proj, sizes, tuple(src_nhwc.shape[2:4]))
        depth_sample = depth_sample.contiguous()
        cost, vw, argmax, sim = ops.warp_correlate(
            ref_nhwc, src_nhwc, rel_proj.contiguous(), depth_sample, view_weights.contiguous() if have_vw else None,
            int(f.get("vw_shift", 0)), self.similarity_net.packed_device(),
            None if have_vw else self.pixel_wise_net.packed_device(), self.G,
            want_similarity=debug is not None, want_argmax=debug is not None and not have_vw)
        if fused is not None and grid is None:
            score, depth = ops.aggregate_regress(cost, depth_sample, f["xnorm"], f["feature_weight"], f["eval_offsets"],
                                                 f["table"], f["interval_scale"], is_inverse)
        else:
            if grid is None or weight is None:
                raise PmnError("Evaluation.forward needs grid and weight (reference signature) or the fused-form arguments")
            B, D, h, w = depth_sample.shape
            K = grid.shape[1] // h
            sampled = torch.nn.functional.grid_sample(cost.contiguous(), grid, mode="bilinear", padding_mode="border",
                                                      align_corners=False).view(B, D, K, h, w)
            score = torch.sum(sampled * weight, dim=2)
            score = torch.exp(self.softmax(score))
            if is_inverse:
                index = torch.arange(0, D, 1, device=score.device).view(1, D, 1, 1)
                index = torch.sum(index * score, dim=1)
                inv_min = 1.0 / depth_sample[:, -1, :, :]
                inv_max = 1.0 / depth_sample[:, 0, :, :]
                depth = 1.0 / (inv_max + index / (D - 1) * (inv_min - inv_max))
            else:
                depth = torch.sum(depth_sample * score, dim=1)
        if debug is not None:
            debug.update(cost=cost, similarity=sim, view_weight_argmax=argmax)
        return depth, score, vw.detach()


class PatchMatch(nn.Module):
    """One PatchMatch stage; constructor and ``forward`` as the reference (models/patchmatch.py:242-312, 428-529)."""

    def __init__(self, propagation_out_range: int = 2, patchmatch_iteration: int = 2, patchmatch_num_sample: int = 16,
                 patchmatch_interval_scale: float = 0.025, num_feature: int = 64, G: int = 8,
                 propagate_neighbors: int = 16, evaluate_neighbors: int = 9, stage: int = 3) -> None:
        super().__init__()
        self.patchmatch_iteration = patchmatch_iteration
        self.patchmatch_interval_scale = patchmatch_interval_scale
        self.propa_num_feature = num_feature
        self.G = G
        self.stage = stage
        self.dilation = propagation_out_range
        self.propagate_neighbors = propagate_neighbors
        self.evaluate_neighbors = evaluate_neighbors
        if propagate_neighbors not in (0, 4, 8, 16) or evaluate_neighbors not in (9, 17):
            raise NotImplementedError  # same legal sets as reference get_grid (:331-394)

        self.depth_initialization = DepthInitialization(patchmatch_num_sample)
        self.propagation = Propagation()
        self.evaluation = Evaluation(self.G)
        # offset heads: parameter containers (run as HIP convolutions, see forward); zero-initialised like the reference (:288-311)
        self.propa_conv = nn.Conv2d(num_feature, max(2 * propagate_neighbors, 1), kernel_size=3, stride=1,
                                    padding=self.dilation, dilation=self.dilation, bias=True)
        nn.init.constant_(self.propa_conv.weight, 0.0)
        nn.init.constant_(self.propa_conv.bias, 0.0)
        self.eval_conv = nn.Conv2d(num_feature, 2 * evaluate_neighbors, kernel_size=3, stride=1, padding=self.dilation,
                                   dilation=self.dilation, bias=True)
        nn.init.constant_(self.eval_conv.weight, 0.0)
        nn.init.constant_(self.eval_conv.bias, 0.0)
        self.feature_weight_net = FeatureWeightNet(evaluate_neighbors, self.G)

        # offset heads in HIP (True) or on MIOpen (False: what the parity tests compare with)
        self.hip_offset_heads = True
        # ONE switch for the HIP form: f16_split = True (default) = both heads of the stage as one dilated convolution on the FP16 matrix
        # cores with split operands (pmn_offset_heads_f16s: fp32-convolution accuracy, csrc/conv_f16s.hip) wherever the kernel covers the
        # shape -- the reference's three (channels, dilation) pairs with the row count padded to 32, 48 or 64 -- and the heads' weights
        # fit float16's range; otherwise, and with False, pmn_conv2d (fp32 VALU), one launch per head.
        self.f16_split = True
        self.f16_domain_error: Optional[str] = None
        self.heads_hook = None  # None in the product; patchmatchnet_amd/research.py may install round 2's fp32 MFMA form of the heads
        self._heads = None
        self._heads_key = None
        self._ptable = params.propagation_table(propagate_neighbors, self.dilation) if propagate_neighbors > 0 else None
        self._etable = params.evaluation_table(evaluate_neighbors, self.dilation)

    def _packed_heads(self):
        srcs = [self.propa_conv.weight, self.propa_conv.bias, self.eval_conv.weight, self.eval_conv.bias]
        key = params.versions(srcs)
        if self._heads is None or key != self._heads_key:
            dev = self.eval_conv.weight.device
            pk = {}
            for name, m in (("propa", self.propa_conv), ("eval", self.eval_conv)):
                w, s = params.pack_conv(m.weight, bias=m.bias)
                pk[name] = (torch.from_numpy(w).to(dev), torch.from_numpy(s).to(dev))
            # matrix-core form: both heads as ONE convolution (propa rows, then eval rows), with and without the propa rows
            self.f16_domain_error = None
            for name, mods in (("both", (self.propa_conv, self.eval_conv)), ("eval_only", (self.eval_conv,))):
                wcat = torch.cat([m.weight.detach() for m in mods], 0)
                bcat = torch.cat([m.bias.detach() for m in mods], 0)
                rows = wcat.shape[0]
                # pmn_offset_heads_f16s instantiates padded row counts 32, 48 and 64 (16 -- e.g. no propagation and <= 8 evaluation
                # neighbours -- is not one of them: those configurations take pmn_conv2d)
                if (self.eval_conv.in_channels, self.dilation) in ops.F16S_HEAD_SHAPES and (rows + 15) // 16 * 16 in (32, 48, 64):
                    try:
                        w, s = params.pack_offset_heads_f16s(wcat, bcat)
                        pk["f16s_" + name] = (torch.from_numpy(w).to(dev), torch.from_numpy(s).to(dev))
                    except params.F16DomainError as e:
                        self.f16_domain_error = f"offset heads: {e}"
            if self.f16_domain_error is not None:
                warnings.warn(f"PatchMatch stage {self.stage}: " + self.f16_domain_error + " -- using pmn_conv2d (fp32)", RuntimeWarning,
                              stacklevel=2)
            self._heads, self._heads_key = pk, key
        return self._heads

    def forward(self, ref_feature: torch.Tensor, src_features: List[torch.Tensor], ref_proj: torch.Tensor,
                src_projs: List[torch.Tensor], depth_min: torch.Tensor, depth_max: torch.Tensor, depth: torch.Tensor,
                view_weights: torch.Tensor, depth_shift: int = 0, vw_shift: int = 0, noise: Optional[torch.Tensor] = None,
                debug: Optional[list] = None, ref_nhwc: Optional[torch.Tensor] = None,
                src_nhwc: Optional[torch.Tensor] = None, rel_proj: Optional[torch.Tensor] = None
                ) -> Tuple[List[torch.Tensor], torch.Tensor, torch.Tensor]:
        """Reference arguments, plus optional extras that default to reference behaviour:
        ``depth_shift`` / ``vw_shift`` = 1 read ``depth`` / ``view_weights`` given at half resolution through the
        nearest x2 up-sampling (skips materialising F.interpolate; the ``view_weights`` RETURNED are then the tensor that was
        passed in, still at its coarser resolution -- the reference returns them up-sampled); ``noise`` pins the stage-3 random draw;
        ``debug`` (a list) receives one dict of intermediates per iteration; ``ref_nhwc`` [B,h,w,C] / ``src_nhwc``
        [N,B,h,w,C] hand over channels-last copies the caller already made (one layout pass for all views); ``rel_proj``
        [B,N,4,4] hands over src_proj @ inverse(ref_proj) when the caller already has it (then ref_proj / src_projs are
        not read)."""
        if len(src_features) != len(src_projs):
            raise AssertionError("Patchmatch Evaluation: Different number of images and projection matrices")
        if not ref_feature.is_cuda:
            raise PmnError("patchmatchnet_amd.PatchMatch runs on a ROCm GPU only (no CPU fallback)")
        device = ref_feature.device
        batch, _, height, width = ref_feature.size()

        propagate_any = self.propagate_neighbors > 0 and not (self.stage == 1 and self.patchmatch_iteration == 1)
        if ref_nhwc is None:
            ref_nhwc = ops.nchw_to_nhwc(ref_feature.detach())
        if self.hip_offset_heads:
            # offset heads as HIP convolutions on the channels-last reference feature, planar [B,2K,h,w] output
            pk = self._packed_heads()
            fkey = "f16s_both" if propagate_any else "f16s_eval_only"
            if self.f16_split and fkey in pk:  # fp16 matrix cores, split operands (round 3)
                n_p, n_e = (2 * self.propagate_neighbors if propagate_any else 0), 2 * self.evaluate_neighbors
                a_, b_ = ops.offset_heads_f16s(ref_nhwc, *pk[fkey], n_p + n_e, n_p if propagate_any else n_e, self.dilation)
                propa_offsets, eval_offsets = (a_, b_) if propagate_any else (None, a_)
            elif self.heads_hook is not None and (hooked := self.heads_hook(self, ref_nhwc, propagate_any)) is not None:  # research build
                propa_offsets, eval_offsets = hooked
            else:
                propa_offsets = ops.conv2d(ref_nhwc, *pk["propa"], 2 * self.propagate_neighbors, 3, 1, self.dilation,
                                           self.dilation, out_nchw=True) if propagate_any else None
                eval_offsets = ops.conv2d(ref_nhwc, *pk["eval"], 2 * self.evaluate_neighbors, 3, 1, self.dilation,
                                          self.dilation, out_nchw=True)
        else:
            ref_feature = ref_feature.contiguous()
            propa_offsets = self.propa_conv(ref_feature).contiguous() if propagate_any else None
            eval_offsets = self.eval_conv(ref_feature).contiguous()
        if rel_proj is None:
            rel_proj = ops.relative_projection(src_projs, ref_proj)
        if src_nhwc is None:  # source views of different sizes (legal in the reference): zero-padded + projection rows rescaled
            src_nhwc, sizes = ops.stack_sources_padded([f.detach() for f in src_features])
            rel_proj = ops.rescale_projection_rows(rel_proj, sizes, tuple(src_nhwc.shape[2:4]))
        rel_proj = rel_proj.contiguous()
        depth_min = depth_min.float().contiguous()
        depth_max = depth_max.float().contiguous()

        feature_weight = self.feature_weight_net(ref_nhwc, eval_offsets, self._etable)

        depth_sample = depth
        cur_shift = depth_shift
        score = torch.empty(0, device=device)
        depth_samples: List[torch.Tensor] = []
        for it in range(1, self.patchmatch_iteration + 1):
            is_inverse = self.stage == 1 and it == self.patchmatch_iteration
            first_random = is_empty(depth_sample)
            if first_random and noise is None:
                # same RNG call as the reference (models/patchmatch.py:61-62) so a seeded run draws the same numbers
                noise = torch.rand(size=(batch, 48, height, width), device=device)
            propagate = self.propagate_neighbors > 0 and not (self.stage == 1 and it == self.patchmatch_iteration)
            hyp, xnorm = ops.init_hypotheses(
                noise.contiguous() if first_random else None,
                None if first_random else depth_sample.detach().contiguous(), cur_shift, depth_min, depth_max,
                self.depth_initialization.patchmatch_num_sample, self.patchmatch_interval_scale,
                propa_offsets if propagate else None, self._ptable if propagate else None, height, width)
            rec = {} if debug is not None else None
            had_view_weights = not is_empty(view_weights)
            d, score, view_weights = self.evaluation(
                ref_feature=ref_feature, src_features=src_features, ref_proj=ref_proj, src_projs=src_projs, depth_sample=hyp,
                grid=None, weight=None, view_weights=view_weights, is_inverse=is_inverse,
                fused=dict(xnorm=xnorm, eval_offsets=eval_offsets, table=self._etable, feature_weight=feature_weight,
                           interval_scale=self.patchmatch_interval_scale, vw_shift=vw_shift, ref_nhwc=ref_nhwc,
                           src_nhwc=src_nhwc, rel_proj=rel_proj), debug=rec)
            if not had_view_weights:
                vw_shift = 0  # weights were just computed at this stage's resolution
            if rec is not None:
                rec.update(depth_sample=hyp, xnorm=xnorm, feature_weight=feature_weight, score=score, depth=d,
                           view_weights=view_weights, eval_offsets=eval_offsets, propa_offsets=propa_offsets)
                debug.append(rec)
            depth_sample = d.unsqueeze(1)
            cur_shift = 0
            depth_samples.append(depth_sample)
        return depth_samples, score, view_weights
