"""PatchmatchNet cascade on MI355X: the outer drop-in boundary (mirror of the reference's ``models/net.py`` interface).

``PatchmatchNet(...)`` takes the reference's constructor arguments, exposes the same sub-module / state-dict names
(``feature``, ``patchmatch_1..3``, ``upsample_net``) and ``forward`` returns the same triple.  Everything runs in the HIP
kernels of ``patchmatchnet_amd/csrc`` -- the learned-PatchMatch cascade (no other implementation exists: no CPU / eager
fallback) and, by default, FeatureNet and Refinement too (``hip_feature_net``: stem + split-operand fp16 matrix-core convolutions,
folded FPN head, fused refinement; ``f16_split = False`` or a checkpoint outside float16's range: the fp32 kernels).  FeatureNet and Refinement are also ordinary nn.Modules with the reference's parameters; with
``hip_feature_net = False`` they run on PyTorch-ROCm / MIOpen, which is what the parity tests compare the HIP path with.

INFERENCE ONLY: ``forward`` raises in training mode -- the kernels have no backward pass and the reference's train.py /
patchmatchnet_loss have no counterpart here (out of scope).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops, params
from ._lib import PmnError
from .module import ConvBnReLU
from .patchmatch import PatchMatch


class FeatureNet(nn.Module):
    """FPN feature extractor (reference models/net.py:9-70); outputs {3: [B,64,H/8,W/8], 2: [B,32,H/4,W/4],
    1: [B,16,H/2,W/2]}.  ``forward`` = the reference's op sequence on PyTorch-ROCm (MIOpen): the parity reference;
    ``forward_hip`` = the same parameters through the HIP convolutions (what PatchmatchNet uses by default)."""

    def __init__(self) -> None:
        super().__init__()
        spec = [(3, 8, 3, 1, 1), (8, 8, 3, 1, 1), (8, 16, 5, 2, 2), (16, 16, 3, 1, 1), (16, 16, 3, 1, 1),
                (16, 32, 5, 2, 2), (32, 32, 3, 1, 1), (32, 32, 3, 1, 1), (32, 64, 5, 2, 2), (64, 64, 3, 1, 1),
                (64, 64, 3, 1, 1)]
        for i, (cin, cout, k, s, p) in enumerate(spec):
            setattr(self, f"conv{i}", ConvBnReLU(cin, cout, k, s, p))
        self.output1 = nn.Conv2d(64, 64, 1, bias=False)
        self.inner1 = nn.Conv2d(32, 64, 1, bias=True)
        self.inner2 = nn.Conv2d(16, 64, 1, bias=True)
        self.output2 = nn.Conv2d(64, 32, 1, bias=False)
        self.output3 = nn.Conv2d(64, 16, 1, bias=False)
        # forward_hip has ONE switch: f16_split = True (default) runs conv1..conv10 on the FP16 matrix cores with split (hi + lo/2048)
        # operands -- fp32-convolution accuracy at 16/3 the fp32 MFMA rate (csrc/conv_f16s.hip); False = the fp32 kernels (pmn_stem +
        # pmn_conv2d, the VALU direct convolution).  A checkpoint whose BatchNorm-folded weights leave float16's finite range
        # (include/pmn_hip.h, "fp16-split entry points") takes the fp32 kernels too and warns once (``f16_domain_error`` says why).
        self.f16_split = True
        self.f16_domain_error: Optional[str] = None
        # verification switch, not a performance choice: False = the FPN head layer by layer in the reference's order instead of the
        # host-composed 1x1 convolutions (an algebraic re-association)
        self.fold_fpn = True
        # conv3 + conv4 as ONE launch (pmn_conv2d_f16s_pair, bit-identical to the two launches); False = one launch per layer
        self.fuse_conv34 = True
        # verification switch (test hook): the FPN's 1/8 level through the VALU kernel (pmn_fpn_level) instead of the fp32 matrix cores
        self.fpn8_valu = False
        # None in the product.  patchmatchnet_amd/research.py (PMN_EXPERIMENTAL=1 only) installs a callable (layer index, input) ->
        # output-or-None here to run rounds 1-2's fp32 alternatives for single layers when f16_split is False.
        self.layer_hook = None

    # ---- HIP execution (pmn_conv2d): same parameters, channels-last activations, BN/ReLU/FPN-add fused ----------------
    _SPEC = [(3, 1, 1), (3, 1, 1), (5, 2, 2), (3, 1, 1), (3, 1, 1), (5, 2, 2), (3, 1, 1), (3, 1, 1), (5, 2, 2), (3, 1, 1),
             (3, 1, 1)]  # (kernel, stride, pad) of conv0..conv10

    def _packed(self):
        """Device copies of the conv weights in pmn_conv2d layout (BatchNorm folded), cached until a parameter changes."""
        srcs = [p for p in self.parameters()] + [b for b in self.buffers() if b.dtype.is_floating_point]
        key = params.versions(srcs)
        if getattr(self, "_pack_key", None) != key:
            dev = self.output1.weight.device
            pk = {}
            f16_ok, self.f16_domain_error = True, None
            for i in range(11):
                m = getattr(self, f"conv{i}")
                w, s = params.pack_conv(m.conv.weight, bn=(m.bn.weight, m.bn.bias, m.bn.running_mean, m.bn.running_var),
                                        eps=m.bn.eps)
                pk[f"conv{i}"] = (torch.from_numpy(w).to(dev), torch.from_numpy(s).to(dev))
                cv = m.conv
                if (cv.kernel_size[0], cv.stride[0], cv.in_channels, cv.out_channels) in ops.F16S_SHAPES and f16_ok:
                    try:
                        w, s = params.pack_conv_f16s(cv.weight, bn=(m.bn.weight, m.bn.bias, m.bn.running_mean, m.bn.running_var),
                                                     eps=m.bn.eps)
                        pk[f"conv{i}_f16s"] = (torch.from_numpy(w).to(dev), torch.from_numpy(s).to(dev))
                    except params.F16DomainError as e:
                        f16_ok, self.f16_domain_error = False, f"conv{i}: {e}"
            m1 = self.conv1
            try:
                w, s = params.pack_stem_conv1_f16s(m1.conv.weight, bn=(m1.bn.weight, m1.bn.bias, m1.bn.running_mean, m1.bn.running_var),
                                                  eps=m1.bn.eps)
                pk["conv1_f16s"] = (torch.from_numpy(w).to(dev), torch.from_numpy(s).to(dev))
            except params.F16DomainError as e:
                f16_ok, self.f16_domain_error = False, f"conv1: {e}"
            if not f16_ok:  # the whole network then runs the fp32 kernels: one arithmetic per forward, stated once
                for k in [k for k in pk if k.endswith("_f16s")]:
                    del pk[k]
                warnings.warn("FeatureNet: " + self.f16_domain_error + " -- using the fp32 kernels (pmn_stem / pmn_conv2d) instead of "
                              "the fp16-split matrix-core kernels", RuntimeWarning, stacklevel=2)
            for name in ("output1", "inner1", "inner2", "output2", "output3"):
                m = getattr(self, name)
                w, s = params.pack_conv(m.weight, bias=m.bias)
                pk[name] = (torch.from_numpy(w).to(dev), torch.from_numpy(s).to(dev))
            fold = params.fold_fpn(self.output1.weight, self.inner1.weight, self.inner1.bias, self.inner2.weight,
                                   self.inner2.bias, self.output2.weight, self.output3.weight)
            for lvl, (w, b) in fold.items():
                pk[f"fpn{lvl}"] = (torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev))
            w8 = torch.from_numpy(np.ascontiguousarray(fold[8][0].T))[:, :, None, None]  # [112,64,1,1]: matrix-core form of level 1/8
            w, s = params.pack_conv_mfma(w8, bias=torch.from_numpy(fold[8][1]))
            pk["fpn8_mfma"] = (torch.from_numpy(w).to(dev), torch.from_numpy(s).to(dev))
            self._pack, self._pack_key = pk, key
        return self._pack

    def forward_hip(self, x, image_table: Optional["ops.SourceTable"] = None) -> Dict[int, torch.Tensor]:
        """x [N,3,H,W], or a list of same-size [B,3,H,W] images (stacked view-major without a torch.cat copy: the first
        layer writes each image's output into its slice) -> {3: [N,H/8,W/8,64], 2: [N,H/4,W/4,32], 1: [N,H/2,W/2,16]}
        CHANNELS-LAST (inference only).  ``image_table`` (ops.SourceTable of shape (views, B, 3, H, W)): the images are read through
        a device table of addresses instead of ``x`` (one launch; graph.GraphedForward(inputs_in_place=True))."""
        pk = self._packed()
        imgs = list(x) if isinstance(x, (list, tuple)) else [x]
        B, _, H, W = imgs[0].shape
        in_place = image_table is not None
        if in_place and not (self.f16_split and "conv1_f16s" in pk):
            raise PmnError("image_table: only the fp16-split stem (FeatureNet.f16_split, weights inside its domain) reads images in place")
        if in_place:
            t = ops.stem_f16s_views(image_table, *pk["conv0"], *pk["conv1_f16s"])
        else:
            t = torch.empty((B * len(imgs), H, W, 8), dtype=torch.float32, device=imgs[0].device)
        for i, im in enumerate(() if in_place else imgs):  # conv0 + conv1 fused (pmn_stem_f16s: conv1 on the fp16 matrix cores; pmn_stem: all fp32 VALU)
            if self.f16_split and "conv1_f16s" in pk:
                ops.stem_f16s(im.contiguous(), *pk["conv0"], *pk["conv1_f16s"], out=t[i * B:(i + 1) * B])
            else:
                ops.stem(im.contiguous(), *pk["conv0"], *pk["conv1"], out=t[i * B:(i + 1) * B])
        feats = {}
        for i, (k, s, p) in enumerate(self._SPEC):
            if i < 2:
                continue
            if i == 4 and self.f16_split and self.fuse_conv34 and "conv3_f16s" in pk and "conv4_f16s" in pk:
                continue  # (done with conv3 below)
            if i == 3 and self.f16_split and self.fuse_conv34 and "conv3_f16s" in pk and "conv4_f16s" in pk:
                # conv3 + conv4 in one launch: the half-resolution intermediate (184 MB for six 1600x1200 views) stays in LDS (round 6)
                t = ops.conv2d_f16s_pair(t, *pk["conv3_f16s"], *pk["conv4_f16s"], relu=True)
                feats[4] = t
            elif self.f16_split and f"conv{i}_f16s" in pk:  # fp16 matrix cores, split operands
                t = ops.conv2d_f16s(t, *pk[f"conv{i}_f16s"], k, s, relu=True)
            elif self.layer_hook is not None and (hooked := self.layer_hook(i, t)) is not None:  # research build only
                t = hooked
            else:  # fp32 VALU direct convolution
                w, sh = pk[f"conv{i}"]
                t = ops.conv2d(t, w, sh, getattr(self, f"conv{i}").conv.out_channels, k, s, p, relu=True)
            if i in (4, 7, 10):
                feats[i] = t
        half, quarter, eighth = feats[4], feats[7], feats[10]
        if self.fold_fpn:
            # the FPN head is linear: its 1x1 convolutions are composed on the host (params.fold_fpn) and each level is one
            # bandwidth-bound kernel -- the 64-channel intermediates at 1/4 and 1/2 resolution never exist
            if self.fpn8_valu:  # (test hook: the VALU form of the same level)
                f3, u8 = ops.fpn_level(eighth, None, *pk["fpn8"], ca=64)
            else:  # 64 -> 112 channels: a GEMM, on the fp32 matrix cores (pmn_conv2d_mfma's 1x1 form)
                f3, u8 = ops.pointwise_split_mfma(eighth, *pk["fpn8_mfma"], cout=112, ca=64)
            f2, u4 = ops.fpn_level(quarter, u8, *pk["fpn4"], ca=32)
            f1, _ = ops.fpn_level(half, u4, *pk["fpn2"], ca=16)
            return {3: f3, 2: f2, 1: f1}
        out = {3: ops.conv2d(eighth, *pk["output1"], 64, 1)}
        top = ops.conv2d(quarter, *pk["inner1"], 64, 1, up=eighth)       # upsample(conv10) + inner1(conv7)
        out[2] = ops.conv2d(top, *pk["output2"], 32, 1)
        # output3(upsample(intra) + inner2(conv4)) in one kernel: the 64-channel half-resolution map stays in registers
        out[1] = ops.fpn_tail(half, top, pk["inner2"][0], pk["inner2"][1], pk["output3"][0])
        return out

    def forward(self, x: torch.Tensor) -> Dict[int, torch.Tensor]:
        half = self.conv4(self.conv3(self.conv2(self.conv1(self.conv0(x)))))
        quarter = self.conv7(self.conv6(self.conv5(half)))
        eighth = self.conv10(self.conv9(self.conv8(quarter)))
        out: Dict[int, torch.Tensor] = {3: self.output1(eighth)}
        top = F.interpolate(eighth, scale_factor=2.0, mode="bilinear", align_corners=False) + self.inner1(quarter)
        out[2] = self.output2(top)
        top = F.interpolate(top, scale_factor=2.0, mode="bilinear", align_corners=False) + self.inner2(half)
        out[1] = self.output3(top)
        return out


class Refinement(nn.Module):
    """Depth-residual refinement at full resolution (reference models/net.py:73-122).  ``forward`` = PyTorch-ROCm (parity
    reference), ``forward_hip`` = pmn_conv2d at half resolution + pmn_refine_fused (default; ``f16_split = False``: pmn_refine_front / pmn_refine_tail)."""

    def __init__(self) -> None:
        super().__init__()
        self.conv0 = ConvBnReLU(in_channels=3, out_channels=8)
        self.conv1 = ConvBnReLU(in_channels=1, out_channels=8)
        self.conv2 = ConvBnReLU(in_channels=8, out_channels=8)
        self.deconv = nn.ConvTranspose2d(8, 8, kernel_size=3, padding=1, output_padding=1, stride=2, bias=False)
        self.bn = nn.BatchNorm2d(8)
        self.conv3 = ConvBnReLU(in_channels=16, out_channels=8)
        self.res = nn.Conv2d(8, 1, kernel_size=3, padding=1, bias=False)
        # forward_hip, full-resolution half: f16_split = True (default) = ONE launch with conv3 on the fp16 matrix cores (split operands:
        # pmn_refine_fused); False -- or conv3's folded weights outside float16's range -- = the fp32 pair pmn_refine_front +
        # pmn_refine_tail.
        self.f16_split = True
        self.f16_domain_error: Optional[str] = None
        # verification switch, not a performance choice: one pmn_conv2d / pmn_deconv3x3s2 launch per layer in the reference's order
        self.research = dict(layers=False)

    def _packed(self):
        srcs = [p for p in self.parameters()] + [b for b in self.buffers() if b.dtype.is_floating_point]
        key = params.versions(srcs)
        if getattr(self, "_pack_key", None) != key:
            dev = self.res.weight.device

            def bn_of(m):
                return (m.weight, m.bias, m.running_mean, m.running_var)

            pk = {}
            for name in ("conv0", "conv1", "conv2", "conv3"):
                m = getattr(self, name)
                w, s = params.pack_conv(m.conv.weight, bn=bn_of(m.bn), eps=m.bn.eps)
                pk[name] = (torch.from_numpy(w).to(dev), torch.from_numpy(s).to(dev))
            w, s = params.pack_deconv(self.deconv.weight, bn=bn_of(self.bn), eps=self.bn.eps)
            pk["deconv"] = (torch.from_numpy(w).to(dev), torch.from_numpy(s).to(dev))
            w, s = params.pack_conv(self.res.weight)
            pk["res"] = (torch.from_numpy(w).to(dev), torch.from_numpy(s).to(dev))
            pk["tail"] = tuple(torch.from_numpy(a).to(dev) for a in params.pack_refine_tail(
                self.conv3.conv.weight, bn_of(self.conv3.bn), self.res.weight, eps=self.conv3.bn.eps))
            self.f16_domain_error = None
            try:
                w, s = params.pack_refine_conv3_f16s(self.conv3.conv.weight, bn_of(self.conv3.bn), eps=self.conv3.bn.eps)
                pk["conv3_f16s"] = (torch.from_numpy(w).to(dev), torch.from_numpy(s).to(dev))
            except params.F16DomainError as e:
                self.f16_domain_error = f"conv3: {e}"
                warnings.warn("Refinement: " + self.f16_domain_error + " -- using the fp32 kernels (pmn_refine_front / pmn_refine_tail)",
                              RuntimeWarning, stacklevel=2)
            self._pack, self._pack_key = pk, key
        return self._pack

    def forward_hip(self, img: torch.Tensor, depth_0: torch.Tensor, depth_min: torch.Tensor, depth_max: torch.Tensor
                    ) -> torch.Tensor:
        """Same computation through pmn_conv2d / pmn_deconv3x3s2 (channels-last, BatchNorm + ReLU fused)."""
        pk = self._packed()
        b = depth_min.size()[0]
        if not self.research["layers"]:
            d = ops.normalize_depth(depth_0, depth_min, depth_max)  # (depth_0 - lo) / span, same bits, a launch of the library
        else:
            lo = depth_min.view(b, 1, 1, 1)
            span = (depth_max - depth_min).view(b, 1, 1, 1)
            d = ((depth_0 - lo) / span).contiguous()
        t = ops.conv2d(d, *pk["conv1"], 8, 3, 1, 1, relu=True, in_nchw=True)                             # [B,H/2,W/2,8]
        t = ops.conv2d(t, *pk["conv2"], 8, 3, 1, 1, relu=True)
        if not self.research["layers"]:
            if self.f16_split and "conv3_f16s" in pk:  # the full-resolution half in one launch: x16 never leaves LDS
                return ops.refine_fused(img.contiguous(), t, *pk["conv0"], *pk["deconv"], *pk["conv3_f16s"], pk["tail"][2], d,
                                        depth_min.float().contiguous(), depth_max.float().contiguous())
            # ... in two launches: (deconv || conv0) -> x16, then conv3 -> res -> residual + de-normalisation
            x16 = ops.refine_front(img.contiguous(), t, *pk["conv0"], *pk["deconv"])
            return ops.refine_tail(x16, *pk["tail"], d, depth_min.float().contiguous(), depth_max.float().contiguous())
        img_feat = ops.conv2d(img.contiguous(), *pk["conv0"], 8, 3, 1, 1, relu=True, in_nchw=True)      # [B,H,W,8]
        up = ops.deconv3x3s2(t, *pk["deconv"], relu=True)                                                # [B,H,W,8]
        t = ops.conv2d(torch.cat((up, img_feat), dim=3), *pk["conv3"], 8, 3, 1, 1, relu=True)
        res = ops.conv2d(t, *pk["res"], 1, 3, 1, 1, out_nchw=True)                                       # [B,1,H,W]
        d = F.interpolate(d, scale_factor=2.0, mode="nearest") + res
        return d * span + lo

    def forward(self, img: torch.Tensor, depth_0: torch.Tensor, depth_min: torch.Tensor, depth_max: torch.Tensor
                ) -> torch.Tensor:
        b = depth_min.size()[0]
        lo = depth_min.view(b, 1, 1, 1)
        span = (depth_max - depth_min).view(b, 1, 1, 1)
        d = (depth_0 - lo) / span
        img_feat = self.conv0(img)
        up = F.relu(self.bn(self.deconv(self.conv2(self.conv1(d)))), inplace=True)
        res = self.res(self.conv3(torch.cat((up, img_feat), dim=1)))
        d = F.interpolate(d, scale_factor=2.0, mode="nearest") + res
        return d * span + lo


class PatchmatchNet(nn.Module):
    """Coarse-to-fine learned PatchMatch MVS network; interface of reference models/net.py:125-301."""

    def __init__(self, patchmatch_interval_scale: List[float], propagation_range: List[int],
                 patchmatch_iteration: List[int], patchmatch_num_sample: List[int], propagate_neighbors: List[int],
                 evaluate_neighbors: List[int]) -> None:
        super().__init__()
        self.stages = 4
        self.feature = FeatureNet()
        self.patchmatch_num_sample = patchmatch_num_sample
        num_features = [16, 32, 64]
        self.propagate_neighbors = propagate_neighbors
        self.evaluate_neighbors = evaluate_neighbors
        self.G = [4, 8, 8]
        for i in range(self.stages - 1):
            setattr(self, f"patchmatch_{i + 1}", PatchMatch(
                propagation_out_range=propagation_range[i], patchmatch_iteration=patchmatch_iteration[i],
                patchmatch_num_sample=patchmatch_num_sample[i], patchmatch_interval_scale=patchmatch_interval_scale[i],
                num_feature=num_features[i], G=self.G[i], propagate_neighbors=self.propagate_neighbors[i],
                evaluate_neighbors=evaluate_neighbors[i], stage=i + 1))
        self.upsample_net = Refinement()
        # Run FeatureNet once on the N+1 images stacked along the batch axis when they share a size (same per-sample
        # arithmetic, N+1 times fewer launches).  Set False to mirror the reference's per-image loop exactly.
        self.batch_feature_extraction = True
        # FeatureNet through the HIP convolutions (pmn_conv2d, fp32, channels-last) instead of MIOpen; False = PyTorch-ROCm.
        self.hip_feature_net = True
        # all relative projections from one pmn_stage_projections launch; False = the reference's torch op sequence
        self.hip_projections = True

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """Accepts both plain and ``module.``-prefixed (nn.DataParallel) checkpoints (reference eval.py:33-35)."""
        if any(k.startswith("module.") for k in state_dict):
            state_dict = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        return super().load_state_dict(state_dict, strict=strict, **kw)

    @staticmethod
    def scripted_module_config(archive) -> Dict[str, list]:
        """The six constructor lists of a TorchScript archive of the reference's PatchmatchNet (reference eval.py:37-39,
        checkpoints/module_000007.pt): the scripted module keeps them as readable attributes -- ``patchmatch_num_sample`` on the
        top module (models/net.py:151), the per-stage scalars on ``patchmatch_1..3`` (models/patchmatch.py:270-286)."""
        stages = [getattr(archive, f"patchmatch_{i}") for i in (1, 2, 3)]
        return dict(patchmatch_interval_scale=[float(s.patchmatch_interval_scale) for s in stages],
                    propagation_range=[int(s.dilation) for s in stages],
                    patchmatch_iteration=[int(s.patchmatch_iteration) for s in stages],
                    patchmatch_num_sample=[int(x) for x in archive.patchmatch_num_sample],
                    propagate_neighbors=[int(s.propagate_neighbors) for s in stages],
                    evaluate_neighbors=[int(s.evaluate_neighbors) for s in stages])

    @classmethod
    def from_scripted_module(cls, path: str) -> "PatchmatchNet":
        """``--input_type module`` (reference eval.py:37-39): the archive carries the reference's own code, which cannot run the HIP
        path -- but also everything needed to build the HIP module: the 242 state-dict tensors under the same names and the
        constructor lists (the command line's patchmatch flags are ignored, as the reference ignores them for a module)."""
        archive = torch.jit.load(path, map_location="cpu")
        model = cls(**cls.scripted_module_config(archive))
        model.load_state_dict({k: v.detach().clone() for k, v in archive.state_dict().items()}, strict=True)
        return model

    def extract_features(self, images: List[torch.Tensor], stacked: Optional[dict] = None,
                         image_table: Optional["ops.SourceTable"] = None) -> List[Dict[int, torch.Tensor]]:
        """Per-view feature pyramids.  ``stacked`` (a dict) additionally receives {stage: [V*B,C,h,w]} when all views
        went through FeatureNet as one batch (view-major), so the caller can change layout in one pass."""
        same = all(im.shape == images[0].shape for im in images)
        B = images[0].shape[0]
        if image_table is not None and not (self.hip_feature_net and same and images[0].is_cuda and self.feature.f16_split):
            raise PmnError("image_table needs the HIP FeatureNet (f16_split) and same-size images on a ROCm device")
        if self.hip_feature_net and same and images[0].is_cuda:
            f = self.feature.forward_hip(images, image_table=image_table)
            if stacked is not None:
                stacked.update({("nhwc", s): t for s, t in f.items()})
            # per-view NCHW-shaped views over the channels-last storage (no copy)
            return [{s: t[i * B:(i + 1) * B].permute(0, 3, 1, 2) for s, t in f.items()} for i in range(len(images))]
        if self.hip_feature_net and images[0].is_cuda:  # views of different sizes: one HIP FeatureNet pass per image
            outs = [self.feature.forward_hip(im) for im in images]
            return [{s: t.permute(0, 3, 1, 2) for s, t in f.items()} for f in outs]
        if self.batch_feature_extraction and same and len(images) > 1:
            f = self.feature(torch.cat(images, dim=0))
            if stacked is not None:
                stacked.update(f)
            return [{s: t[i * B:(i + 1) * B] for s, t in f.items()} for i in range(len(images))]
        return [self.feature(im) for im in images]

    def forward(self, images: List[torch.Tensor], intrinsics: torch.Tensor, extrinsics: torch.Tensor,
                depth_min: torch.Tensor, depth_max: torch.Tensor, noise: Optional[torch.Tensor] = None,
                features: Optional[List[Dict[int, torch.Tensor]]] = None, debug: Optional[dict] = None,
                features_nhwc: Optional[Dict[int, torch.Tensor]] = None, source_tables: Optional[Dict[int, "ops.SourceTable"]] = None,
                ref_nhwc_maps: Optional[Dict[int, torch.Tensor]] = None, image_table: Optional["ops.SourceTable"] = None
                ) -> Tuple[torch.Tensor, torch.Tensor, Dict[int, List[torch.Tensor]]]:
        """Reference arguments (images: N x [B,3,H,W]; intrinsics [B,N,3,3]; extrinsics [B,N,4,4]; depth_min/max [B]).

        Optional extras (default = reference behaviour): ``noise`` [B,48,H/8,W/8] pins the stage-3 random draw,
        ``features`` injects pre-computed FeatureNet outputs (``features_nhwc``: {stage: [(N+1)*B,h,w,C]} = the SAME maps, view-major
        in one channels-last buffer per stage -- what ``features[v][stage]`` are views of -- so that the kernels read them in place
        instead of stacking the source views into a buffer of their own; or ``source_tables`` {stage: ops.SourceTable} +
        ``ref_nhwc_maps`` {stage: [B,h,w,C]}: the source views stay wherever they are and the kernels find them through a device
        table of addresses, pmn_warp_correlate_views), ``image_table`` (ops.SourceTable, shape (N, B, 3, H, W)): FeatureNet reads the
        images through a device table of addresses (pmn_stem_f16s_views) -- ``images`` then only supplies shapes and, entry 0, the
        reference image Refinement reads --, ``debug`` (dict) collects per-stage intermediates.
        Returns (depth [B,1,H,W], photometric confidence [B,H,W] (empty in training mode), {stage: [depths]})."""
        assert len(images) == intrinsics.size()[1], "Different number of images and intrinsic matrices"
        assert len(images) == extrinsics.size()[1], "Different number of images and extrinsic matrices"
        if self.training:
            raise PmnError("patchmatchnet_amd.PatchmatchNet is inference-only: call .eval() (the HIP kernels have no "
                           "backward pass)")
        images, intrinsics, orig_height, orig_width = adjust_image_dims(images, intrinsics)
        ref_image = images[0]
        _, _, ref_height, ref_width = ref_image.size()

        stacked: Dict[int, torch.Tensor] = {}
        if features is None:
            features = self.extract_features(images, stacked, image_table)
        elif features_nhwc is not None:
            stacked.update({("nhwc", st): t for st, t in features_nhwc.items()})
        ref_feature, src_features = features[0], features[1:]
        batch = ref_image.shape[0]

        depth_min = depth_min.float()
        depth_max = depth_max.float()
        device = intrinsics.device
        depth = torch.empty(0, device=device)
        score = torch.empty(0, device=device)
        view_weights = torch.empty(0, device=device)
        depth_patchmatch: Dict[int, List[torch.Tensor]] = {}

        scale = 0.125
        depth_shift, vw_shift = 0, 0
        # relative projections of all stages in one launch (replaces ~40 tiny ATen kernels per forward)
        rel_all = ops.stage_projections(intrinsics, extrinsics, self.stages - 1, scale) if self.hip_projections else None
        for stage in range(self.stages - 1, 0, -1):
            # stage projection matrices (reference models/net.py:225-231): the reference's op sequence is only run when
            # hip_projections is off; otherwise PatchMatch receives rel_proj and never reads ref_proj / src_projs
            if rel_all is None:
                intrinsics_l = intrinsics.clone()
                intrinsics_l[:, :, :2] *= scale
                proj = extrinsics.clone()
                proj[:, :, :3, :4] = torch.matmul(intrinsics_l, extrinsics[:, :, :3, :4])
                proj_l = torch.unbind(proj, 1)
                ref_proj, src_proj = proj_l[0], proj_l[1:]
            else:
                ref_proj, src_proj = extrinsics[:, 0], [extrinsics[:, i] for i in range(1, extrinsics.shape[1])]
            scale *= 2.0

            dbg = [] if debug is not None else None
            pm: PatchMatch = getattr(self, f"patchmatch_{stage}")
            ref_nhwc = src_nhwc = None
            if ("nhwc", stage) in stacked:  # HIP FeatureNet: already channels-last
                allv = stacked[("nhwc", stage)]
            elif stage in stacked:  # one layout pass for all views of the stage
                allv = ops.nchw_to_nhwc(stacked[stage].contiguous())
            else:
                allv = None
            if source_tables is not None:
                ref_nhwc, src_nhwc = ref_nhwc_maps[stage], source_tables[stage]
            elif allv is not None:
                ref_nhwc = allv[:batch]
                src_nhwc = allv[batch:].view(len(src_features), batch, *allv.shape[1:])
            depths, score, view_weights = pm(
                ref_feature=ref_feature[stage], src_features=[f[stage] for f in src_features], ref_proj=ref_proj,
                src_projs=list(src_proj), depth_min=depth_min, depth_max=depth_max, depth=depth,
                view_weights=view_weights, depth_shift=depth_shift, vw_shift=vw_shift,
                noise=noise if stage == self.stages - 1 else None, debug=dbg, ref_nhwc=ref_nhwc, src_nhwc=src_nhwc,
                rel_proj=None if rel_all is None else rel_all[self.stages - 1 - stage])
            if debug is not None:
                debug[stage] = dbg
            depth_patchmatch[stage] = depths
            depth = depths[-1].detach()
            if stage > 1:
                # the nearest x2 up-sampling of depth and view weights (reference :272-275) is folded into the
                # consumers: the next stage reads both maps through a coordinate shift
                depth_shift = 1
                vw_shift = vw_shift + 1 if view_weights.shape[-1] != depths[-1].shape[-1] else 1

        if self.hip_feature_net and ref_image.is_cuda:
            depth = self.upsample_net.forward_hip(ref_image, depth, depth_min, depth_max)
        else:
            depth = self.upsample_net(ref_image, depth, depth_min, depth_max)
        if ref_width != orig_width or ref_height != orig_height:
            depth = F.interpolate(depth, size=[orig_height, orig_width], mode="bilinear", align_corners=False)
        depth_patchmatch[0] = [depth]

        confidence, _ = ops.confidence(score.contiguous(), orig_height, orig_width)
        return depth, confidence, depth_patchmatch


def adjust_image_dims(images: List[torch.Tensor], intrinsics: torch.Tensor
                      ) -> Tuple[List[torch.Tensor], torch.Tensor, int, int]:
    """Resize every image to multiples of 8 and rescale its intrinsics IN PLACE, as reference models/net.py:304-318
    does (callers that reuse ``intrinsics`` / the ``images`` list observe the same mutation)."""
    _, _, ref_height, ref_width = images[0].size()
    for i in range(len(images)):
        _, _, height, width = images[i].size()
        new_height = int(round(height / 8)) * 8
        new_width = int(round(width / 8)) * 8
        if new_width != width or new_height != height:
            intrinsics[:, i, 0] *= new_width / width
            intrinsics[:, i, 1] *= new_height / height
            images[i] = F.interpolate(images[i], size=[new_height, new_width], mode="bilinear", align_corners=False)
    return images, intrinsics, ref_height, ref_width
