"""Research build only (PMN_EXPERIMENTAL=1, libpmn_hip_experimental.so): rounds 1-2's fp32 alternatives for single layers, kept for
A/B measurements and their parity tests.  The product never imports this module; net.py / patchmatch.py only carry the two hooks it
installs (FeatureNet.layer_hook, PatchMatch.heads_hook), both None by default.

    from patchmatchnet_amd import research
    research.install(model, winograd=True, winograd5=True, mfma_convs=True, mfma_offset_heads=True)   # with model.feature.f16_split = False
"""
from __future__ import annotations

import torch

from . import _lib, ops, params
from ._lib import PmnError


def install(model, winograd: bool = False, winograd5: bool = False, mfma_convs: bool = False, mfma_offset_heads: bool = False) -> None:
    if not _lib.experimental():
        raise PmnError("patchmatchnet_amd.research needs the research build: make -C patchmatchnet_amd/csrc EXPERIMENTAL=1 and "
                       "PMN_EXPERIMENTAL=1")
    feature = model.feature
    packed = {}

    def bn_of(m):
        return (m.bn.weight, m.bn.bias, m.bn.running_mean, m.bn.running_var)

    def layer_hook(i, t):
        m = getattr(feature, f"conv{i}")
        cv = m.conv
        k, s_, cin, cout = cv.kernel_size[0], cv.stride[0], cv.in_channels, cv.out_channels
        dev = t.device

        def pack(kind, fn):
            key = (i, kind)
            if key not in packed:
                w, sh = fn(cv.weight, bn=bn_of(m), eps=m.bn.eps)
                packed[key] = (torch.from_numpy(w).to(dev), torch.from_numpy(sh).to(dev))
            return packed[key]

        if winograd and k == 3 and s_ == 1 and cin == cout and cin in (16, 32, 64):  # Winograd F(2x2,3x3), fp32 matrix cores
            return ops.conv3x3_wino(t, *pack("wino", params.pack_conv_wino), relu=True)
        if winograd5 and k == 5 and s_ == 2 and (cin, cout) in ((8, 16), (16, 32), (32, 64)):  # four Winograd sub-convolutions
            return ops.conv5x5s2_wino(t, *pack("wino5", params.pack_conv5x5s2_wino), relu=True)
        if mfma_convs and (cin, cout, k, s_) in ops.MFMA_CONV_SHAPES:  # fp32 implicit GEMM on the matrix cores
            return ops.conv2d_mfma(t, *pack("mfma", params.pack_conv_mfma), k, s_, cv.padding[0], relu=True)
        return None

    feature.layer_hook = layer_hook
    if mfma_offset_heads:
        heads = {}

        def heads_hook(pm, ref_nhwc, propagate_any):
            if (pm.eval_conv.in_channels, pm.dilation) not in ops.MFMA_HEAD_SHAPES:
                return None
            mods = (pm.propa_conv, pm.eval_conv) if propagate_any else (pm.eval_conv,)
            key = (pm.stage, propagate_any)
            if key not in heads:
                wcat = torch.cat([m.weight.detach() for m in mods], 0)
                bcat = torch.cat([m.bias.detach() for m in mods], 0)
                if wcat.shape[0] > 64:
                    return None
                w, sh = params.pack_conv_mfma(wcat, bias=bcat)
                heads[key] = (torch.from_numpy(w).to(ref_nhwc.device), torch.from_numpy(sh).to(ref_nhwc.device))
            n_p, n_e = (2 * pm.propagate_neighbors if propagate_any else 0), 2 * pm.evaluate_neighbors
            a_, b_ = ops.offset_heads_mfma(ref_nhwc, *heads[key], n_p + n_e, n_p if propagate_any else n_e, pm.dilation)
            return (a_, b_) if propagate_any else (None, a_)

        for stage in (1, 2, 3):
            pm = getattr(model, f"patchmatch_{stage}")
            pm.f16_split = False
            pm.heads_hook = heads_hook
