#!/usr/bin/env python
"""Upper bound for "fuse conv6+conv7 / conv9+conv10 into one kernel each" (priced at -50 us per forward): what a
fused pair can save is the intermediate map's trip through HBM -- the first layer's stores and the second layer's patch loads.  Two
ablation builds of conv_f16s.hip measure exactly that on the real layers (WRONG results, same instruction streams):

    nostore   the channels-last epilogue's stores are predicated off (a run-time condition that never holds)
    l2input   the patch loads read a 1 MB window of the input (pixel offset & 0x1FFF): every load hits L2

    python scripts/experiments/lds_conflicts/build_fusion_bound.py && python scripts/call_ab.py --ops conv2d_f16s --libs \
        patchmatchnet_amd/csrc/libpmn_hip.so,build/ldsab/libpmn_hip_nostore.so,build/ldsab/libpmn_hip_l2input.so

Results: profiles/r06_conv_fusion_bound.log."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build_ablation as BA  # noqa: E402

BA.variant("nostore", "conv_f16s.hip", [
    ("                        *reinterpret_cast<f32x4_t*>(po + t * rs + 16 * nt) = v;",
     "                        if (a.relu == 12345) *reinterpret_cast<f32x4_t*>(po + t * rs + 16 * nt) = v;"),
])
BA.variant("l2input", "conv_f16s.hip", [
    ("v[k] = *reinterpret_cast<const float4*>(src + ((unsigned)(gy * a.W + gx) * CIN + q4x4));",
     "v[k] = *reinterpret_cast<const float4*>(src + (((unsigned)(gy * a.W + gx) & 0x1FFFu) * CIN + q4x4));"),
])
