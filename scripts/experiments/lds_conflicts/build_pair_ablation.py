#!/usr/bin/env python
"""Where do the 160 us of conv_f16s_pair16_kernel go?  Ablation builds (wrong results, otherwise the same kernel):
    noload   the patch loads are replaced by a constant (no global reads)
    nostore  the final stores are predicated off
    nolds2   layer B's MFMA phase removed (layer A + the in-place split + stores of zeros stay)
python scripts/call_ab.py --ops conv2d_f16s_pair --libs patchmatchnet_amd/csrc/libpmn_hip.so,build/ldsab/libpmn_hip_pair_noload.so,..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build_ablation as BA  # noqa: E402

BA.variant("pair_noload", "conv_f16s.hip", [
    ("            if (idx < TOT && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)\n                v[k] = *reinterpret_cast<const float4*>(src + ((unsigned)(gy * a.W + gx) * C + 4 * q));",
     "            if (idx < TOT && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)\n                v[k] = make_float4(0.25f * gy, 0.5f, 0.125f * gx, 1.0f);"),
])
BA.variant("pair_nostore", "conv_f16s.hip", [
    ("                    *reinterpret_cast<f32x4_t*>(out + (((size_t)n * a.H + oy) * a.W + ox) * C + 4 * kb) = v;",
     "                    if (a.relu == 12345) *reinterpret_cast<f32x4_t*>(out + (((size_t)n * a.H + oy) * a.W + ox) * C + 4 * kb) = v;"),
])
BA.variant("pair_both", "conv_f16s.hip", [
    ("            if (idx < TOT && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)\n                v[k] = *reinterpret_cast<const float4*>(src + ((unsigned)(gy * a.W + gx) * C + 4 * q));",
     "            if (idx < TOT && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)\n                v[k] = make_float4(0.25f * gy, 0.5f, 0.125f * gx, 1.0f);"),
    ("                    *reinterpret_cast<f32x4_t*>(out + (((size_t)n * a.H + oy) * a.W + ox) * C + 4 * kb) = v;",
     "                    if (a.relu == 12345) *reinterpret_cast<f32x4_t*>(out + (((size_t)n * a.H + oy) * a.W + ox) * C + 4 * kb) = v;"),
])
