#!/usr/bin/env python
"""Tile variants of the 1/8-resolution FPN level (pmn_conv2d_mfma's 1x1 form, 64 -> 112 channels, 62 us for 127 MB of traffic and
2.6 GFLOP: three times its matrix-pipe time).  Same arithmetic per output (k order unchanged): bits must not change.
    python scripts/experiments/bounds_r6/build_fpn8_variants.py
    python scripts/call_ab.py --ops pointwise_split_mfma --libs patchmatchnet_amd/csrc/libpmn_hip.so,build/ldsab/libpmn_hip_fpn8_nw2.so,..."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lds_conflicts"))
import build_ablation as BA  # noqa: E402

OLD = "            return launch_mfma<64, 64, 128, 1, 1, 1, 4, 1, 4, false>(in, weights, shift, out, out_b, a, st);"
for name, args in (("fpn8_nw2", "64, 64, 128, 1, 1, 1, 2, 1, 4"), ("fpn8_nw8", "64, 64, 128, 1, 1, 1, 8, 1, 4"),
                   ("fpn8_cc32", "64, 32, 128, 1, 1, 1, 4, 1, 4"), ("fpn8_nw2_cc32", "64, 32, 128, 1, 1, 1, 2, 1, 4"),
                   ("fpn8_pg2", "64, 64, 128, 1, 1, 1, 2, 2, 4"), ("fpn8_cc32_d2", "64, 32, 128, 1, 1, 1, 4, 1, 2"),
                   ("fpn8_cc16_d2", "64, 16, 128, 1, 1, 1, 4, 1, 2"), ("fpn8_cc16_d1", "64, 16, 128, 1, 1, 1, 4, 1, 1")):
    BA.variant(name, "conv_mfma.hip", [(OLD, OLD.replace("64, 64, 128, 1, 1, 1, 4, 1, 4", args))])
