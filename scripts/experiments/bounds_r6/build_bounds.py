#!/usr/bin/env python
"""Upper bounds for two items of the round-5 review that were priced "percent-level" without a measurement (wrong results, the same
instruction streams otherwise; built from scratch copies of the sources, never shipped):

    views_nobarrier   gather_corr_kernel<.., MODE_VIEWS, ..>: the one workgroup barrier between the view loop and the SimilarityNet
                      hand-over removed -- what a wave-private hand-over (review item 4, second half) could buy at most
    stem_noconv0      stem_f16s_kernel without conv0's 216 multiply-adds per halo pixel (one LDS read instead of 27)
    stem_nomfma       ... without conv1's MFMAs (the operand loads stay)
    stem_nostore      ... without the output stores
    stem_skeleton     all three: staging, split, LDS traffic, barriers, index math only
                      -- whether a producer / consumer wave split (review item 5) has anything to overlap

    python scripts/experiments/bounds_r6/build_bounds.py          # -> build/ldsab/libpmn_hip_<name>.so
    python scripts/call_ab.py --ops warp_correlate,stem_f16s --libs patchmatchnet_amd/csrc/libpmn_hip.so,build/ldsab/libpmn_hip_views_nobarrier.so,...

Results: profiles/r06_bounds.log."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lds_conflicts"))
import build_ablation as BA  # noqa: E402

BA.variant("views_nobarrier", "gather_corr.hip", [
    ("        __syncthreads();\n        if (!okA) return;\n        float wtot = 1e-5f;",
     "        if (!okA) return;\n        float wtot = 1e-5f;"),
])

NOCONV0 = ("""#pragma unroll 1
            for (int ky = 0; ky < 3; ++ky) {
                const cfloat* wq = cw0 + __builtin_amdgcn_readfirstlane(ky * 72);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci) {
                        const float v = xp[(ci * IW + ky) * XS + kx];
#pragma unroll
                        for (int c = 0; c < 8; ++c) acc[c] = fmaf(v, wq[(kx * 3 + ci) * 8 + c], acc[c]);
                    }
            }
""", """            {
                const float v = xp[0];
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = v * cw0[c];
            }
""")
NOMFMA = ("""#pragma unroll
        for (int t = 0; t < 4; ++t) accM[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ks][0], bh[t], accM[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) accL[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ks][0], blo[t], accL[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) accL[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ks][1], bh[t], accL[t], 0, 0, 0);
""", """#pragma unroll
        for (int t = 0; t < 4; ++t) {
            accM[t][0] += (float)bh[t][0] * (float)wa[ks][0][0];
            accL[t][0] += (float)blo[t][0] * (float)wa[ks][1][0];
        }
""")
NOSTORE = ("                *reinterpret_cast<f32x4_t*>(po + t * rs) = v;", "                if (v[0] == 12345.678f) *reinterpret_cast<f32x4_t*>(po + t * rs) = v;")
BA.variant("stem_noconv0", "conv_f16s.hip", [NOCONV0])
BA.variant("stem_nomfma", "conv_f16s.hip", [NOMFMA])
BA.variant("stem_nostore", "conv_f16s.hip", [NOSTORE])
BA.variant("stem_skeleton", "conv_f16s.hip", [NOCONV0, NOMFMA, NOSTORE])
