/* pmn_hip_experimental.h -- NOT part of the product ABI.
 *
 * Declares the extra entry points of libpmn_hip_experimental.so (`make -C patchmatchnet_amd/csrc EXPERIMENTAL=1`): the
 * product library libpmn_hip.so neither contains the kernels this selects nor exports the symbol.  The experimental build
 * carries the three LDS-window research families of pmn_warp_correlate (patchmatchnet_amd/csrc/experimental/), kept as a
 * bit-identical, measured record of why the streaming kernel is the product (DESIGN.md lessons 16-19, 23), and round 4's
 * matrix-core formulation (corr_mfma.hip, lesson 31).
 * No reference counterpart (the reference, models/module.py:130-181 + models/patchmatch.py:193-217, has one formulation).
 */
#ifndef PMN_HIP_EXPERIMENTAL_H
#define PMN_HIP_EXPERIMENTAL_H
#include "pmn_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Process-wide tuning of pmn_warp_correlate (diagnostics / benchmarking; defaults are the measured best):
 *   key 1: kernel family.  bit 0 = windowed kernels where they cover the shape (0 = always the streaming kernels of
 *          gather_corr.hip); bit 4 = the first windowed form (gather_win.hip) instead of the lane = item engine
 *          (gather_lane.hip); bits 2 / 3 = keep the streaming kernel for the PixelwiseNet / the known-weights launches;
 *          bit 1 = gather_win.hip without the per-lane channel-quad rotation (bank-conflict A/B);
 *          bit 5 = the tile-window kernel (gather_tile.hip) for the known-weights launches;
 *          bit 6 = the round-4 matrix-core formulation (corr_mfma.hip: correlate on fp32 MFMAs, then interpolate; same taps and
 *                  weights, re-associated sums -- agrees with the streaming kernel to ~1e-6, tests/test_corr_mfma.py);
 *   key 10: bytes of one of its two window buffers (multiple of 1024, 16384..65536);
 *   key 4: bytes of LDS each wave of gather_lane.hip may use for its source-map window (multiple of 1024, 1024..36864);
 *   key 6: gather_lane.hip build, 3 (168 registers, 3 waves per SIMD) or 2 (256 registers);
 *   keys 0, 2: window bytes of gather_win.hip (known-weights / PixelwiseNet kernels); keys 3, 5: timing ablations of
 *          gather_win.hip / gather_lane.hip (non-zero values skip work: results are then meaningless).
 * The three LDS-window families compute bit-identical results (tests/test_gather_win.py).  Not thread-safe against concurrent launches. */
int pmn_set_tuning(int key, int value);

/* ---- rounds 1-2's fp32 FeatureNet alternatives (product entry points until ABI 16; no default path has used them since round 3) ---- */

/* Winograd F(2x2,3x3) form of FeatureNet's 3x3 stride-1 ConvBnReLU layers with cin == cout == C in {16, 32, 64} (conv3/4, conv6/7,
 * conv9/10; reference models/net.py:21-31) on the fp32 matrix cores (v_mfma_f32_16x16x4_f32): 16 instead of 36 multiplies per 2x2
 * output tile and input channel, same fp32 error level as the direct form (scripts/winograd_study.py).  in / out [N,H,W,C]
 * channels-last; weights DEVICE float [C/16][16][C/16][64][4] = G g G^T with the BatchNorm scale folded in, computed in float64 and
 * laid out in matrix-operand lane order (patchmatchnet_amd/params.py: pack_conv_wino); shift DEVICE float[C]. */
int pmn_conv3x3_wino(const float *in, const float *weights, const float *shift, float *out, int N, int H, int W, int C, int relu,
                     void *stream);

/* Winograd form of FeatureNet's 5x5 stride-2 ConvBnReLU layers conv2 (8->16), conv5 (16->32), conv8 (32->64) (reference
 * models/net.py:20, 24, 28): the stride-2 convolution is split into four stride-1 convolutions on the parity sub-images (3x3, 3x2, 2x3,
 * 2x2 taps), each in minimal-filtering form F(2,3) / F(2,2) per dimension: 49 instead of 100 multiplies per 2x2 output tile and input
 * channel, on v_mfma_f32_16x16x4_f32.  in [N,H,W,cin] channels-last; weights DEVICE float [cin/8][49][cout/16][64][2]
 * (patchmatchnet_amd/params.py: pack_conv5x5s2_wino, transforms in float64, BatchNorm scale folded in); shift DEVICE float[cout];
 * out [N,(H-1)/2+1,(W-1)/2+1,cout].  Supported (cin,cout): (8,16), (16,32), (32,64). */
int pmn_conv5x5s2_wino(const float *in, const float *weights, const float *shift, float *out, int N, int H, int W, int cin,
                       int cout, int relu, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PMN_HIP_EXPERIMENTAL_H */
