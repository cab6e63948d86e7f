/* pmn_hip_experimental.h -- NOT part of the product ABI.
 *
 * Declares the one extra entry point of libpmn_hip_experimental.so (`make -C patchmatchnet_amd/csrc EXPERIMENTAL=1`): the
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

#ifdef __cplusplus
}
#endif
#endif /* PMN_HIP_EXPERIMENTAL_H */
