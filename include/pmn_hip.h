/*
 * pmn_hip.h -- C ABI of libpmn_hip.so: the MI355X (gfx950) kernels behind PatchmatchNet's learned-PatchMatch
 * hot path.  This is the drop-in boundary: plain pointers and sizes, no torch types.
 *
 * The reference (FangjinhuaWang/PatchmatchNet) is pure Python over PyTorch and has no FFI of its own; each entry
 * point below replaces the PyTorch op sequence at the cited reference lines.  INTEGRATION.md shows the ctypes
 * binding a reference maintainer would add to models/patchmatch.py / models/module.py.
 *
 * Contract common to every entry point
 *   - all tensor arguments are DEVICE pointers to contiguous float32 buffers owned (and pre-allocated) by the
 *     caller; the library never allocates, never synchronises, and launches on `stream` (a hipStream_t; pass the
 *     caller's current stream, NULL = default stream);
 *   - arguments named *_host are small HOST arrays (neighbour tables) copied into the kernel-argument segment;
 *   - returns PMN_OK (0) or a negative PMN_ERR_* code; nothing is launched when an argument check fails;
 *   - entry points never block: each one only ENQUEUES kernels (no hip*Synchronize, no hipMalloc / hipFree, no hipMemcpy /
 *     hipMemset, no stream or event waits), so a call costs microseconds, a whole forward is HIP-graph capturable, and a binding
 *     may call in without releasing its interpreter lock (patchmatchnet_amd/_lib.py uses ctypes.PyDLL);
 *     tests/test_abi.py::test_entry_points_never_block enforces it on the sources;
 *   - entry points are re-entrant; the only process state is a cache of per-(kernel, device) launch attributes.
 *
 * Layouts
 *   feature maps    channels-last  [B, h, w, C]           (source views stacked: [N, B, hs, ws, C])
 *   hypotheses      [B, D, h, w]   (D <= PMN_MAX_DEPTH)   depth_sample of the reference
 *   per-view weight [B, N, h, w]
 *   offsets         [B, 2K, h, w]  output of the reference's propa_conv / eval_conv (channel 2k -> x, 2k+1 -> y)
 *   neighbour table host int[2K]   (dy,dx) pairs, reference models/patchmatch.py:331-392
 *   MLP block       DEVICE float[PMN_MLP_FLOATS], BatchNorm folded, packed once per model by
 *                   patchmatchnet_amd/params.py: 16 records of 20 floats, one per hidden unit j
 *                   { w0[j][0..7] (first G used) | w1[0..7][j] | t0[j] | 3 pad }, then t1[8] | w2[8] | b2 | 3 pad
 */
#ifndef PMN_HIP_H
#define PMN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMN_ABI_VERSION 22
#define PMN_MLP_FLOATS 340
#define PMN_MAX_DEPTH 64
#define PMN_MAX_NEIGHBORS 17
#define PMN_MAX_FUSE_SRC 32

#define PMN_OK 0
#define PMN_ERR_ARG (-1)     /* null pointer / size out of range */
#define PMN_ERR_SHAPE (-2)   /* unsupported channel / group / neighbour / hypothesis count */
#define PMN_ERR_LAUNCH (-3)  /* hipGetLastError() after the launch */

/* ABI version of the loaded library (== PMN_ABI_VERSION of the header it was built from). */
int pmn_abi_version(void);

/* Human-readable text for a PMN_ERR_* code. */
const char *pmn_error_string(int code);

/* [B,C,h,w] -> [B,h,w,C].  Feeds FeatureNet outputs (reference models/net.py:203-208) to the kernels below.
 * `out` may be a slot of the stacked source buffer. */
int pmn_nchw_to_nhwc(const float *in, float *out, int B, int C, int h, int w, void *stream);

/* FeatureWeightNet.forward (reference models/patchmatch.py:603-624) fused with get_grid (:396-426):
 * gather the K evaluation neighbours of every reference pixel (bilinear, border, align_corners=False on a grid
 * normalised with (size-1)/2), group-wise correlation with the centre feature, MLP, sigmoid.
 * out_feature_weight [B,K,h,w].  C in {16,32,64}, G in {4,8} with C/G in {4,8}, K in {9,17}. */
int pmn_feature_weight(const float *ref_nhwc, const float *eval_offsets, const int *eval_table_host,
                       const float *mlp, int B, int C, int G, int K, int h, int w,
                       float *out_feature_weight, void *stream);

/* DepthInitialization.forward + Propagation.forward (reference models/patchmatch.py:53-94, 115-124), fused:
 *   noise != NULL : first iteration on the coarsest stage; 48 hypotheses from the caller's torch.rand draw
 *                   noise [B,48,h,w] (the RNG stays with the caller so it matches the reference stream);
 *   noise == NULL : num_sample hypotheses around `depth` ([B,1,h>>depth_shift,w>>depth_shift]; depth_shift=1
 *                   reads the previous stage's map through the nearest x2 up-sampling of net.py:274);
 *                   num_sample == 1 passes depth through.
 *   K > 0         : the centre hypothesis (index D0/2) is gathered at the K propagation neighbours, appended, and
 *                   the D = D0 + K values are sorted ascending per pixel.
 * depth_min/depth_max: device float[B].  PRECONDITION of every entry point that takes them or depth hypotheses: finite,
 * 0 < depth_min < depth_max, previous depths finite and > 0 -- the kernels' divisions are hipcc's IEEE fma sequence without its
 * v_div_scale / v_div_fixup shell (csrc/pmn_common.hpp), bit-identical for normal operands; a zero or infinite divisor (a degenerate
 * range) yields NaN where the reference's division yields inf.  eval.py refuses such samples before upload (its _check_depth_range).
 * Outputs: depth_sample [B,D,h,w] and its normalised inverse depth
 * xnorm = (1/d - 1/dmax)/(1/dmin - 1/dmax) (reference :655-657), stored HYPOTHESIS-LAST [B,h,w,D] for
 * pmn_aggregate_regress (its only consumer: one bilinear corner of a neighbour = D contiguous floats). */
int pmn_init_hypotheses(const float *noise, const float *depth, int depth_shift, const float *depth_min,
                        const float *depth_max, int num_sample, float interval_scale, const float *propa_offsets,
                        const int *propa_table_host, int K, int B, int h, int w, float *depth_sample,
                        float *xnorm, void *stream);

/* The fused hot kernel.  Evaluation.forward up to and including SimilarityNet's pointwise MLP
 * (reference models/patchmatch.py:192-217, 570; differentiable_warping models/module.py:130-181;
 * PixelwiseNet :695-702):  for every source view: homography warp of the D hypotheses, bilinear (zeros,
 * align_corners=True) gather of the source features, group-wise correlation with the reference feature;
 * view weights either read from view_weights_in ([B,N,h>>vw_shift,w>>vw_shift]; vw_shift = 1 / 2 reads a
 * coarser stage's map through the nearest x2 / x4 up-sampling of net.py:275) or -- view_weights_in == NULL -- computed by PixelwiseNet (max over D of the
 * sigmoid response) and written to view_weights_out [B,N,h,w] (+ optional arg-max index vw_argmax_out);
 * weighted aggregation over views; SimilarityNet MLP -> cost_out, stored HYPOTHESIS-LAST [B,h,w,D] (consumed only by
 * pmn_aggregate_regress).
 * rel_proj [B,N,4,4] = src_proj @ inverse(ref_proj).  similarity_out (optional) receives the aggregated
 * similarity [B,G,D,h,w] (the tensor the reference materialises at :217).
 * The warped volume [B,C,D,h,w] is never materialised. */
int pmn_warp_correlate(const float *ref_nhwc, const float *src_nhwc, const float *rel_proj,
                       const float *depth_sample, const float *view_weights_in, int vw_shift,
                       const float *similarity_mlp, const float *pixelwise_mlp, int B, int N, int C,
                       int G, int D, int h, int w, int hs, int ws, float *cost_out, float *view_weights_out,
                       int *vw_argmax_out, float *similarity_out, void *stream);

/* The same, with the source views handed over as a DEVICE table of N addresses (64-bit each) of per-view [B,hs,ws,C] channels-last maps
 * instead of one stacked [N,B,hs,ws,C] tensor: the maps stay where their producer left them.  eval.py's per-scan feature cache uses it
 * (every view's pyramid is encoded once and read by ~num_views samples; stacking would copy 0.5 GB per 1600x1200 sample), the table
 * itself is a static device buffer a HIP graph can keep reading while the host rewrites it between replays.  Reference: the list
 * `src_features` of models/patchmatch.py:179-191 -- a list of separately allocated tensors there too. */
int pmn_warp_correlate_views(const float *ref_nhwc, const void *src_view_table, const float *rel_proj,
                             const float *depth_sample, const float *view_weights_in, int vw_shift,
                             const float *similarity_mlp, const float *pixelwise_mlp, int B, int N, int C,
                             int G, int D, int h, int w, int hs, int ws, float *cost_out, float *view_weights_out,
                             int *vw_argmax_out, float *similarity_out, void *stream);

/* Adaptive spatial cost aggregation + softmax + regression: depth_weight (reference models/patchmatch.py:650-669),
 * weight normalisation (:509-510), SimilarityNet's neighbour gather and weighted sum (:569-577),
 * exp(log_softmax) (:221) and depth regression (:226-237).
 * cost and xnorm are hypothesis-last [B,h,w,D] (as pmn_warp_correlate / pmn_init_hypotheses write them); depth_sample
 * [B,D,h,w]; score_out [B,D,h,w] = probabilities, depth_out [B,h,w].  is_inverse selects the inverse-depth regression. */
int pmn_aggregate_regress(const float *cost, const float *depth_sample, const float *xnorm,
                          const float *feature_weight, const float *eval_offsets, const int *eval_table_host,
                          int K, float interval_scale, int is_inverse, int B, int D, int h, int w,
                          float *score_out, float *depth_out, void *stream);

/* Photometric-confidence epilogue (reference models/net.py:288-299): 4-wide window sum of the stage-1
 * probabilities around trunc(sum_d d*p_d), nearest resize to [B,H,W].  depth_index_out (optional) [B,h,w]. */
int pmn_confidence(const float *score, int B, int D, int h, int w, int H, int W, float *confidence_out,
                   int *depth_index_out, void *stream);

/* Direct fp32 convolution with fused epilogue for the small-channel CNNs around the hot path: FeatureNet's ConvBnReLU
 * stack and FPN head (reference models/net.py:17-67), the offset heads propa_conv / eval_conv
 * (models/patchmatch.py:288-311, 467-471).  out = [relu]( conv(in, W) + shift [+ bilinear_x2(up)] ).
 *   in       [N,H,W,cin] channels-last, or [N,cin,H,W] when in_nchw (cin = 3 or 1: the image / depth planes)
 *   weights  DEVICE float [K][K][cin][coutp], coutp = cout rounded up to 8 (cout <= 8) or a multiple of 16, BatchNorm scale
 *            folded in (patchmatchnet_amd/params.py: pack_conv); shift DEVICE float[coutp] (folded BN shift or conv bias)
 *   up       optional [N,up_h,up_w,cout] map, bilinearly up-sampled x2 (align_corners=False) and added (net.py:60,65)
 *   out      [N,Ho,Wo,cout] channels-last, or planar [N,cout,Ho,Wo] when out_nchw (the offset planes the PatchMatch
 *            kernels read).  Supported (cin,K,stride): (3|1,3,1 nchw-in), (8|16|32|64,3,1), (8|16|32,5,2), (16|32|64,1,1). */
int pmn_conv2d(const float *in, const float *weights, const float *shift, const float *up, float *out, int N, int H,
               int W, int cin, int cout, int K, int stride, int pad, int dil, int relu, int in_nchw, int out_nchw,
               int up_h, int up_w, void *stream);

/* Fused tail of FeatureNet's FPN (reference models/net.py:64-67): out = output3( bilinear_x2(up) + inner2(x) ) without
 * materialising the 64-channel half-resolution map.  x [N,H,W,16], up [N,H/2,W/2,64], w_in [16][64] / b_in [64] and
 * w_out [64][16] in pack_conv layout (device) -> out [N,H,W,16]. */
int pmn_fpn_tail(const float *x, const float *up, const float *w_in, const float *b_in, const float *w_out, float *out,
                 int N, int H, int W, int cin, int cmid, int cout, void *stream);

/* The same convolution (+ folded BatchNorm shift / bias, optional ReLU) as fp32 implicit GEMM on the matrix cores
 * (v_mfma_f32_32x32x2_f32: exact fp32, bitwise a k-ordered fmaf chain).  in [N,H,W,cin] channels-last; weights DEVICE float
 * [K*K][cin/8][coutp/32][64][4] with coutp = cout rounded up to 32 (patchmatchnet_amd/params.py: pack_conv_mfma); shift
 * DEVICE float[coutp].
 *   planar == 0  the 1x1 form (cin 64, cout <= 128, K 1, stride 1, pad 0): channels [0,ca) -> out [N,H,W,ca], [ca,cout) -> out_b
 *                [N,H,W,cout-ca] -- the 1/8-resolution level of the folded FPN head (reference models/net.py:36-70), same arithmetic
 *                as pmn_fpn_level.  This is the ONLY shape of the product library since round 4 (ABI 17): everything else returns
 *                PMN_ERR_SHAPE.  The research build (pmn_hip_experimental.h) additionally keeps rounds 1-2's forms: FeatureNet's wide
 *                layers (cin,cout,K,stride) = (64,64,3,1), (32,32,3,1), (32,64,5,2), (16,32,5,2) with out [N,Ho,Wo,cout], and
 *   planar == 1  the offset heads of one stage as ONE dilated 3x3 convolution ((cin,dil) = (64,2), (32,4), (16,6), cout <= 64;
 *                channels [0,ca) -> out [N,ca,Ho,Wo], the rest -> out_b, planar) -- superseded by pmn_conv2d_f16s /
 *                pmn_offset_heads_f16s in round 3. */
int pmn_conv2d_mfma(const float *in, const float *weights, const float *shift, float *out, float *out_b, int N, int H, int W,
                    int cin, int cout, int ca, int K, int stride, int pad, int dil, int relu, int planar, void *stream);

/* ---- fp16-split entry points (pmn_conv2d_f16s, pmn_offset_heads_f16s, pmn_stem_f16s, pmn_refine_fused): accepted magnitudes ----
 * Every activation and weight x is split as hi = fp16(x), lo = fp16((x - hi) * 2048) and the product sum is taken as
 * hi*hi + (hi*lo + lo*hi) / 2048 with fp32 accumulation.  That reproduces an fp32 convolution (2-4e-7 of the output scale) for
 *     6.1e-5 <= |x| < 65504   (fp16's normal range: 22 significant bits per factor);
 * below 6.1e-5 `hi` is a subnormal (absolute error <= 3e-8 per factor -- irrelevant next to factors of normal size; an operand
 * tensor that is ENTIRELY below 1e-6 loses relative precision); at |x| >= 65504 `hi` becomes +-inf and the output is inf / NaN where
 * an fp32 convolution is finite.  The kernels do not check (opt-in: pmn_check_f16_domain, PMN_CHECK_F16_DOMAIN=1).  WEIGHTS are checked on the host when they are packed
 * (patchmatchnet_amd/params.py raises F16DomainError for a BatchNorm-folded weight outside (-65504, 65504), and the modules then use
 * the fp32 kernels pmn_stem / pmn_conv2d / pmn_refine_front + pmn_refine_tail and say so); ACTIVATIONS are the caller's contract:
 * images in [0, 1] (or any range below 6e4) and the reference checkpoint keep every intermediate below 1e2.
 * tests/test_hip_parity.py::test_f16_split_domain documents both ends of the range on the GPU. */

/* FeatureNet's ConvBnReLU layers conv2..conv10 (reference models/net.py:20-31: 3x3 stride 1 with cin == cout in {16,32,64}; 5x5
 * stride 2 with (cin,cout) in {(8,16),(16,32),(32,64)}) on the FP16 matrix cores with SPLIT operands: every activation and every
 * weight is x = hi + lo/2048 (hi = fp16(x), lo = fp16((x-hi)*2048): 22 significant bits), and sum x*w is evaluated as
 * sum hi*hi + (sum hi*lo + sum lo*hi)/2048 -- three v_mfma_f32_16x16x32_f16 per k-step with fp32 accumulation, 16/3 the rate of
 * v_mfma_f32_16x16x4_f32, the error of an fp32 direct convolution (2-4e-7 of the output scale; scripts/fp16_split_study.py).
 * in [N,H,W,cin] channels-last float32; weights DEVICE float16 [cin/CC][k-steps][cout/16][2][64][8] (hi | lo B operands in lane
 * order, BatchNorm scale folded in float64: patchmatchnet_amd/params.py pack_conv_f16s); shift DEVICE float[cout]; out
 * [N,(H-1)/stride+1,(W-1)/stride+1,cout] float32 (padding k/2). */
int pmn_conv2d_f16s(const float *in, const void *weights, const float *shift, float *out, int N, int H, int W, int cin, int cout,
                    int k, int stride, int relu, void *stream);

/* ABI 22.  Two consecutive 3x3 / stride-1 / 16 -> 16 ConvBnReLU layers in ONE launch (FeatureNet conv3 + conv4 at half resolution,
 * reference models/net.py:21-22, 52): the 184 MB intermediate map of six 1600x1200 views never goes to HBM (the first layer is
 * evaluated on each 14 x 14 tile's 16 x 16 halo region and kept in LDS as split fp16 planes).  in / out [N,H,W,16] channels-last
 * float32; weights_* / shift_* exactly as pmn_conv2d_f16s takes them for a (k 3, stride 1, 16 -> 16) layer; relu applies to both
 * layers.  Bit-identical to two pmn_conv2d_f16s calls.  channels != 16: PMN_ERR_SHAPE (the other layer pairs of FeatureNet do not
 * pay: profiles/r06_conv_fusion_bound.log). */
int pmn_conv2d_f16s_pair(const float *in, const void *weights_a, const float *shift_a, const void *weights_b, const float *shift_b,
                         float *out, int N, int H, int W, int channels, int relu, void *stream);

/* The offset heads of one PatchMatch stage -- propa_conv rows first, then eval_conv (reference models/patchmatch.py:288-311, :467, :471) --
 * as ONE dilated 3x3 convolution with bias on the fp16 matrix cores with split operands (see pmn_conv2d_f16s), planar outputs.
 * in [N,H,W,cin] channels-last; weights DEVICE float16 [cin/16][k-steps][coutp/16][2][64][8] with coutp = cout rounded up to 16
 * (patchmatchnet_amd/params.py pack_offset_heads_f16s); shift DEVICE float[coutp]; out_a [N,ca,H,W] = channels [0,ca), out_b
 * [N,cout-ca,H,W] = the rest (NULL when ca == cout).  Supported (cin, dilation): (64,2), (32,4), (16,6) -- the reference's three stages
 * -- with cout rounded up to 32, 48 or 64; PMN_ERR_SHAPE otherwise (the caller then uses pmn_conv2d). */
int pmn_offset_heads_f16s(const float *in, const void *weights, const float *shift, float *out_a, float *out_b, int N, int H, int W,
                          int cin, int cout, int ca, int dil, void *stream);

/* One level of FeatureNet's FPN head in FOLDED form (reference models/net.py:57-67).  The head is linear (1x1 convolutions,
 * bilinear x2 up-sampling, sums), so output_k(upsample(intra) + inner_k(conv)) is evaluated as
 *   out[c] = bilinear_x2(u)[c] + b[c] + sum_ci x[ci] * w[ci][c]
 * with w = output . inner and b = output . bias_inner formed in float64 on the host (patchmatchnet_amd/params.py: fold_fpn):
 *   1/8:  [feature3 | u8] = W8 conv10          (cin 64, cout 112, ca 64, u = NULL)
 *   1/4:  [feature2 | u4] = up(u8) + W4 conv7   (cin 32, cout 48,  ca 32)
 *   1/2:   feature1       = up(u4) + W2 conv4   (cin 16, cout 16,  ca 16)
 * x [N,H,W,cin]; u [N,H/2,W/2,cout] or NULL; w DEVICE float [cin][cout]; b DEVICE float [cout]; channels [0,ca) go to
 * out_a [N,H,W,ca], the remaining cout-ca to out_b [N,H,W,cout-ca] (NULL when ca == cout). */
int pmn_fpn_level(const float *x, const float *u, const float *w, const float *b, float *out_a, float *out_b, int N, int H,
                  int W, int cin, int cout, int ca, void *stream);

/* ConvTranspose2d(8, 8, k=3, stride=2, padding=1, output_padding=1, bias=False) + BatchNorm + ReLU of the Refinement net
 * (reference models/net.py:86-88, 114).  in [N,Hi,Wi,8]; weights DEVICE float [3][3][8][8] ([ky][kx][ci][co], BatchNorm scale
 * folded in: patchmatchnet_amd/params.py pack_deconv); shift DEVICE float[8] -> out [N,2Hi,2Wi,8]. */
int pmn_deconv3x3s2(const float *in, const float *weights, const float *shift, float *out, int N, int Hi, int Wi, int cin,
                    int cout, int relu, void *stream);

/* Full-resolution half of the Refinement network (reference models/net.py:73-126) in two launches.
 * pmn_refine_front: x16 = cat( relu(bn(deconv(t2))), conv0(img) ) (net.py:110-117): img [B,3,H,W] planar, t2 [B,H/2,W/2,8]
 *   channels-last (conv2 output); w0 [3][3][3][8] / s0 [8] = pack_conv of conv0, wd [3][3][8][8] / sd [8] = pack_deconv of
 *   deconv + bn (DEVICE) -> x16 [B,H,W,16].  H, W even.
 * pmn_refine_tail: depth = (nearest_x2(dnorm) + res(conv3(x16))) * (depth_max - depth_min) + depth_min (net.py:117-122):
 *   w3 [2][3][3][16][4] / s3 [8] and wr [3][3][8] from patchmatchnet_amd/params.py pack_refine_tail (DEVICE); dnorm
 *   [B,1,H/2,W/2] normalised input depth; depth_min / depth_max DEVICE float[B] -> out [B,1,H,W]. */
int pmn_refine_front(const float *img, const float *t2, const float *w0, const float *s0, const float *wd, const float *sd,
                     float *x16, int B, int H, int W, void *stream);
int pmn_refine_tail(const float *x16, const float *w3, const float *s3, const float *wr, const float *dnorm,
                    const float *depth_min, const float *depth_max, float *out, int B, int H, int W, void *stream);

/* The same half in ONE launch (what Refinement.forward_hip runs): pmn_refine_front's and pmn_refine_tail's arguments without the x16
 * buffer between them, and conv3 on the fp16 matrix cores with split operands -- w3a DEVICE float16 [5][2][64][8] / s3 [8] from
 * patchmatchnet_amd/params.py pack_refine_conv3_f16s (conv3's weights as MFMA A operands, hi | lo, BatchNorm folded).  The x16
 * intermediate (123 MB per 1600x1200 depth map) stays in LDS.  H, W even. */
int pmn_refine_fused(const float *img, const float *t2, const float *w0, const float *s0, const float *wd, const float *sd,
                     const void *w3a, const float *s3, const float *wr, const float *dnorm, const float *depth_min,
                     const float *depth_max, float *out, int B, int H, int W, void *stream);

/* Relative projections of every (stage, batch element, source view): rel = P_src @ inverse(P_ref) with
 * P = [[K_s @ E[:3,:4]], [E[3,:]]] and K_s = K with rows 0,1 scaled by scale0 * 2^stage (reference models/net.py:225-231,
 * models/module.py:148).  intrinsics [B,V,3,3], extrinsics [B,V,4,4] (view 0 = reference) -> rel [nstages,B,V-1,4,4]. */
int pmn_stage_projections(const float *intrinsics, const float *extrinsics, int B, int V, int nstages, float scale0,
                          float *rel, void *stream);

/* Fused full-resolution stem of FeatureNet: conv0 (3->8) + conv1 (8->8), each 3x3 + BatchNorm + ReLU (reference
 * models/net.py:17-19, 51).  img [N,3,H,W] planar; w0 [3][3][3][8] / s0 [8], w1 [3][3][8][8] / s1 [8] (pack_conv layout, device)
 * -> out [N,H,W,8] channels-last. */
int pmn_stem(const float *img, const float *w0, const float *s0, const float *w1, const float *s1, float *out, int N, int H,
             int W, void *stream);

/* The same stem with conv1 (72 % of its multiplies) on the FP16 matrix cores with split operands (see pmn_conv2d_f16s): conv0 on the
 * fp32 VALU into LDS, split into hi / lo fp16 planes there, conv1 as three v_mfma_f32_16x16x32_f16 per k-step with the output channels
 * as MFMA rows.  w1a DEVICE float16 [3][2][64][8] (patchmatchnet_amd/params.py pack_stem_conv1_f16s); everything else as pmn_stem. */
int pmn_stem_f16s(const float *img, const float *w0, const float *s0, const void *w1a, const float *s1, float *out, int N, int H,
                  int W, void *stream);

/* pmn_stem_f16s for `views` separately allocated images of one size (reference models/net.py:203-208: FeatureNet runs per image of
 * the `images` list): img_table DEVICE array of `views` addresses, entry v = a dense [B,3,H,W] float32 tensor, every address 16-byte
 * aligned (caller's contract); out [views*B,H,W,8] view-major.  The table is read when the kernel runs: a captured launch follows
 * whatever the table holds at replay time, so a HIP graph reads each sample's images in place (patchmatchnet_amd/graph.py). */
int pmn_stem_f16s_views(const float *const *img_table, int views, const float *w0, const float *s0, const void *w1a,
                        const float *s1, float *out, int B, int H, int W, void *stream);

/* Stand-alone differentiable_warping (reference models/module.py:130-181) for API completeness and unit
 * parity: src_nchw [B,C,hs,ws], rel_proj [B,4,4], depth [B,D,h,w] -> warped [B,C,D,h,w].  Not on the fast path. */
int pmn_differentiable_warping(const float *src_nchw, const float *rel_proj, const float *depth, int B, int C,
                               int D, int h, int w, int hs, int ws, float *warped, void *stream);

/* Photometric + geometric consistency filtering of ONE reference view against its n_src source views and the fused world points
 * (reference eval.py:86-190 reproject_with_depth / check_geometric_consistency, :207-281 filter_depth), one launch, a thread per
 * reference pixel.  maps: the per-scan buffer [V][2][H][W] (slot v = depth, confidence of view v -- what the per-scan all-gather
 * leaves on every rank), slot_stride floats between slots; ref_slot and src_slots_host[n_src] (HOST ints, n_src <=
 * PMN_MAX_FUSE_SRC) index it.  H x W is the REFERENCE view's size; src_hw_host (HOST int[2*n_src]: height, width of every source
 * view's maps; NULL = all H x W) gives every source map its own size, as the reference reads every view's file at its own size
 * (eval.py:203-237; --image_max_dim on a scan with mixed image sizes): slot v holds depth [h_v][w_v] then confidence [h_v][w_v]
 * packed at the start of the slot, slot_stride >= 2*h*w of the largest view.  mats (DEVICE float32, built by patchmatchnet_amd/fusion.py with numpy's own
 * float32 inverse / matmul so that they are the reference's matrices): 48 floats for the reference view -- [0..8] inverse(K_ref),
 * [9..17] K_ref, [18..33] inverse(E_ref) -- then 64 floats per source view -- [0..15] E_src @ inverse(E_ref), [16..24] K_src,
 * [25..33] inverse(K_src), [34..49] E_ref @ inverse(E_src).  Outputs: masks [3][H][W] bytes (photo = confidence > photo_thres,
 * geo = consistent sources >= geo_mask_thres, final = both), xyz [H][W][3] world point of the averaged depth (meaningful where
 * final), optional depth_avg [H][W] float64 and geo_sum [H][W] int32.  Numeric types follow the reference's numpy dtype flow;
 * cv2.remap(INTER_LINEAR) is restated with OpenCV's 1/32-pixel fixed-point coordinates (oracle/fusion_oracle.py). */
int pmn_fuse_view(const float *maps, long long slot_stride, int ref_slot, const int *src_slots_host,
                  const int *src_hw_host, int n_src, const float *mats, int H, int W, float geo_pixel_thres, float geo_depth_thres, int geo_mask_thres,
                  float photo_thres, unsigned char *masks, float *xyz, double *depth_avg, int *geo_sum, void *stream);

/* ABI 19.  The point list of one fused reference view as PLY vertex records, packed on the device (reference eval.py:270-281: the
 * valid pixels' world points and colours, row-major; :283-297: plyfile's vertex element = x, y, z little-endian float32 + red,
 * green, blue uint8 = 15 bytes).  final_mask [H][W] bytes and xyz [H][W][3] are pmn_fuse_view's outputs; image_hwc [H][W][3] is
 * the reference view's image, uint8 as decoded (image_is_float = 0: the bytes are the colours) or float32 in [0,1]
 * (image_is_float = 1: (unsigned char)(f * 255.0f), the reference's (color * 255).astype(uint8)).  The records of the pixels whose
 * mask byte is non-zero are APPENDED to `records` (device bytes, room for capacity_points records) at record index *cursor
 * (DEVICE int64), in row-major pixel order; then *cursor += count and *view_count (DEVICE int32) = count.  If the view does not
 * fit, nothing is written, the cursor stays and *view_count = -1.  scratch: DEVICE int64[ceil(H*W / 1024)].  Three launches on
 * `stream`; consecutive calls on one stream append view after view -- a scan's whole PLY body as one device buffer. */
int pmn_pack_points(const unsigned char *final_mask, const float *xyz, const void *image_hwc, int image_is_float, int H, int W,
                    unsigned char *records, long long capacity_points, long long *cursor, int *view_count, long long *scratch,
                    void *stream);

/* ABI 20.  Refinement's input normalisation (reference models/net.py:104-106): out = (depth - depth_min[b]) / (depth_max[b] -
 * depth_min[b]) over n floats per batch element, IEEE subtraction and correctly rounded division = the bits of the torch expression.
 * With it a whole forward consists of launches of this library only, which is what makes it recordable as a launch plan. */
int pmn_normalize_depth(const float *depth, const float *depth_min, const float *depth_max, int B, int n, float *out,
                        void *stream);

/* ABI 21.  Opt-in range check for the fp16-split entry points (see "accepted magnitudes" above): ORs 1 into *flag (DEVICE int, zeroed by
 * the caller) when some element of x[0..n) is not finite or has |x| >= 65504 -- the magnitude at which the split's `hi` half becomes inf.
 * patchmatchnet_amd/ops.py runs it on the inputs of pmn_conv2d_f16s / pmn_offset_heads_f16s / pmn_stem_f16s / pmn_refine_fused when
 * PMN_CHECK_F16_DOMAIN=1 and raises from ops.f16_domain_check(); intermediates that never leave a fused kernel (the stem's conv0 output,
 * pmn_refine_fused's x16) are covered through the next layer's input. */
int pmn_check_f16_domain(const float *x, long long n, int *flag, void *stream);

/* ---- ABI 20: launch plans -------------------------------------------------------------------------------------------------------
 * A plan is a recorded list of kernel launches that pmn_plan_launch replays on a stream with plain hipLaunchKernel calls from C: the
 * whole forward (reference models/net.py:176-301, the body of the loop at eval.py:56-65) as ONE library call per sample, without a
 * HIP graph (same replay rate; nothing but kernel launches to depend on; the plan's contents can be listed).
 *
 *   pmn_plan_create(&plan)
 *   pmn_plan_begin(plan)          from now on every pmn_* entry point called BY THIS THREAD validates its arguments as usual but
 *   ... pmn_* calls ...           appends its launches (kernel, grid, LDS size, a copy of the by-value arguments, neighbour tables
 *   pmn_plan_end(plan)            included) to the plan instead of enqueuing them; their `stream` argument is ignored
 *   pmn_plan_launch(plan, stream) enqueues the recorded launches, in order, on `stream`; any number of times, from any thread
 *   pmn_plan_destroy(plan)
 *
 * The plan holds the DEVICE addresses that were passed while recording: the caller keeps those buffers alive and in place for the
 * life of the plan and feeds new inputs by writing into them (or into the device address tables of pmn_warp_correlate_views /
 * pmn_stem_f16s_views) before each launch, on the same stream.  One recording per thread at a time; a plan is recorded once.
 * pmn_plan_count = number of recorded launches; pmn_plan_kernel_name(plan, i) = the i-th kernel's symbol name (diagnostics).
 * Like every entry point, none of these synchronises, allocates device memory or copies. */
int pmn_plan_create(void **plan_out);
int pmn_plan_begin(void *plan);
int pmn_plan_end(void *plan);
int pmn_plan_count(const void *plan);
const char *pmn_plan_kernel_name(const void *plan, int index);
int pmn_plan_launch(const void *plan, void *stream);
int pmn_plan_destroy(void *plan);


#ifdef __cplusplus
}
#endif
#endif /* PMN_HIP_H */
