// aggregate.hip -- adaptive spatial cost aggregation + softmax over hypotheses + depth regression.
// Reference: models/patchmatch.py:650-669 (depth_weight), :509-510 (weight normalisation), :569-577 (SimilarityNet
// neighbour gather + weighted sum), :221 (exp(log_softmax)), :226-237 (regression).
//
// The pointwise cost and the normalised inverse depth arrive HYPOTHESIS-LAST ([B,h,w,D]): the K neighbour tap sets of a pixel
// do not depend on d, so one bilinear corner is D contiguous floats and a thread fetches four hypotheses per 16-byte load
// (planar [B,D,h,w] maps cost one 4-byte gather per corner per hypothesis: 72 address-limited gathers per (pixel, d), which
// bound the planar version of this kernel at 70-170 us per launch).
//
// Main kernel (D % 4 == 0): a thread owns (pixel, quad of 4 consecutive hypotheses); the D/4 threads of a pixel are adjacent
// lanes, so their corner loads are one contiguous run.  The K tap sets (offset + 4 corner weights) and feature weights are
// formed once per thread and kept in registers; per neighbour the thread forms 4 depth weights and 4 sampled costs, then
// normalises over the K neighbours exactly in the reference's order.  Softmax and regression partials are combined across
// the pixel's threads through LDS; the [B,D,K,h,w] weight tensor of the reference is never materialised.  A scalar kernel
// with the same arithmetic covers D % 4 != 0.
#include <cstring>

#include "pmn_common.hpp"
#define PMN_DIV_DM1(acc, D) ((acc) / (float)((D) - 1))

struct AggArgs {
    const float* cost;     // [B,h,w,D]
    const float* depth;    // [B,D,h,w]
    const float* xnorm;    // [B,h,w,D]
    const float* fweight;  // [B,K,h,w]
    const float* offsets;  // [B,2K,h,w]
    float* score;          // [B,D,h,w]
    float* depth_out;      // [B,h,w]
    int K, is_inverse, B, D, h, w;
    float interval_scale;
    int table[2 * PMN_MAX_NEIGHBORS];
};

// Workgroup = NPX pixels x DL "hypothesis lanes" (NPX*DL = 256, pixel fastest so plane accesses stay coalesced):
// thread (pixel, dl) handles hypotheses dl, dl+DL, ...; max / sum / regression partials are combined through LDS.
// DL = 16 when D >= 32 (stage 3 has only 30k pixels at 1600x1200: one thread per pixel would leave most SIMDs idle).
template <int KMAX, int DL>
__global__ __launch_bounds__(PMN_BLOCK, (KMAX <= 9 ? 3 : 1)) void aggregate_regress_kernel(const AggArgs a) {
#pragma clang fp contract(off)
    constexpr int NPX = PMN_BLOCK / DL;
    __shared__ float red[DL][NPX];
    __shared__ float red2[DL][NPX];
    const int h = a.h, w = a.w, hw = h * w, D = a.D, K = a.K;
    const int px = threadIdx.x % NPX, dl = threadIdx.x / NPX;
    const int p_raw = blockIdx.x * NPX + px;
    const bool ok = p_raw < hw;
    const int p = ok ? p_raw : hw - 1;  // out-of-range threads shadow the last pixel (no stores) so barriers stay uniform
    const int b = blockIdx.y;
    const int y = p / w, x = p - y * w;

    int off[KMAX];
    float w00[KMAX], w01[KMAX], w10[KMAX], w11[KMAX], fw[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        off[k] = 0;
        w00[k] = w01[k] = w10[k] = w11[k] = fw[k] = 0.0f;
        if (k < K) {
            const float ox = a.offsets[((size_t)b * 2 * K + 2 * k) * hw + p];
            const float oy = a.offsets[((size_t)b * 2 * K + 2 * k + 1) * hw + p];
            float ix, iy;
            pmn_neighbor_position((float)x, (float)y, a.table[2 * k], a.table[2 * k + 1], ox, oy, h, w, ix, iy);
            const PmnTaps t = pmn_make_taps(ix, iy, h, w);
            off[k] = t.off;
            w00[k] = t.w00;
            w01[k] = t.w01;
            w10[k] = t.w10;
            w11[k] = t.w11;
            fw[k] = a.fweight[((size_t)b * K + k) * hw + p];
        }
    }

    // pass 1: aggregated score of my hypotheses; parked in the (caller-owned) score buffer, re-read by this thread only
    float smax = -__builtin_inff();
#pragma unroll 1
    for (int d = dl; d < D; d += DL) {
        const float* xp = a.xnorm + (size_t)b * hw * D + d;  // hypothesis-last: pixel stride D
        const float* cp = a.cost + (size_t)b * hw * D + d;
        const float xc = xp[(size_t)p * D];
        float wk[KMAX], ck[KMAX];
        float wsum = 0.0f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            wk[k] = 0.0f;
            ck[k] = 0.0f;
            if (k < K) {
                const size_t o = (size_t)off[k] * D, o1 = o + D, o2 = o + (size_t)w * D, o3 = o2 + D;
                const float x1 = fmaf(xp[o3], w11[k], fmaf(xp[o2], w10[k], fmaf(xp[o1], w01[k], xp[o] * w00[k])));
                ck[k] = fmaf(cp[o3], w11[k], fmaf(cp[o2], w10[k], fmaf(cp[o1], w01[k], cp[o] * w00[k])));
                float t = fabsf(x1 - xc) / a.interval_scale;
                t = fminf(fmaxf(t, 0.0f), 4.0f);
                wk[k] = pmn_sigmoid(4.0f - 2.0f * t) * fw[k];
                wsum = wsum + wk[k];
            }
        }
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) s = s + ck[k] * (wk[k] / wsum);
        if (ok) a.score[((size_t)b * D + d) * hw + p] = s;
        smax = fmaxf(smax, s);
    }
    red[dl][px] = smax;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < DL; ++i) smax = fmaxf(smax, red[i][px]);
    __syncthreads();

    // pass 2: exp(log_softmax): log of the sum of exponentials over all hypotheses of the pixel
    float esum = 0.0f;
#pragma unroll 1
    for (int d = dl; d < D; d += DL)
        if (ok) esum = esum + expf(a.score[((size_t)b * D + d) * hw + p] - smax);
    red[dl][px] = esum;
    __syncthreads();
    esum = 0.0f;
#pragma unroll
    for (int i = 0; i < DL; ++i) esum = esum + red[i][px];
    const float lse = logf(esum);

    // pass 3: probabilities + regression partial sums
    float acc = 0.0f;
#pragma unroll 1
    for (int d = dl; d < D; d += DL) {
        if (ok) {
            const size_t o = ((size_t)b * D + d) * hw + p;
            const float prob = expf((a.score[o] - smax) - lse);
            a.score[o] = prob;
            acc = acc + (a.is_inverse ? (float)d : a.depth[o]) * prob;
        }
    }
    red2[dl][px] = acc;
    __syncthreads();
    if (dl != 0 || !ok) return;
    acc = 0.0f;
#pragma unroll
    for (int i = 0; i < DL; ++i) acc = acc + red2[i][px];
    float out = acc;
    if (a.is_inverse) {
        const float inv_min = 1.0f / a.depth[((size_t)b * D + (D - 1)) * hw + p];
        const float inv_max = 1.0f / a.depth[((size_t)b * D) * hw + p];
        const float inv = inv_max + PMN_DIV_DM1(acc, D) * (inv_min - inv_max);
        out = 1.0f / inv;
    }
    a.depth_out[(size_t)b * hw + p] = out;
}

__device__ __forceinline__ float sigmoid_rcp(float x) {  // 1 / (1 + exp(-x)) for |x| <= 4, relative error < 4e-7
    const float d = 1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f);
    float r = __builtin_amdgcn_rcpf(d);
    return r * fmaf(-d, r, 2.0f);
}

// Broadcast `v` from lane SL of every aligned group of G lanes (G, SL compile-time): one DPP move (two for 8-lane groups, the two
// halves of a 16-lane DPP row), no LDS.  Every lane of the wave is active where this is called.
template <int G, int SL>
__device__ __forceinline__ int agg_group_bcast_i(int v) {
    static_assert(G == 2 || G == 4 || G == 8 || G == 16, "lane groups inside a DPP row");
    if constexpr (G == 2) {
        return __builtin_amdgcn_mov_dpp(v, SL | (SL << 2) | ((2 + SL) << 4) | ((2 + SL) << 6), 0xF, 0xF, true);  // quad_perm
    } else if constexpr (G == 4) {
        return __builtin_amdgcn_mov_dpp(v, SL | (SL << 2) | (SL << 4) | (SL << 6), 0xF, 0xF, true);
    } else if constexpr (G == 16) {
        return __builtin_amdgcn_mov_dpp(v, 0x150 + SL, 0xF, 0xF, true);  // row_newbcast
    } else {
        const int lo = __builtin_amdgcn_update_dpp(v, v, 0x150 + SL, 0xF, 0x3, false);   // banks 0-1 = lanes 0..7 of the row
        return __builtin_amdgcn_update_dpp(lo, v, 0x150 + 8 + SL, 0xF, 0xC, false);      // banks 2-3 = lanes 8..15
    }
}
template <int G, int SL>
__device__ __forceinline__ float agg_group_bcast_f(float v) {
    return __int_as_float(agg_group_bcast_i<G, SL>(__float_as_int(v)));
}

// The K tap sets of one pixel (offset + 4 corner weights + feature weight; ~80 VALU instructions each: position, border clip,
// floor, corner logic) do not depend on the hypothesis, and the DQ = D/4 threads of a pixel are adjacent lanes of one wave: with
// DQT > 0 (= DQ, a power of two <= 16: the cascade's D = 8, 16, 32, 64) thread q forms only the tap sets k = q, q + DQ, ... and the
// group exchanges them by DPP broadcasts (6 moves per tap set) -- 9 tap sets per PIXEL instead of per thread, which was a third of
// this kernel's instructions.  Same values, so the results are bit-identical to DQT = 0 (every thread forms all K; any D % 4 == 0).
template <int KMAX>
__device__ __forceinline__ void agg_form_tap_set(const AggArgs& a, int b, int k, int p, int x, int y, int& off, float& w00, float& w01,
                                                 float& w10, float& w11, float& fw) {
    const int h = a.h, w = a.w, hw = h * w, K = a.K;
    const float ox = a.offsets[((size_t)b * 2 * K + 2 * k) * hw + p];
    const float oy = a.offsets[((size_t)b * 2 * K + 2 * k + 1) * hw + p];
    float ix, iy;
    pmn_neighbor_position((float)x, (float)y, a.table[2 * k], a.table[2 * k + 1], ox, oy, h, w, ix, iy);
    const PmnTaps t = pmn_make_taps(ix, iy, h, w);
    off = t.off;
    w00 = t.w00;
    w01 = t.w01;
    w10 = t.w10;
    w11 = t.w11;
    fw = a.fweight[((size_t)b * K + k) * hw + p];
}

// ---- main kernel: D % 4 == 0, thread = (pixel, hypothesis quad) ------------------------------------------------------------
template <int KMAX, int DQT>
__global__ __launch_bounds__(PMN_BLOCK, (KMAX <= 9 ? 3 : 1)) void aggregate_regress_q4_kernel(const AggArgs a) {
#pragma clang fp contract(off)
    __shared__ float red[PMN_BLOCK];
    const int h = a.h, w = a.w, hw = h * w, D = a.D, K = a.K, DQ = DQT > 0 ? DQT : D >> 2;
    const int npx = blockDim.x / DQ;  // host: blockDim.x = npx * DQ
    const int px = threadIdx.x / DQ, q = threadIdx.x - px * DQ;
    const int p_raw = blockIdx.x * npx + px;
    const bool ok = p_raw < hw;
    const int p = ok ? p_raw : hw - 1;  // out-of-range threads shadow the last pixel (no stores) so barriers stay uniform
    const int b = blockIdx.y;
    const int y = p / w, x = p - y * w;

    int off[KMAX];
    float w00[KMAX], w01[KMAX], w10[KMAX], w11[KMAX], fw[KMAX];
    if constexpr (DQT > 1) {
        constexpr int KPT = (KMAX + DQT - 1) / DQT;  // tap sets formed by this thread: k = q + i DQT
        int moff[KPT];
        float m00[KPT], m01[KPT], m10[KPT], m11[KPT], mfw[KPT];
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int k = q + i * DQT;
            moff[i] = 0;
            m00[i] = m01[i] = m10[i] = m11[i] = mfw[i] = 0.0f;
            if (k < K) agg_form_tap_set<KMAX>(a, b, k, p, x, y, moff[i], m00[i], m01[i], m10[i], m11[i], mfw[i]);
        }
        // static (slot, source lane) per k: slot = k / DQT, lane = k % DQT
#define AGG_TAKE(k_)                                                                                         \
    if constexpr ((k_) < KMAX) {                                                                               \
        constexpr int sl_ = (k_) % DQT, it_ = (k_) / DQT;                                                      \
        off[k_] = agg_group_bcast_i<DQT, sl_>(moff[it_]);                                                      \
        w00[k_] = agg_group_bcast_f<DQT, sl_>(m00[it_]);                                                       \
        w01[k_] = agg_group_bcast_f<DQT, sl_>(m01[it_]);                                                       \
        w10[k_] = agg_group_bcast_f<DQT, sl_>(m10[it_]);                                                       \
        w11[k_] = agg_group_bcast_f<DQT, sl_>(m11[it_]);                                                       \
        fw[k_] = agg_group_bcast_f<DQT, sl_>(mfw[it_]);                                                        \
    }
        AGG_TAKE(0) AGG_TAKE(1) AGG_TAKE(2) AGG_TAKE(3) AGG_TAKE(4) AGG_TAKE(5) AGG_TAKE(6) AGG_TAKE(7) AGG_TAKE(8)
        static_assert(KMAX <= 9, "AGG_TAKE list");
#undef AGG_TAKE
    } else {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            off[k] = 0;
            w00[k] = w01[k] = w10[k] = w11[k] = fw[k] = 0.0f;
            if (k < K) agg_form_tap_set<KMAX>(a, b, k, p, x, y, off[k], w00[k], w01[k], w10[k], w11[k], fw[k]);
        }
    }

    // Per (pixel, hypothesis, neighbour) the reference does three divisions and an exp (models/patchmatch.py:663-668, :509) --
    // 50 of the ~60 VALU instructions of this kernel's inner step, which is VALU-bound (68 % busy).  Here: multiply by
    // 1/interval_scale, sigmoid through v_exp_f32 + v_rcp_f32 with one Newton step (relative error < 4e-7 on [-4,4]), one IEEE
    // reciprocal of the weight sum per hypothesis instead of K divisions.  Measured 418 -> 336 us over the five launches of a
    // depth map; probabilities / depth stay inside the parity tolerances (2e-4 abs / 2e-5 rel, tests/test_hip_parity.py).
    // The scalar fallback kernel keeps the reference's operation sequence.
    const float inv_interval = 1.0f / a.interval_scale;
    const float* xn = a.xnorm + (size_t)b * hw * D + 4 * q;
    const float* cs = a.cost + (size_t)b * hw * D + 4 * q;
    const float4 xc4 = *reinterpret_cast<const float4*>(xn + (size_t)p * D);
    const float xc[4] = {xc4.x, xc4.y, xc4.z, xc4.w};
    float wk[KMAX][4], ck[KMAX][4], wsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wk[k][j] = ck[k][j] = 0.0f;
        if (k < K) {
            const size_t o = (size_t)off[k] * D, o2 = o + (size_t)w * D;
            const float4 xa = *reinterpret_cast<const float4*>(xn + o), xb = *reinterpret_cast<const float4*>(xn + o + D);
            const float4 xcn = *reinterpret_cast<const float4*>(xn + o2), xd = *reinterpret_cast<const float4*>(xn + o2 + D);
            const float4 ca = *reinterpret_cast<const float4*>(cs + o), cb = *reinterpret_cast<const float4*>(cs + o + D);
            const float4 cc = *reinterpret_cast<const float4*>(cs + o2), cd = *reinterpret_cast<const float4*>(cs + o2 + D);
            const float x00[4] = {xa.x, xa.y, xa.z, xa.w}, x01[4] = {xb.x, xb.y, xb.z, xb.w};
            const float x10[4] = {xcn.x, xcn.y, xcn.z, xcn.w}, x11[4] = {xd.x, xd.y, xd.z, xd.w};
            const float c00[4] = {ca.x, ca.y, ca.z, ca.w}, c01[4] = {cb.x, cb.y, cb.z, cb.w};
            const float c10[4] = {cc.x, cc.y, cc.z, cc.w}, c11[4] = {cd.x, cd.y, cd.z, cd.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x1 = fmaf(x11[j], w11[k], fmaf(x10[j], w10[k], fmaf(x01[j], w01[k], x00[j] * w00[k])));
                ck[k][j] = fmaf(c11[j], w11[k], fmaf(c10[j], w10[k], fmaf(c01[j], w01[k], c00[j] * w00[k])));
                float t = fabsf(x1 - xc[j]) * inv_interval;
                t = fminf(fmaxf(t, 0.0f), 4.0f);
                wk[k][j] = sigmoid_rcp(4.0f - 2.0f * t) * fw[k];
                wsum[j] = wsum[j] + wk[k][j];
            }
        }
    }
    float s[4] = {0.f, 0.f, 0.f, 0.f}, rsum[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rsum[j] = 1.0f / wsum[j];
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < K) {
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] = s[j] + ck[k][j] * (wk[k][j] * rsum[j]);
        }

    // exp(log_softmax) over the D hypotheses of the pixel: max, log-sum-exp, probabilities; then the regression
    float smax = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
    red[threadIdx.x] = smax;
    __syncthreads();
    for (int i = 0; i < DQ; ++i) smax = fmaxf(smax, red[px * DQ + i]);
    __syncthreads();
    float esum = ((expf(s[0] - smax) + expf(s[1] - smax)) + expf(s[2] - smax)) + expf(s[3] - smax);
    red[threadIdx.x] = esum;
    __syncthreads();
    esum = 0.0f;
    for (int i = 0; i < DQ; ++i) esum = esum + red[px * DQ + i];
    __syncthreads();
    const float lse = logf(esum);
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int d = 4 * q + j;
        const size_t o = ((size_t)b * D + d) * hw + p;
        const float prob = expf((s[j] - smax) - lse);
        if (ok) a.score[o] = prob;
        acc = acc + (a.is_inverse ? (float)d : a.depth[o]) * prob;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (q != 0 || !ok) return;
    acc = 0.0f;
    for (int i = 0; i < DQ; ++i) acc = acc + red[px * DQ + i];
    float out = acc;
    if (a.is_inverse) {
        const float inv_min = 1.0f / a.depth[((size_t)b * D + (D - 1)) * hw + p];
        const float inv_max = 1.0f / a.depth[((size_t)b * D) * hw + p];
        const float inv = inv_max + PMN_DIV_DM1(acc, D) * (inv_min - inv_max);
        out = 1.0f / inv;
    }
    a.depth_out[(size_t)b * hw + p] = out;
}

template <int KMAX>
static int launch_agg_q4(const AggArgs& a, hipStream_t s) {
    const int DQ = a.D / 4, npx = PMN_BLOCK / DQ;
    const dim3 grid((a.h * a.w + npx - 1) / npx, a.B), block(npx * DQ);
    // the cascade's hypothesis counts (8, 16, 32, 64) share the tap sets inside the pixel's lane group; other D % 4 == 0: every thread
    // forms its own
    if (DQ == 2) PMN_LAUNCH((aggregate_regress_q4_kernel<KMAX, 2>), grid, block, 0, s, a);
    else if (DQ == 4) PMN_LAUNCH((aggregate_regress_q4_kernel<KMAX, 4>), grid, block, 0, s, a);
    else if (DQ == 8) PMN_LAUNCH((aggregate_regress_q4_kernel<KMAX, 8>), grid, block, 0, s, a);
    else if (DQ == 16) PMN_LAUNCH((aggregate_regress_q4_kernel<KMAX, 16>), grid, block, 0, s, a);
    else PMN_LAUNCH((aggregate_regress_q4_kernel<KMAX, 0>), grid, block, 0, s, a);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

template <int KMAX, int DL>
static int launch_agg_dl(const AggArgs& a, hipStream_t s) {
    constexpr int NPX = PMN_BLOCK / DL;
    const dim3 grid((a.h * a.w + NPX - 1) / NPX, a.B), block(PMN_BLOCK);
    PMN_LAUNCH((aggregate_regress_kernel<KMAX, DL>), grid, block, 0, s, a);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

template <int KMAX>
static int launch_agg(const AggArgs& a, hipStream_t s) {
    // 17 neighbours x (tap set + 4 weights + 4 costs) does not fit the register file: the scalar kernel takes those
    if constexpr (KMAX <= 9) {
        if (a.D % 4 == 0) return launch_agg_q4<KMAX>(a, s);
    }
    if (a.D >= 32) return launch_agg_dl<KMAX, 16>(a, s);
    if (a.D >= 16) return launch_agg_dl<KMAX, 4>(a, s);
    return launch_agg_dl<KMAX, 1>(a, s);  // stage 1: 480k pixels already fill the chip (measured: DL=4 is slower)
}

extern "C" int pmn_aggregate_regress(const float* cost, const float* depth_sample, const float* xnorm,
                                     const float* feature_weight, const float* eval_offsets, const int* eval_table_host,
                                     int K, float interval_scale, int is_inverse, int B, int D, int h, int w,
                                     float* score_out, float* depth_out, void* stream) {
    if (!cost || !depth_sample || !xnorm || !feature_weight || !eval_offsets || !eval_table_host || !score_out ||
        !depth_out)
        return PMN_ERR_ARG;
    if (B < 1 || D < 1 || h < 2 || w < 2) return PMN_ERR_ARG;
    if (K < 1 || K > PMN_MAX_NEIGHBORS || D > PMN_MAX_DEPTH) return PMN_ERR_SHAPE;
    if (is_inverse && D < 2) return PMN_ERR_ARG;
    AggArgs a;
    memset(&a, 0, sizeof(a));
    a.cost = cost;
    a.depth = depth_sample;
    a.xnorm = xnorm;
    a.fweight = feature_weight;
    a.offsets = eval_offsets;
    a.score = score_out;
    a.depth_out = depth_out;
    a.K = K;
    a.is_inverse = is_inverse;
    a.B = B; a.D = D; a.h = h; a.w = w;
    a.interval_scale = interval_scale;
    for (int i = 0; i < 2 * K; ++i) a.table[i] = eval_table_host[i];
    if (K <= 9) return launch_agg<9>(a, (hipStream_t)stream);
    return launch_agg<17>(a, (hipStream_t)stream);
}
