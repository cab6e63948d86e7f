// aggregate.hip -- adaptive spatial cost aggregation + softmax over hypotheses + depth regression.
// Reference: models/patchmatch.py:650-669 (depth_weight), :509-510 (weight normalisation), :569-577 (SimilarityNet
// neighbour gather + weighted sum), :221 (exp(log_softmax)), :226-237 (regression).
//
// A thread owns a pixel and a strided subset of its hypotheses.  The K neighbour tap sets (offset + 4 corner weights) and
// the K feature weights are computed once and kept in registers; for every hypothesis d the thread gathers the neighbour's normalised inverse depth and
// pointwise cost at the same taps (both planes are [D,h,w], x fastest, so a wave's taps of one corner are a nearly
// contiguous run), forms the depth weight, normalises over the K neighbours and accumulates the aggregated score.
// The D scores are parked in the output score buffer (each thread re-reads only its own column) for the softmax and
// the regression; the [B,D,K,h,w] weight tensor of the reference is never materialised.
#include <cstring>

#include "pmn_common.hpp"

struct AggArgs {
    const float* cost;     // [B,D,h,w]
    const float* depth;    // [B,D,h,w]
    const float* xnorm;    // [B,D,h,w]
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
        const float* xp = a.xnorm + ((size_t)b * D + d) * hw;
        const float* cp = a.cost + ((size_t)b * D + d) * hw;
        const float xc = xp[p];
        float wk[KMAX], ck[KMAX];
        float wsum = 0.0f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            wk[k] = 0.0f;
            ck[k] = 0.0f;
            if (k < K) {
                const int o = off[k];
                const float x1 =
                    fmaf(xp[o + w + 1], w11[k], fmaf(xp[o + w], w10[k], fmaf(xp[o + 1], w01[k], xp[o] * w00[k])));
                ck[k] = fmaf(cp[o + w + 1], w11[k], fmaf(cp[o + w], w10[k], fmaf(cp[o + 1], w01[k], cp[o] * w00[k])));
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
        const float inv = inv_max + acc / (float)(D - 1) * (inv_min - inv_max);
        out = 1.0f / inv;
    }
    a.depth_out[(size_t)b * hw + p] = out;
}

template <int KMAX, int DL>
static int launch_agg_dl(const AggArgs& a, hipStream_t s) {
    constexpr int NPX = PMN_BLOCK / DL;
    const dim3 grid((a.h * a.w + NPX - 1) / NPX, a.B), block(PMN_BLOCK);
    hipLaunchKernelGGL((aggregate_regress_kernel<KMAX, DL>), grid, block, 0, s, a);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

template <int KMAX>
static int launch_agg(const AggArgs& a, hipStream_t s) {
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
