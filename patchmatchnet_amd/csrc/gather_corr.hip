// gather_corr.hip -- the fused warp + bilinear gather + group-wise correlation kernel (and its two epilogues).
//
// One kernel template, three modes:
//   MODE_VIEWS      Evaluation.forward with known view weights  (every PatchMatch iteration but the first)
//   MODE_PIXELWISE  Evaluation.forward computing the view weights with PixelwiseNet (stage-3, iteration 1)
//   MODE_NEIGHBOR   FeatureWeightNet.forward (K learned-offset neighbours of the reference feature itself)
// Reference: models/patchmatch.py:192-217 + :570 (Evaluation / SimilarityNet MLP), :695-702 (PixelwiseNet),
// :603-624 (FeatureWeightNet); models/module.py:130-181 (differentiable_warping).
//
// Mapping to CDNA4 (wave64, 256-thread workgroups = 4 waves):
//   * feature maps are channels-last fp32, so one bilinear corner of one texel is C*4 contiguous bytes
//     (256/128/64 B at C = 64/32/16).  A lane owns one float4 channel quad; LPI = C/4 lanes cooperate on one
//     (pixel, hypothesis) item and a wave covers 64/LPI consecutive pixels at the same hypothesis, so a wave-level
//     corner load is one contiguous ~1 KB run of the source map whenever the homography is locally ~1 px/px.
//   * a workgroup owns a tile of NPIX = 256/LPI consecutive pixels x all D hypotheses.  Per view:
//       phase A  every thread projects (pixel, d) items and parks {texel offset, 4 corner weights} in LDS
//                (the projection is done once per item, not once per lane);
//       phase B  lane groups walk their pixel's D items: LDS broadcast of the record, 4 x global_load_dwordx4,
//                bilinear blend, product with the (register-resident) reference quad, in-lane + one DPP step
//                group reduction, view-weighted accumulation into an LDS tile [G][items];
//       (PIXELWISE) the per-view similarity tile is pushed through PixelwiseNet (weights in SGPRs via kernarg),
//                max over D by a 64-bit LDS atomic max (value bits | ~d  ->  first arg-max), then accumulated.
//     Epilogue (phase C): divide by the view-weight sum, SimilarityNet / FeatureWeightNet MLP per item, coalesced
//     store of cost[d][pixel].  The [C,D,h,w] warped volume and the per-view [G,D,h,w] similarity never touch HBM.
//   * no MFMA: ~10 flop per gathered float, no dense contraction.
#include <cstdlib>
#include <cstring>

#include "pmn_common.hpp"

enum { MODE_VIEWS = 0, MODE_PIXELWISE = 1, MODE_NEIGHBOR = 2 };

struct GatherArgs {
    const float* ref;      // [B,h,w,C]
    const float* src;      // [N,B,hs,ws,C]
    const float* proj;     // [B,N,4,4]
    const float* depth;    // [B,D,h,w]
    const float* offsets;  // [B,2K,h,w]   (MODE_NEIGHBOR)
    const float* vw_in;    // [B,N,h>>s,w>>s]
    float* vw_out;         // [B,N,h,w]
    int* vw_argmax;        // [B,N,h,w] or null
    float* sim_out;        // [B,G,D,h,w] or null
    float* out;            // [B,D,h,w]
    int B, N, D, h, w, hs, ws, vw_shift, vchunk, ntiles;
    int table[2 * PMN_MAX_NEIGHBORS];
    PmnMlp mlp_a;  // similarity_net | feature_weight_net
    PmnMlp mlp_b;  // pixel_wise_net
};

__device__ __forceinline__ float pmn_pair_swap(float v) {
    // lane l <-> lane l^1 through DPP quad_perm [1,0,3,2]
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
}

template <int C, int G, int MODE>
__global__ __launch_bounds__(PMN_BLOCK) void gather_corr_kernel(const GatherArgs a) {
    constexpr int LPI = C / 4;           // lanes per (pixel, hypothesis) item
    constexpr int NPIX = PMN_BLOCK / LPI;  // pixels per workgroup tile
    constexpr int CG = C / G;            // channels per correlation group (4 or 8)
    constexpr int LPG = CG / 4;          // lanes per group (1 or 2)
    constexpr int PAD = 32 / G;          // LDS row padding: rows of different groups land on different banks
    static_assert(CG == 4 || CG == 8, "group size must be 4 or 8 channels");

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int tile = pmn_xcd_tile(blockIdx.x, a.ntiles);
    const int D = a.D, N = a.N, h = a.h, w = a.w, hs = a.hs, ws = a.ws;
    const int hw = h * w;
    const int items = NPIX * D;
    const int SS = items + PAD;
    const int p0 = tile * NPIX;
    const int vchunk = (MODE == MODE_VIEWS) ? a.vchunk : 1;

    extern __shared__ float4 smem4[];
    float4* recw = smem4;                                         // [vchunk][items]
    int* reco = reinterpret_cast<int*>(recw + vchunk * items);    // [vchunk][items]
    float* sim_sum = reinterpret_cast<float*>(reco + vchunk * items);  // [G][SS]
    float* sim_v = sim_sum + G * SS;                              // [G][SS]           (PIXELWISE)
    unsigned long long* vwkey = reinterpret_cast<unsigned long long*>(sim_v + (MODE == MODE_PIXELWISE ? G * SS : 0) +
                                                                      ((G * SS) & 1));  // 8-byte aligned
    float* wsum = reinterpret_cast<float*>(vwkey + NPIX);         // [NPIX]            (PIXELWISE)
    int* tab = reinterpret_cast<int*>(wsum + NPIX);               // [2K]              (NEIGHBOR)

    // ---- prologue -------------------------------------------------------------------------------------------
    for (int i = tid; i < G * SS; i += PMN_BLOCK) sim_sum[i] = 0.0f;
    if (MODE == MODE_PIXELWISE && tid < NPIX) wsum[tid] = 1e-5f;
    if (MODE == MODE_NEIGHBOR) {
        // static indices only: a dynamic index into the by-value kernarg struct would spill it to scratch
#pragma unroll
        for (int i = 0; i < 2 * PMN_MAX_NEIGHBORS; ++i)
            if (tid == i) tab[i] = a.table[i];
    }

    // phase-A role: a fixed pixel of the tile, hypotheses dA0, dA0 + 256/NPIX, ...
    const int pixA = tid % NPIX, dA0 = tid / NPIX;
    const int pA = p0 + pixA;
    const bool okA = pA < hw;
    const int yA = okA ? pA / w : 0, xA = okA ? pA - yA * w : 0;

    // phase-B role: lane lc of the lane group that owns pixel `grp` of the tile
    const int grp = tid / LPI, lc = tid % LPI;
    const int pB = p0 + grp;
    const bool okB = pB < hw;
    float4 refq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (okB) refq = reinterpret_cast<const float4*>(a.ref)[((size_t)b * hw + pB) * LPI + lc];
    const int yB = okB ? pB / w : 0, xB = okB ? pB - yB * w : 0;
    const int wv = w >> a.vw_shift, hwv = (h >> a.vw_shift) * wv;
    const int vw_idx = (yB >> a.vw_shift) * wv + (xB >> a.vw_shift);

    if (MODE == MODE_NEIGHBOR) __syncthreads();  // tab visible

    for (int v0 = 0; v0 < N; v0 += vchunk) {
        const int nv = min(vchunk, N - v0);
        // ---- phase A: project items, park tap records in LDS ----------------------------------------------------
        if (MODE == MODE_PIXELWISE && tid < NPIX) vwkey[tid] = 0ull;
        for (int vc = 0; vc < nv; ++vc) {
            const float* P = a.proj + ((size_t)b * N + (v0 + vc)) * 16;
            for (int d = dA0; d < D; d += PMN_BLOCK / NPIX) {
                PmnTaps t;
                t.off = 0;
                t.w00 = t.w01 = t.w10 = t.w11 = 0.0f;
                if (okA) {
                    float ix, iy;
                    if (MODE == MODE_NEIGHBOR) {
                        const float ox = a.offsets[((size_t)b * 2 * D + 2 * d) * hw + pA];
                        const float oy = a.offsets[((size_t)b * 2 * D + 2 * d + 1) * hw + pA];
                        pmn_neighbor_position((float)xA, (float)yA, tab[2 * d], tab[2 * d + 1], ox, oy, h, w, ix, iy);
                    } else {
                        const float dep = a.depth[((size_t)b * D + d) * hw + pA];
                        pmn_warp_position(P, (float)xA, (float)yA, dep, h, w, hs, ws, ix, iy);
                    }
                    t = pmn_make_taps(ix, iy, hs, ws);
                }
                const int i = vc * items + d * NPIX + pixA;
                recw[i] = make_float4(t.w00, t.w01, t.w10, t.w11);
                reco[i] = t.off;
            }
        }
        __syncthreads();

        // ---- phase B: gather + correlate ----------------------------------------------------------------------
        for (int vc = 0; vc < nv; ++vc) {
            const int v = v0 + vc;
            const float4* srcv = reinterpret_cast<const float4*>(MODE == MODE_NEIGHBOR ? a.ref : a.src) +
                                 ((size_t)(MODE == MODE_NEIGHBOR ? b : v * a.B + b) * hs * ws) * LPI + lc;
            float vw = 1.0f;
            if (MODE == MODE_VIEWS) vw = okB ? a.vw_in[((size_t)b * N + v) * hwv + vw_idx] : 0.0f;
            const float4* rw = recw + vc * items + grp;
            const int* ro = reco + vc * items + grp;
#pragma unroll 4
            for (int d = 0; d < D; ++d) {
                const float4 w4 = rw[d * NPIX];
                const int off = ro[d * NPIX];
                const float4* bp = srcv + (size_t)off * LPI;
                const float4 t00 = bp[0];
                const float4 t01 = bp[LPI];
                const float4 t10 = bp[(size_t)ws * LPI];
                const float4 t11 = bp[(size_t)ws * LPI + LPI];
                float4 val;
                val.x = fmaf(t11.x, w4.w, fmaf(t10.x, w4.z, fmaf(t01.x, w4.y, t00.x * w4.x)));
                val.y = fmaf(t11.y, w4.w, fmaf(t10.y, w4.z, fmaf(t01.y, w4.y, t00.y * w4.x)));
                val.z = fmaf(t11.z, w4.w, fmaf(t10.z, w4.z, fmaf(t01.z, w4.y, t00.z * w4.x)));
                val.w = fmaf(t11.w, w4.w, fmaf(t10.w, w4.z, fmaf(t01.w, w4.y, t00.w * w4.x)));
                float s = fmaf(val.w, refq.w, fmaf(val.z, refq.z, fmaf(val.y, refq.y, val.x * refq.x)));
                if (LPG == 2) s += pmn_pair_swap(s);
                s *= (1.0f / CG);
                if (lc % LPG == 0) {
                    const int idx = (lc / LPG) * SS + d * NPIX + grp;
                    if (MODE == MODE_VIEWS) {
                        atomicAdd(&sim_sum[idx], s * vw);  // ds_add_f32, one owner lane per address
                    } else if (MODE == MODE_PIXELWISE) {
                        sim_v[idx] = s;
                    } else {
                        sim_sum[idx] = s;
                    }
                }
            }
        }
        __syncthreads();

        if (MODE == MODE_PIXELWISE) {
            // ---- PixelwiseNet on this view's similarity tile; max over D per pixel ----------------------------------
            const int v = v0;
            for (int d = dA0; d < D; d += PMN_BLOCK / NPIX) {
                const int i = d * NPIX + pixA;
                float x[G];
#pragma unroll
                for (int g = 0; g < G; ++g) x[g] = sim_v[g * SS + i];
                const float r = pmn_sigmoid(pmn_mlp_eval<G>(a.mlp_b, x));
                const unsigned long long key =
                    ((unsigned long long)__float_as_uint(r) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)d);
                atomicMax(&vwkey[pixA], key);
            }
            __syncthreads();
            {
#pragma clang fp contract(off)
                const unsigned long long key = vwkey[pixA];
                const float vwp = __uint_as_float((unsigned)(key >> 32));
                for (int d = dA0; d < D; d += PMN_BLOCK / NPIX) {
                    const int i = d * NPIX + pixA;
#pragma unroll
                    for (int g = 0; g < G; ++g) sim_sum[g * SS + i] = sim_sum[g * SS + i] + sim_v[g * SS + i] * vwp;
                }
                if (tid < NPIX) {
                    wsum[tid] = wsum[tid] + vwp;
                    if (okA) {
                        const size_t o = ((size_t)b * N + v) * hw + pA;
                        a.vw_out[o] = vwp;
                        if (a.vw_argmax) a.vw_argmax[o] = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
                    }
                }
            }
            __syncthreads();
        }
    }

    // ---- phase C: normalise by the view-weight sum, pointwise MLP, store -------------------------------------------
    if (!okA) return;
    float wtot = 1.0f;
    if (MODE == MODE_VIEWS) {
        wtot = 1e-5f;
        const int vwi = (yA >> a.vw_shift) * wv + (xA >> a.vw_shift);
        for (int v = 0; v < N; ++v) wtot += a.vw_in[((size_t)b * N + v) * hwv + vwi];
    } else if (MODE == MODE_PIXELWISE) {
        wtot = wsum[pixA];
    }
    for (int d = dA0; d < D; d += PMN_BLOCK / NPIX) {
        const int i = d * NPIX + pixA;
        float x[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            x[g] = sim_sum[g * SS + i];
            if (MODE != MODE_NEIGHBOR) x[g] = x[g] / wtot;
        }
        if (MODE != MODE_NEIGHBOR && a.sim_out) {
#pragma unroll
            for (int g = 0; g < G; ++g) a.sim_out[(((size_t)b * G + g) * D + d) * hw + pA] = x[g];
        }
        float o = pmn_mlp_eval<G>(a.mlp_a, x);
        if (MODE == MODE_NEIGHBOR) o = pmn_sigmoid(o);
        a.out[((size_t)b * D + d) * hw + pA] = o;
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------

static int env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    return s ? atoi(s) : dflt;
}

template <int C, int G, int MODE>
static int launch_gather(GatherArgs& a, hipStream_t stream) {
    constexpr int LPI = C / 4, NPIX = PMN_BLOCK / LPI, PAD = 32 / G;
    const int hw = a.h * a.w;
    a.ntiles = (hw + NPIX - 1) / NPIX;
    if (MODE != MODE_VIEWS) a.vchunk = 1;
    a.vchunk = a.vchunk < 1 ? 1 : (a.vchunk > a.N ? a.N : a.vchunk);
    const int items = NPIX * a.D, SS = items + PAD;
    size_t lds = (size_t)a.vchunk * items * 20 + (size_t)G * SS * 4 * (MODE == MODE_PIXELWISE ? 2 : 1) + 8 /*align*/ +
                 NPIX * 12 + 2 * PMN_MAX_NEIGHBORS * 4;
    lds = (lds + 15) & ~(size_t)15;
    auto kern = gather_corr_kernel<C, G, MODE>;
    if (lds > 160 * 1024) return PMN_ERR_SHAPE;
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return PMN_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(a.ntiles, a.B), dim3(PMN_BLOCK), lds, stream, a);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

template <int MODE>
static int dispatch_gather(GatherArgs& a, int C, int G, hipStream_t stream) {
    if (C == 64 && G == 8) return launch_gather<64, 8, MODE>(a, stream);
    if (C == 32 && G == 8) return launch_gather<32, 8, MODE>(a, stream);
    if (C == 16 && G == 4) return launch_gather<16, 4, MODE>(a, stream);
    return PMN_ERR_SHAPE;
}

static void load_mlp(PmnMlp& dst, const float* host) { memcpy(&dst, host, sizeof(PmnMlp)); }

extern "C" int pmn_warp_correlate(const float* ref_nhwc, const float* src_nhwc, const float* rel_proj,
                                  const float* depth_sample, const float* view_weights_in, int vw_shift,
                                  const float* similarity_mlp_host, const float* pixelwise_mlp_host, int B, int N, int C,
                                  int G, int D, int h, int w, int hs, int ws, float* cost_out, float* view_weights_out,
                                  int* vw_argmax_out, float* similarity_out, void* stream) {
    if (!ref_nhwc || !src_nhwc || !rel_proj || !depth_sample || !similarity_mlp_host || !cost_out) return PMN_ERR_ARG;
    if (!view_weights_in && (!pixelwise_mlp_host || !view_weights_out)) return PMN_ERR_ARG;
    if (B < 1 || N < 1 || D < 1 || h < 2 || w < 2 || hs < 2 || ws < 2 || vw_shift < 0 || vw_shift > 2) return PMN_ERR_ARG;
    if (D > PMN_MAX_DEPTH) return PMN_ERR_SHAPE;
    if ((h | w) & ((1 << vw_shift) - 1)) return PMN_ERR_ARG;
    GatherArgs a;
    memset(&a, 0, sizeof(a));
    a.ref = ref_nhwc;
    a.src = src_nhwc;
    a.proj = rel_proj;
    a.depth = depth_sample;
    a.vw_in = view_weights_in;
    a.vw_out = view_weights_out;
    a.vw_argmax = vw_argmax_out;
    a.sim_out = similarity_out;
    a.out = cost_out;
    a.B = B; a.N = N; a.D = D; a.h = h; a.w = w; a.hs = hs; a.ws = ws;
    a.vw_shift = vw_shift;
    a.vchunk = env_int("PMN_VCHUNK", 2);
    load_mlp(a.mlp_a, similarity_mlp_host);
    if (view_weights_in) return dispatch_gather<MODE_VIEWS>(a, C, G, (hipStream_t)stream);
    load_mlp(a.mlp_b, pixelwise_mlp_host);
    return dispatch_gather<MODE_PIXELWISE>(a, C, G, (hipStream_t)stream);
}

extern "C" int pmn_feature_weight(const float* ref_nhwc, const float* eval_offsets, const int* eval_table_host,
                                  const float* mlp_host, int B, int C, int G, int K, int h, int w,
                                  float* out_feature_weight, void* stream) {
    if (!ref_nhwc || !eval_offsets || !eval_table_host || !mlp_host || !out_feature_weight) return PMN_ERR_ARG;
    if (B < 1 || h < 2 || w < 2) return PMN_ERR_ARG;
    if (K < 1 || K > PMN_MAX_NEIGHBORS) return PMN_ERR_SHAPE;
    GatherArgs a;
    memset(&a, 0, sizeof(a));
    a.ref = ref_nhwc;
    a.src = ref_nhwc;
    a.offsets = eval_offsets;
    a.out = out_feature_weight;
    a.B = B; a.N = 1; a.D = K; a.h = h; a.w = w; a.hs = h; a.ws = w;
    a.vchunk = 1;
    for (int i = 0; i < 2 * K; ++i) a.table[i] = eval_table_host[i];
    load_mlp(a.mlp_a, mlp_host);
    return dispatch_gather<MODE_NEIGHBOR>(a, C, G, (hipStream_t)stream);
}
