// hypotheses.hip -- depth-hypothesis generation: DepthInitialization + Propagation (gather, concat, per-pixel sort).
// Reference: models/patchmatch.py:53-94 (initialisation), :115-124 (propagation), :396-426 (get_grid).
//
// One thread per pixel; the D <= 64 hypotheses of a pixel live in registers and are sorted with a fully unrolled
// bitonic network (static indices only, so nothing spills to scratch).  The K propagated values are bilinear/border
// samples of the CENTRE hypothesis (index D0/2) of the neighbouring pixels; that hypothesis is a closed-form function
// of the neighbour's noise / previous depth, so it is recomputed at the 4 taps instead of needing a grid-wide pass.
// All planes are [.., h, w] with x fastest: consecutive lanes read/write consecutive addresses.
#include <cstring>

#include "pmn_common.hpp"
#define PMN_DIV48(u) ((u) / 48.0f)
#define PMN_DIV48R(u) pmn_div_by((u), 48.0f, r48)

struct HypArgs {
    const float* noise;      // [B,48,h,w] or null
    const float* depth;      // [B,1,h>>s,w>>s] or null
    const float* depth_min;  // [B]
    const float* depth_max;  // [B]
    const float* offsets;    // [B,2K,h,w] or null
    float* depth_sample;     // [B,D,h,w]
    float* xnorm;            // [B,h,w,D]
    int depth_shift, num_sample, K, B, h, w;
    float interval_scale;
    int table[2 * PMN_MAX_NEIGHBORS];
};

// centre hypothesis of pixel q (the value Propagation gathers), reference patchmatch.py:118
__device__ __forceinline__ float centre_hypothesis(const HypArgs& a, int b, int qy, int qx, float inv_min, float inv_max,
                                                   float interval, int kc) {
#pragma clang fp contract(off)
    const int h = a.h, w = a.w;
    if (a.noise) {
        const float u = a.noise[((size_t)b * 48 + 24) * h * w + (size_t)qy * w + qx] + 24.0f;
        const float inv = inv_max + PMN_DIV48(u) * (inv_min - inv_max);
        return 1.0f / inv;
    }
    const int ws = w >> a.depth_shift;
    const float dprev = a.depth[(size_t)b * (h >> a.depth_shift) * ws + (size_t)(qy >> a.depth_shift) * ws +
                                (qx >> a.depth_shift)];
    if (a.num_sample == 1) return dprev;
    float inv = 1.0f / dprev + interval * (float)kc;
    inv = fminf(fmaxf(inv, inv_max), inv_min);
    return 1.0f / inv;
}

template <int NP2>
__global__ __launch_bounds__(PMN_BLOCK) void init_hypotheses_kernel(const HypArgs a) {
#pragma clang fp contract(off)
    const int h = a.h, w = a.w, hw = h * w;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= hw) return;
    const int y = p / w, x = p - y * w;
    const float inv_min = 1.0f / a.depth_min[b];
    const float inv_max = 1.0f / a.depth_max[b];
    const int D0 = a.noise ? 48 : a.num_sample;
    const int K = a.K;
    const int D = D0 + K;
    const float interval = (inv_min - inv_max) * a.interval_scale;
    // arange(-n//2, n//2): first entry -ceil(n/2); the centre entry D0//2 maps to kc
    const int k0 = -((a.num_sample + 1) / 2);
    const int kc = k0 + a.num_sample / 2;

    float v[NP2];
#pragma unroll
    for (int j = 0; j < NP2; ++j) v[j] = __builtin_inff();

    if (a.noise) {
#pragma unroll
        for (int j = 0; j < 48; ++j) {
            if (j < NP2) {
                const float u = a.noise[((size_t)b * 48 + j) * hw + p] + (float)j;
                const float inv = inv_max + PMN_DIV48(u) * (inv_min - inv_max);
                v[j] = 1.0f / inv;
            }
        }
    } else {
        const int ws = w >> a.depth_shift;
        const float dprev =
            a.depth[(size_t)b * (h >> a.depth_shift) * ws + (size_t)(y >> a.depth_shift) * ws + (x >> a.depth_shift)];
        if (a.num_sample == 1) {
            v[0] = dprev;
        } else {
            const float inv_prev = 1.0f / dprev;
#pragma unroll
            for (int j = 0; j < NP2; ++j) {
                if (j < a.num_sample) {
                    float inv = inv_prev + interval * (float)(k0 + j);
                    inv = fminf(fmaxf(inv, inv_max), inv_min);
                    v[j] = 1.0f / inv;
                }
            }
        }
    }

    if (K > 0) {
        // propagated hypotheses -> slots D0 .. D0+K-1 (static slot index via the unrolled select below)
        for (int k = 0; k < K; ++k) {
            const float ox = a.offsets[((size_t)b * 2 * K + 2 * k) * hw + p];
            const float oy = a.offsets[((size_t)b * 2 * K + 2 * k + 1) * hw + p];
            float ix, iy;
            pmn_neighbor_position((float)x, (float)y, a.table[2 * k], a.table[2 * k + 1], ox, oy, h, w, ix, iy);
            const PmnTaps t = pmn_make_taps(ix, iy, h, w);
            const int qy = t.off / w, qx = t.off - qy * w;
            const float c00 = centre_hypothesis(a, b, qy, qx, inv_min, inv_max, interval, kc);
            const float c01 = centre_hypothesis(a, b, qy, qx + 1, inv_min, inv_max, interval, kc);
            const float c10 = centre_hypothesis(a, b, qy + 1, qx, inv_min, inv_max, interval, kc);
            const float c11 = centre_hypothesis(a, b, qy + 1, qx + 1, inv_min, inv_max, interval, kc);
            const float val = fmaf(c11, t.w11, fmaf(c10, t.w10, fmaf(c01, t.w01, c00 * t.w00)));
            const int slot = D0 + k;
#pragma unroll
            for (int j = 0; j < NP2; ++j)
                if (j == slot) v[j] = val;
        }
        // ascending bitonic sort of NP2 values (+inf padding sorts to the end), reference patchmatch.py:124
#pragma unroll
        for (int kk = 2; kk <= NP2; kk <<= 1) {
#pragma unroll
            for (int jj = kk >> 1; jj > 0; jj >>= 1) {
#pragma unroll
                for (int i = 0; i < NP2; ++i) {
                    const int l = i ^ jj;
                    if (l > i) {
                        const float lo = fminf(v[i], v[l]), hi = fmaxf(v[i], v[l]);
                        if ((i & kk) == 0) {
                            v[i] = lo;
                            v[l] = hi;
                        } else {
                            v[i] = hi;
                            v[l] = lo;
                        }
                    }
                }
            }
        }
    }

    const float range = inv_min - inv_max;
#pragma unroll
    for (int j = 0; j < NP2; ++j) {
        if (j < D) {
            const size_t o = ((size_t)b * D + j) * hw + p;
            a.depth_sample[o] = v[j];
            a.xnorm[((size_t)b * hw + p) * D + j] = (1.0f / v[j] - inv_max) / range;  // hypothesis-last [B,h,w,D]
        }
    }
}

// The same kernel with the hypothesis counts as template constants (D0T initial / local samples, KT propagated neighbours) for
// the combinations the cascade uses.  The generic kernel above walks the neighbours one by one -- offsets -> position -> taps ->
// 4 dependent loads -> blend, a serial chain of two memory round trips per neighbour in a kernel that runs ONE wave per SIMD --
// and files each value into its slot through a chain of NP2 selects (the slot index D0 + k is a run-time value).  Here the
// neighbours go in batches of 8 with all their loads in flight together, and every slot index is static.  Same arithmetic per
// value, so the results are bit-identical (tests/test_hip_parity.py compares both against the oracle).
template <int NP2, int D0T, int KT>
__global__ __launch_bounds__(PMN_BLOCK) void init_hypotheses_fixed_kernel(const HypArgs a) {
#pragma clang fp contract(off)
    static_assert(D0T + KT <= NP2, "hypotheses fit the sorting network");
    const int h = a.h, w = a.w, hw = h * w;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= hw) return;
    const int y = p / w, x = p - y * w;
    const float inv_min = 1.0f / a.depth_min[b];
    const float inv_max = 1.0f / a.depth_max[b];
    constexpr int D = D0T + KT;
    const float interval = (inv_min - inv_max) * a.interval_scale;
    const int k0 = -((a.num_sample + 1) / 2);
    const int kc = k0 + a.num_sample / 2;
    const bool from_noise = a.noise != nullptr;
    // the divisions below are IEEE divisions in hipcc's own fma sequence (pmn_div / pmn_div_by, csrc/pmn_common.hpp: same bits as `/`,
    // 8 instructions instead of 12, 5 where the divisor is a launch constant); init_hypotheses_kernel keeps the plain operator and
    // tests/test_hip_parity.py holds both against the oracle
    const float r48 = pmn_uniform(pmn_rcp_refined(48.0f));

    float v[NP2];
#pragma unroll
    for (int j = 0; j < NP2; ++j) v[j] = __builtin_inff();

    const int ws = w >> a.depth_shift;
    if (from_noise) {
        float u[D0T];
#pragma unroll
        for (int j = 0; j < D0T; ++j) u[j] = a.noise[((size_t)b * 48 + j) * hw + p];
#pragma unroll
        for (int j = 0; j < D0T; ++j) {
            const float inv = inv_max + PMN_DIV48R(u[j] + (float)j) * (inv_min - inv_max);
            v[j] = pmn_div(1.0f, inv);
        }
    } else {
        const float dprev =
            a.depth[(size_t)b * (h >> a.depth_shift) * ws + (size_t)(y >> a.depth_shift) * ws + (x >> a.depth_shift)];
        if (D0T == 1) {
            v[0] = dprev;
        } else {
            const float inv_prev = pmn_div(1.0f, dprev);
#pragma unroll
            for (int j = 0; j < D0T; ++j) {
                float inv = inv_prev + interval * (float)(k0 + j);
                inv = fminf(fmaxf(inv, inv_max), inv_min);
                v[j] = pmn_div(1.0f, inv);
            }
        }
    }

    if constexpr (KT > 0) {
        constexpr int KB = KT < 8 ? KT : 8;  // neighbours per batch
        static_assert(KT % KB == 0, "whole batches");
        const float* cbase = from_noise ? a.noise + ((size_t)b * 48 + 24) * hw
                                        : a.depth + (size_t)b * (h >> a.depth_shift) * ws;
        const int sh = from_noise ? 0 : a.depth_shift, cw = from_noise ? w : ws;
#pragma unroll
        for (int k0b = 0; k0b < KT; k0b += KB) {
            float ox[KB], oy[KB];
#pragma unroll
            for (int i = 0; i < KB; ++i) {
                ox[i] = a.offsets[((size_t)b * 2 * KT + 2 * (k0b + i)) * hw + p];
                oy[i] = a.offsets[((size_t)b * 2 * KT + 2 * (k0b + i) + 1) * hw + p];
            }
            PmnTaps t[KB];
            float c[KB][4];
#pragma unroll
            for (int i = 0; i < KB; ++i) {
                float ix, iy;
                pmn_neighbor_position((float)x, (float)y, a.table[2 * (k0b + i)], a.table[2 * (k0b + i) + 1], ox[i], oy[i], h, w, ix,
                                      iy);
                t[i] = pmn_make_taps(ix, iy, h, w);
                const int qy = t[i].off / w, qx = t[i].off - qy * w;
                // raw planes the centre hypothesis is a function of: the noise channel 24 or the previous depth (centre_hypothesis)
                c[i][0] = cbase[(size_t)(qy >> sh) * cw + (qx >> sh)];
                c[i][1] = cbase[(size_t)(qy >> sh) * cw + ((qx + 1) >> sh)];
                c[i][2] = cbase[(size_t)((qy + 1) >> sh) * cw + (qx >> sh)];
                c[i][3] = cbase[(size_t)((qy + 1) >> sh) * cw + ((qx + 1) >> sh)];
            }
#pragma unroll
            for (int i = 0; i < KB; ++i) {
                float cc[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (from_noise) {
                        const float inv = inv_max + PMN_DIV48R(c[i][q] + 24.0f) * (inv_min - inv_max);
                        cc[q] = pmn_div(1.0f, inv);
                    } else if (a.num_sample == 1) {
                        cc[q] = c[i][q];
                    } else {
                        float inv = pmn_div(1.0f, c[i][q]) + interval * (float)kc;
                        inv = fminf(fmaxf(inv, inv_max), inv_min);
                        cc[q] = pmn_div(1.0f, inv);
                    }
                }
                v[D0T + k0b + i] = fmaf(cc[3], t[i].w11, fmaf(cc[2], t[i].w10, fmaf(cc[1], t[i].w01, cc[0] * t[i].w00)));
            }
        }
#pragma unroll
        for (int kk = 2; kk <= NP2; kk <<= 1) {
#pragma unroll
            for (int jj = kk >> 1; jj > 0; jj >>= 1) {
#pragma unroll
                for (int i = 0; i < NP2; ++i) {
                    const int l = i ^ jj;
                    if (l > i) {
                        const float lo = fminf(v[i], v[l]), hi = fmaxf(v[i], v[l]);
                        if ((i & kk) == 0) {
                            v[i] = lo;
                            v[l] = hi;
                        } else {
                            v[i] = hi;
                            v[l] = lo;
                        }
                    }
                }
            }
        }
    }

    const float range = inv_min - inv_max, rrange = pmn_rcp_refined(range);
#pragma unroll
    for (int j = 0; j < D; ++j) {
        const size_t o = ((size_t)b * D + j) * hw + p;
        a.depth_sample[o] = v[j];
        a.xnorm[((size_t)b * hw + p) * D + j] = pmn_div_by(pmn_div(1.0f, v[j]) - inv_max, range, rrange);  // hypothesis-last [B,h,w,D]
    }
}

extern "C" int pmn_init_hypotheses(const float* noise, const float* depth, int depth_shift, const float* depth_min,
                                   const float* depth_max, int num_sample, float interval_scale,
                                   const float* propa_offsets, const int* propa_table_host, int K, int B, int h, int w,
                                   float* depth_sample, float* xnorm, void* stream) {
    if ((!noise && !depth) || !depth_min || !depth_max || !depth_sample || !xnorm) return PMN_ERR_ARG;
    if (B < 1 || h < 2 || w < 2 || K < 0 || depth_shift < 0 || depth_shift > 1) return PMN_ERR_ARG;
    if (K > 0 && (!propa_offsets || !propa_table_host)) return PMN_ERR_ARG;
    if (K > 16) return PMN_ERR_SHAPE;
    if (!noise && num_sample < 1) return PMN_ERR_ARG;
    if (depth_shift && ((h | w) & 1)) return PMN_ERR_ARG;
    const int D0 = noise ? 48 : num_sample;
    const int D = D0 + K;
    if (D > PMN_MAX_DEPTH) return PMN_ERR_SHAPE;
    HypArgs a;
    memset(&a, 0, sizeof(a));
    a.noise = noise;
    a.depth = noise ? nullptr : depth;
    a.depth_min = depth_min;
    a.depth_max = depth_max;
    a.offsets = propa_offsets;
    a.depth_sample = depth_sample;
    a.xnorm = xnorm;
    a.depth_shift = depth_shift;
    a.num_sample = noise ? 48 : num_sample;
    a.K = K;
    a.B = B; a.h = h; a.w = w;
    a.interval_scale = interval_scale;
    for (int i = 0; i < 2 * K; ++i) a.table[i] = propa_table_host[i];
    // one wave per workgroup while the map has fewer pixels than the chip has wave slots worth filling (stage 3: 30 000 pixels =
    // 470 waves; as 118 workgroups of 4 waves they would sit on 118 of the 256 CUs)
    const int bs = (h * w <= 64 * 1024) ? 64 : PMN_BLOCK;
    const dim3 grid((h * w + bs - 1) / bs, B), block(bs);
    hipStream_t s = (hipStream_t)stream;
    // the cascade's own combinations (patchmatch_num_sample / propagate_neighbors of the released models) take the fixed kernel
    if (noise && K == 16) PMN_LAUNCH((init_hypotheses_fixed_kernel<64, 48, 16>), grid, block, 0, s, a);
    else if (!noise && num_sample == 16 && K == 16) PMN_LAUNCH((init_hypotheses_fixed_kernel<32, 16, 16>), grid, block, 0, s, a);
    else if (!noise && num_sample == 8 && K == 8) PMN_LAUNCH((init_hypotheses_fixed_kernel<16, 8, 8>), grid, block, 0, s, a);
    else if (!noise && num_sample == 8 && K == 0) PMN_LAUNCH((init_hypotheses_fixed_kernel<8, 8, 0>), grid, block, 0, s, a);
    else if (D <= 8) PMN_LAUNCH(init_hypotheses_kernel<8>, grid, block, 0, s, a);
    else if (D <= 16) PMN_LAUNCH(init_hypotheses_kernel<16>, grid, block, 0, s, a);
    else if (D <= 32) PMN_LAUNCH(init_hypotheses_kernel<32>, grid, block, 0, s, a);
    else PMN_LAUNCH(init_hypotheses_kernel<64>, grid, block, 0, s, a);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}
