// stem_conv2.hip -- CANDIDATE (not part of libpmn_hip.so; scripts/experiments/README.md): FeatureNet's conv0 + conv1 + conv2 (reference
// models/net.py:17-20, 51-52) in ONE kernel instead of pmn_stem_f16s followed by pmn_conv2d_f16s(8 -> 16, 5x5, stride 2).
//
// Why: the stem's output [N,H,W,8] fp32 (61 MB per 1600x1200 view) is the largest tensor FeatureNet touches; it is written once and
// read once, by conv2, whose own kernel spends as many VALU cycles staging + splitting it as MFMA cycles convolving it.  Here a
// workgroup owns 8 x 16 conv2 outputs (half resolution) and keeps both intermediates in LDS as split fp16 planes:
//   P1  image patch 23 x 40 x 3 -> LDS                                      (rows 2 oy0 - 4 .., columns 2 ox0 - 4 ..: float4-aligned)
//   P2  conv0 + BN + ReLU (fp32 VALU, stem_f16s_kernel's arithmetic) on 21 x 37, zero outside the image -> hi / lo planes M0
//   P3  conv1 as split-operand MFMAs (stem_f16s_kernel's k-steps and accumulation order) on the 19 x 35 patch, 42 M-tiles of 16 linear
//       pixels; + shift, ReLU, zero outside the image, split exactly as conv_f16s_kernel's staging splits what it loads -> planes M1
//   P4  conv2 as conv_f16s_kernel<8,16,5,2,...> computes it from its LDS patch (same k-steps, operand order and epilogue) -> [N,Ho,Wo,16]
// Every value is produced by the same operations in the same order as in the two product kernels: the output must be BIT-IDENTICAL
// (ab.py checks with torch.equal).  HBM traffic per view: 23 + 31 MB instead of 23 + 61 + 61 + 31.
#include "pmn_common.hpp"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
#define SC_LO_SCALE 2048.0f

template <bool VEC4>
__global__ __launch_bounds__(256, 3) void stem_conv2_kernel(const float* __restrict__ img, const float* __restrict__ w0,
                                                           const float* __restrict__ s0, const f16x8* __restrict__ w1A,
                                                           const float* __restrict__ s1, const f16x8* __restrict__ w2B,
                                                           const float* __restrict__ s2, float* __restrict__ out, const int N,
                                                           const int H, const int W, const int Ho, const int Wo) {
    constexpr int TWO = 16, THO = 8, NTHR = 256;
    constexpr int IR = 23, IC = 40;            // image patch rows / floats per row
    constexpr int R0 = 21, C0 = 37;            // conv0 patch (planes M0: 16 B per pixel)
    constexpr int R1 = 19, C1 = 35;            // conv1 patch (planes M1)
    constexpr int M0PLANE = R0 * C0 * 8, M1PLANE = R1 * C1 * 8;  // halves
    // LDS: [M0 hi | M0 lo] 24,864 B, then one region that holds the image patch (11,040 B) in P1-P2 and [M1 hi | M1 lo] (21,280 B) after
    extern __shared__ float4 sc_lds4[];
    _Float16* m0h = reinterpret_cast<_Float16*>(sc_lds4);
    _Float16* m1h = m0h + 2 * M0PLANE;
    float* xin = reinterpret_cast<float*>(m1h);
    typedef const float __attribute__((address_space(4))) cfloat;
    const cfloat* cw0 = (const cfloat*)w0;  // [3][3][3][8]
    const cfloat* cs0 = (const cfloat*)s0;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 15, kb = lane >> 4;
    const int tiles_x = (Wo + TWO - 1) / TWO, tiles_y = (Ho + THO - 1) / THO;
    const int bt = pmn_xcd_tile(blockIdx.x, N * tiles_x * tiles_y);
    const int n = bt / (tiles_x * tiles_y), tr = bt - n * tiles_x * tiles_y;
    const int oy0 = (tr / tiles_x) * THO, ox0 = (tr % tiles_x) * TWO;
    const int yi = 2 * oy0 - 4, xi = 2 * ox0 - 4;  // image patch origin; conv0 patch origin = +1, conv1 patch origin = +2

    // conv1's weights (A operands of its three k-steps, hi | lo): in flight across P1 / P2
    f16x8 wa[3][2];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        wa[ks][0] = w1A[(ks * 2 + 0) * 64 + lane];
        wa[ks][1] = w1A[(ks * 2 + 1) * 64 + lane];
    }
    // ---- P1: image patch, every load of the thread in flight before the first LDS write.  Row R of 69 = (channel, patch row) --------
    if constexpr (VEC4) {
        float4 v[5];
        const int j = tid & 15, gx = xi + 4 * j;  // float4 j of the row (10 used)
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int R = (tid >> 4) + 16 * u;
            const int c = (R >= IR) + (R >= 2 * IR), r = R - IR * c, gy = yi + r;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < 10 && R < 3 * IR && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                v[u] = *reinterpret_cast<const float4*>(img + (((size_t)n * 3 + c) * H + gy) * W + gx);
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int R = (tid >> 4) + 16 * u;
            if (j < 10 && R < 3 * IR) *reinterpret_cast<float4*>(xin + R * IC + 4 * j) = v[u];
        }
    } else {  // any width / base alignment: one float per thread, 4 rows of 64 columns (40 used) per pass
        const int q = tid & 63, gx = xi + q;
#pragma unroll 1
        for (int u0 = 0; u0 < 18; u0 += 6) {
            float v[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int R = (tid >> 6) + 4 * (u0 + u);
                const int c = (R >= IR) + (R >= 2 * IR), r = R - IR * c, gy = yi + r;
                v[u] = 0.0f;
                if (q < IC && R < 3 * IR && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                    v[u] = img[(((size_t)n * 3 + c) * H + gy) * W + gx];
            }
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int R = (tid >> 6) + 4 * (u0 + u);
                if (q < IC && R < 3 * IR) xin[R * IC + q] = v[u];
            }
        }
    }
    __syncthreads();
    // ---- P2: conv0 + BN + ReLU on the 21 x 37 patch (global origin (yi + 1, xi + 1)) -> hi / lo planes M0 --------------------------
    for (int m = tid; m < R0 * C0; m += NTHR) {
        const int r = m / C0, q = m - r * C0;
        const int gy = yi + 1 + r, gx = xi + 1 + q;
        f16x8 hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = {0, 0, 0, 0, 0, 0, 0, 0};
        if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
            float acc[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = 0.0f;
            const float* xp = xin + r * IC + q;  // input (gy - 1 + ky, gx - 1 + kx) = patch (r + ky, q + kx)
#pragma unroll 1
            for (int ky = 0; ky < 3; ++ky) {
                const cfloat* wq = cw0 + __builtin_amdgcn_readfirstlane(ky * 72);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci) {
                        const float v = xp[(ci * IR + ky) * IC + kx];
#pragma unroll
                        for (int c = 0; c < 8; ++c) acc[c] = fmaf(v, wq[(kx * 3 + ci) * 8 + c], acc[c]);
                    }
            }
#pragma unroll
            for (int c = 0; c < 8; c += 2) {
                const f32x2_t x = {fmaxf(acc[c] + cs0[c], 0.0f), fmaxf(acc[c + 1] + cs0[c + 1], 0.0f)};
                const f16x2_t h = __builtin_convertvector(x, f16x2_t);
                const f32x2_t d = (x - __builtin_convertvector(h, f32x2_t)) * SC_LO_SCALE;
                const f16x2_t l = __builtin_convertvector(d, f16x2_t);
                hi[c] = h[0];
                hi[c + 1] = h[1];
                lo[c] = l[0];
                lo[c + 1] = l[1];
            }
        }
        *reinterpret_cast<f16x8*>(m0h + m * 8) = hi;
        *reinterpret_cast<f16x8*>(m0h + M0PLANE + m * 8) = lo;
    }
    __syncthreads();  // M0 complete; the image patch is dead: its region becomes M1
    // ---- P3: conv1 on the 19 x 35 patch (global origin (yi + 2, xi + 2)): M-tile t = pixels 16 t .. 16 t + 15 in linear order ---------
    {
        const f32x4_t sh1 = *reinterpret_cast<const f32x4_t*>(s1 + 4 * (kb & 1));
        constexpr int NT1 = (R1 * C1 + 15) / 16;  // 42
#pragma unroll 1
        for (int g = 0; g < 12; g += 2) {  // tiles wave + 4 j, two at a time (two independent accumulator chains)
            if (wave + 4 * g >= NT1) break;
            f32x4_t accM[2], accL[2];
            const _Float16* pb[2];
            int mm[2], tt[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                accM[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                accL[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                tt[j] = wave + 4 * (g + j);
                const int t = tt[j] < NT1 ? tt[j] : NT1 - 1;
                const int m = min(t * 16 + li, R1 * C1 - 1);
                mm[j] = m;
                const int r = m / C1, c = m - r * C1;
                pb[j] = m0h + (r * C0 + c) * 8;
            }
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                int q = 4 * ks + kb;
                q = q < 8 ? q : 8;  // padding blocks 9..11 (zero weights) read tap 8
                const int dy = q / 3, dx = q - dy * 3;
                const int off = (dy * C0 + dx) * 8;
                f16x8 bh[2], blo[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bh[j] = *reinterpret_cast<const f16x8*>(pb[j] + off);
                    blo[j] = *reinterpret_cast<const f16x8*>(pb[j] + off + M0PLANE);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) accM[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ks][0], bh[j], accM[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j) accL[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ks][0], blo[j], accL[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j) accL[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ks][1], bh[j], accL[j], 0, 0, 0);
            }
            // D rows 4 kb + e = output channels: lanes with kb < 2 hold channels [4 kb, 4 kb + 4) of pixel mm[j]
            if (kb < 2) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (tt[j] < NT1 && tt[j] * 16 + li < R1 * C1) {
                        const int m = mm[j], r = m / C1, c = m - r * C1;
                        const int gy = yi + 2 + r, gx = xi + 2 + c;
                        f16x4 hi = {0, 0, 0, 0}, lo = {0, 0, 0, 0};
                        if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                            // stem_f16s_kernel's epilogue, then conv_f16s_kernel's staging split of the fp32 value it would have loaded
                            f32x4_t v = accM[j] + accL[j] * (1.0f / SC_LO_SCALE) + sh1;
                            v = __builtin_elementwise_max(v, f32x4_t{0.f, 0.f, 0.f, 0.f});
                            const f32x2_t x01 = {v[0], v[1]}, x23 = {v[2], v[3]};
                            const f16x2_t h01 = __builtin_convertvector(x01, f16x2_t), h23 = __builtin_convertvector(x23, f16x2_t);
                            const f16x2_t l01 = __builtin_convertvector((x01 - __builtin_convertvector(h01, f32x2_t)) * SC_LO_SCALE, f16x2_t);
                            const f16x2_t l23 = __builtin_convertvector((x23 - __builtin_convertvector(h23, f32x2_t)) * SC_LO_SCALE, f16x2_t);
                            hi = f16x4{h01[0], h01[1], h23[0], h23[1]};
                            lo = f16x4{l01[0], l01[1], l23[0], l23[1]};
                        }
                        *reinterpret_cast<f16x4*>(m1h + m * 8 + 4 * kb) = hi;
                        *reinterpret_cast<f16x4*>(m1h + M1PLANE + m * 8 + 4 * kb) = lo;
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- P4: conv2 (8 -> 16, 5x5, stride 2) as conv_f16s_kernel<8,16,5,2,8,8,2,4,1,0,true>: wave w owns output rows 2 w, 2 w + 1 ----------
    {
        constexpr int MT = 2, NQ = 25, KSTEPS = 7, S = 2, KS = 5;
        f32x4_t accM[MT], accL[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            accM[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            accL[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        const f16x8* bl = w2B + lane;  // element ((ks * 1 + 0) * 2 + split) * 64 + lane
        f16x8 bq[2][2];
        bq[0][0] = bl[0];
        bq[0][1] = bl[64];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            {
                const int sidx = ks + 1 < KSTEPS - 1 ? ks + 1 : KSTEPS - 1;
                bq[nxt][0] = bl[(size_t)(sidx * 2 + 0) * 64];
                bq[nxt][1] = bl[(size_t)(sidx * 2 + 1) * 64];
            }
            int q = 4 * ks + kb;
            if (4 * ks + 3 >= NQ) q = q < NQ - 1 ? q : NQ - 1;
            const int dy = q / KS, dx = q - dy * KS;
            const _Float16* pa = m1h + ((wave * MT * S + dy) * C1 + li * S + dx) * 8;
            f16x8 ah[MT], al[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                ah[t] = *reinterpret_cast<const f16x8*>(pa + t * S * C1 * 8);
                al[t] = *reinterpret_cast<const f16x8*>(pa + t * S * C1 * 8 + M1PLANE);
            }
#pragma unroll
            for (int t = 0; t < MT; ++t) accM[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bq[cur][0], ah[t], accM[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; ++t) accL[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bq[cur][1], ah[t], accL[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; ++t) accL[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bq[cur][0], al[t], accL[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int ox = ox0 + li, oyw = oy0 + wave * MT;
        if (ox < Wo) {
            float* po = out + (((size_t)n * Ho + oyw) * Wo + ox) * 16 + 4 * kb;
            const size_t rs = (size_t)Wo * 16;
            const f32x4_t sh2 = *reinterpret_cast<const f32x4_t*>(s2 + 4 * kb);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                if (oyw + t < Ho) {
                    f32x4_t v = accM[t] + accL[t] * (1.0f / SC_LO_SCALE) + sh2;
                    v = __builtin_elementwise_max(v, f32x4_t{0.f, 0.f, 0.f, 0.f});
                    *reinterpret_cast<f32x4_t*>(po + t * rs) = v;
                }
            }
        }
    }
}

// img [N,3,H,W] planar; w0 / s0 / w1a / s1 as pmn_stem_f16s; w2b DEVICE fp16 [1][7][1][2][64][8] / s2 [16] = params.pack_conv_f16s of
// conv2 (what pmn_conv2d_f16s takes) -> out [N,Ho,Wo,16] channels-last float32, Ho = (H-1)/2 + 1
extern "C" int stem_conv2(const float* img, const float* w0, const float* s0, const void* w1a, const float* s1, const void* w2b,
                          const float* s2, float* out, int N, int H, int W, void* stream) {
    if (!img || !w0 || !s0 || !w1a || !s1 || !w2b || !s2 || !out || N < 1 || H < 1 || W < 1) return PMN_ERR_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const size_t lds = (size_t)2 * 21 * 37 * 8 * sizeof(_Float16) + (size_t)2 * 19 * 35 * 8 * sizeof(_Float16);  // 24,864 + 21,280 B
    const int blocks = N * ((Wo + 15) / 16) * ((Ho + 7) / 8);
    const bool vec4 = W % 4 == 0 && (reinterpret_cast<uintptr_t>(img) & 15) == 0;
    if (vec4)
        hipLaunchKernelGGL(stem_conv2_kernel<true>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, img, w0, s0,
                           reinterpret_cast<const f16x8*>(w1a), s1, reinterpret_cast<const f16x8*>(w2b), s2, out, N, H, W, Ho, Wo);
    else
        hipLaunchKernelGGL(stem_conv2_kernel<false>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, img, w0, s0,
                           reinterpret_cast<const f16x8*>(w1a), s1, reinterpret_cast<const f16x8*>(w2b), s2, out, N, H, W, Ho, Wo);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}
