// conv.hip -- direct fp32 convolutions for the small-channel CNNs around the hot path: FeatureNet (reference
// models/net.py:9-70), the offset heads propa_conv / eval_conv (models/patchmatch.py:288-311) and Refinement pieces.
//
// Why not MIOpen: at 3..64 channels MIOpen's fp32 Winograd kernels plus the separate BatchNorm, ReLU, add and
// bilinear-upsample passes take 7.6 ms per 1600x1200 6-view sample (profiles/), 20 TFLOP/s effective.  Here one kernel
// does conv + folded BatchNorm shift + ReLU (+ the FPN "upsample x2 and add" of net.py:60-66) and keeps every map
// channels-last, which is also the layout the PatchMatch kernels want (no transposition pass).
//
// Mapping (wave64, 256 threads): a thread owns TP consecutive output pixels x COUT_T (8 or 16) output channels in
// registers.  Loops over (ky, kx, 4-channel input chunk) are ROLLED; inside, the thread loads one float4 of input per
// pixel (NHWC: 16 contiguous bytes) and issues 4*COUT_T*TP FMAs whose weight operands are wave-uniform, i.e. scalar
// loads / SGPR operands ([K][K][CIN][COUTP] weight layout: the COUT_T weights of one input channel are contiguous).
// fp32 throughout (fp16/bf16 features are not parity-safe); BatchNorm is folded into the weights in fp64.
#include <cstdlib>
#include <cstring>

#include "pmn_common.hpp"

struct ConvArgs {
    int N, H, W, Ho, Wo, COUTP, cout, pad, dil, relu, up_h, up_w;
};

// bilinear x2 up-sampling (align_corners=False) source taps of ATen's upsample_bilinear2d
__device__ __forceinline__ void up2_taps(int o, int size_in, int& i0, int& i1, float& l1) {
    float src = ((float)o + 0.5f) * 0.5f - 0.5f;
    src = src < 0.0f ? 0.0f : src;
    i0 = (int)src;
    i1 = i0 + (i0 < size_in - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

template <int CIN, int COUT_T, int K, int S, int TP, bool IN_NCHW, bool OUT_NCHW>
__global__ __launch_bounds__(PMN_BLOCK) void conv_kernel(const float* __restrict__ in, const float* __restrict__ wgt,
                                                         const float* __restrict__ shift, const float* __restrict__ up,
                                                         float* __restrict__ out, const ConvArgs a) {
    constexpr int CSTEP = IN_NCHW ? 1 : 4;
    static_assert(IN_NCHW || CIN % 4 == 0, "NHWC input needs a multiple of 4 channels");
    const int wg = (a.Wo + TP - 1) / TP;
    const int gid = blockIdx.x * PMN_BLOCK + threadIdx.x;
    const int total = a.N * a.Ho * wg;
    const bool live = gid < total;
    const int g = live ? gid : total - 1;
    const int ox0 = (g % wg) * TP;
    const int oy = (g / wg) % a.Ho;
    const int n = g / (wg * a.Ho);
    const int co0 = blockIdx.y * COUT_T;

    float acc[TP][COUT_T];
#pragma unroll
    for (int p = 0; p < TP; ++p)
#pragma unroll
        for (int c = 0; c < COUT_T; ++c) acc[p][c] = 0.0f;

    // weights / shifts are read through the constant address space so the wave-uniform loads become scalar (SMEM)
    // loads feeding SGPR operands; as plain global loads hipcc emits one vector load per weight quad
    typedef const float __attribute__((address_space(4))) cfloat;
    const cfloat* wt = (const cfloat*)(wgt) + co0;
#pragma unroll 1
    for (int ky = 0; ky < K; ++ky) {
        const int iy = oy * S + ky * a.dil - a.pad;
        const bool rowok = (unsigned)iy < (unsigned)a.H;
        const int iyc = rowok ? iy : 0;
#pragma unroll 1
        for (int kx = 0; kx < K; ++kx) {
            bool ok[TP];
            size_t base[TP];
#pragma unroll
            for (int p = 0; p < TP; ++p) {
                const int ix = (ox0 + p) * S + kx * a.dil - a.pad;
                ok[p] = rowok && (unsigned)ix < (unsigned)a.W;
                const int ixc = ok[p] ? ix : 0;
                base[p] = IN_NCHW ? ((size_t)n * CIN * a.H + iyc) * a.W + ixc : (((size_t)n * a.H + iyc) * a.W + ixc) * CIN;
            }
            const int wk_ofs = ((ky * K + kx) * CIN) * a.COUTP;
#pragma unroll 1
            for (int c0 = 0; c0 < CIN; c0 += CSTEP) {
                float v[TP][CSTEP];
#pragma unroll
                for (int p = 0; p < TP; ++p) {
                    if constexpr (IN_NCHW) {
                        v[p][0] = ok[p] ? in[base[p] + (size_t)c0 * a.H * a.W] : 0.0f;
                    } else {
                        const float4 t = ok[p] ? *reinterpret_cast<const float4*>(in + base[p] + c0)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
                        v[p][0] = t.x; v[p][1] = t.y; v[p][2] = t.z; v[p][3] = t.w;
                    }
                }
                const cfloat* wq = wt + __builtin_amdgcn_readfirstlane(wk_ofs + c0 * a.COUTP);
#pragma unroll
                for (int ci = 0; ci < CSTEP; ++ci) {
#pragma unroll
                    for (int c = 0; c < COUT_T; ++c) {
                        const float wv = wq[ci * a.COUTP + c];  // wave-uniform -> scalar load
#pragma unroll
                        for (int p = 0; p < TP; ++p) acc[p][c] = fmaf(v[p][ci], wv, acc[p][c]);
                    }
                }
            }
        }
    }
    if (!live) return;

    // epilogue: folded-BN shift / bias, optional FPN term (bilinear x2 of `up`, net.py:60,65), optional ReLU, store
    int uy0 = 0, uy1 = 0;
    float ly = 0.0f;
    if (up) up2_taps(oy, a.up_h, uy0, uy1, ly);
#pragma unroll
    for (int p = 0; p < TP; ++p) {
        const int ox = ox0 + p;
        if (ox >= a.Wo) break;
        float r[COUT_T];
#pragma unroll
        for (int c = 0; c < COUT_T; ++c) r[c] = acc[p][c] + shift[co0 + c];
        if (up) {
            int ux0, ux1;
            float lx;
            up2_taps(ox, a.up_w, ux0, ux1, lx);
            const float* u00 = up + (((size_t)n * a.up_h + uy0) * a.up_w + ux0) * a.COUTP + co0;
            const float* u01 = up + (((size_t)n * a.up_h + uy0) * a.up_w + ux1) * a.COUTP + co0;
            const float* u10 = up + (((size_t)n * a.up_h + uy1) * a.up_w + ux0) * a.COUTP + co0;
            const float* u11 = up + (((size_t)n * a.up_h + uy1) * a.up_w + ux1) * a.COUTP + co0;
            const float hy = 1.0f - ly, hx = 1.0f - lx;
#pragma unroll
            for (int c = 0; c < COUT_T; ++c)  // ATen: h0*(w0*a + w1*b) + h1*(w0*c + w1*d); the sum is upsample + conv
                r[c] = (hy * (hx * u00[c] + lx * u01[c]) + ly * (hx * u10[c] + lx * u11[c])) + r[c];
        }
        if (a.relu) {
#pragma unroll
            for (int c = 0; c < COUT_T; ++c) r[c] = fmaxf(r[c], 0.0f);
        }
        if (OUT_NCHW) {
#pragma unroll
            for (int c = 0; c < COUT_T; ++c)
                if (co0 + c < a.cout) out[(((size_t)n * a.cout + co0 + c) * a.Ho + oy) * a.Wo + ox] = r[c];
        } else {
            float* o = out + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.cout + co0;
#pragma unroll
            for (int c = 0; c < COUT_T; c += 4)
                if (co0 + c < a.cout) *reinterpret_cast<float4*>(o + c) = make_float4(r[c], r[c + 1], r[c + 2], r[c + 3]);
        }
    }
}

// ---- LDS-tiled variant for channels-last inputs ------------------------------------------------------------------------
// The direct kernel above reads 16 bytes per lane at a CIN*4-byte stride: the lines it touches are evicted from the
// 32 KB L1 before the neighbouring taps / channel chunks come back for them, so every access re-fetches 128 B from L2
// (measured: 8-14 TFLOP/s).  Here a workgroup owns a 16x16 tile of output pixels: per chunk of CC input channels it
// stages the (16*S + halo)^2 input patch in LDS with coalesced float4 loads (zero-filled outside the image), then every
// thread computes ONE output pixel x ALL COUT output channels from ds_read_b128's (pixel pitch CC+4 words: conflict
// free) and wave-uniform SGPR weights: 2*COUT packed FMAs per LDS read.
template <int CIN, int CC, int COUT, int COUTP, int K, int S, bool OUT_NCHW, bool UP>
__global__ __launch_bounds__(PMN_BLOCK, 4) void conv_tiled_kernel(const float* __restrict__ in, const float* __restrict__ wgt,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ up, float* __restrict__ out,
                                                               const ConvArgs a) {
    // pixel pitch in LDS: CC+4 words makes unit-stride ds_read_b128 conflict-free; stride-2 readers do best unpadded
    constexpr int TW = 16, TH = 16, CCP = (S == 2 && CC == 4) ? 4 : CC + 4, CQ = CC / 4;
    static_assert(CIN % CC == 0 && CC % 4 == 0 && COUT % 8 == 0, "channel tiling");
    extern __shared__ float4 tile4[];
    float* tile = reinterpret_cast<float*>(tile4);
    typedef const float __attribute__((address_space(4))) cfloat;
    const int co0 = blockIdx.y * COUT;  // this block's slice of the output channels (COUT of a.COUTP)
    const cfloat* wt = (const cfloat*)(wgt) + co0;
    const int tid = threadIdx.x, tx = tid % TW, ty = tid / TW;
    const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
    const int bt = pmn_xcd_tile(blockIdx.x, a.N * tiles_x * tiles_y);
    const int n = bt / (tiles_x * tiles_y), tr = bt - n * tiles_x * tiles_y;
    const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TW;
    const int iw = (TW - 1) * S + (K - 1) * a.dil + 1, ih = (TH - 1) * S + (K - 1) * a.dil + 1;
    const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;

    // The FPN term (bilinear x2 of `up`, net.py:60,65) seeds the accumulators -- loaded while nothing else is live --
    // instead of being added in the epilogue (where 64 live accumulators + 16 quads in flight spill).
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = 0.0f;
    if constexpr (UP) {
        const int oyc = min(oy0 + ty, a.Ho - 1), oxc = min(ox0 + tx, a.Wo - 1);
        int uy0, uy1, ux0, ux1;
        float ly, lx;
        up2_taps(oyc, a.up_h, uy0, uy1, ly);
        up2_taps(oxc, a.up_w, ux0, ux1, lx);
        const float* u00 = up + (((size_t)n * a.up_h + uy0) * a.up_w + ux0) * COUTP + co0;
        const float* u01 = up + (((size_t)n * a.up_h + uy0) * a.up_w + ux1) * COUTP + co0;
        const float* u10 = up + (((size_t)n * a.up_h + uy1) * a.up_w + ux0) * COUTP + co0;
        const float* u11 = up + (((size_t)n * a.up_h + uy1) * a.up_w + ux1) * COUTP + co0;
        const float hy = 1.0f - ly, hx = 1.0f - lx;
#pragma unroll
        for (int c = 0; c < COUT; c += 4) {
            const float4 p00 = *reinterpret_cast<const float4*>(u00 + c), p01 = *reinterpret_cast<const float4*>(u01 + c);
            const float4 p10 = *reinterpret_cast<const float4*>(u10 + c), p11 = *reinterpret_cast<const float4*>(u11 + c);
            // ATen upsample_bilinear2d: h0*(w0*a + w1*b) + h1*(w0*c + w1*d)
            acc[c] = hy * (hx * p00.x + lx * p01.x) + ly * (hx * p10.x + lx * p11.x);
            acc[c + 1] = hy * (hx * p00.y + lx * p01.y) + ly * (hx * p10.y + lx * p11.y);
            acc[c + 2] = hy * (hx * p00.z + lx * p01.z) + ly * (hx * p10.z + lx * p11.z);
            acc[c + 3] = hy * (hx * p00.w + lx * p01.w) + ly * (hx * p10.w + lx * p11.w);
            if ((c & 4) == 4) __builtin_amdgcn_sched_barrier(0);  // 8 channels (8 x dwordx4) in flight at a time
        }
    }

#pragma unroll 1
    for (int cc0 = 0; cc0 < CIN; cc0 += CC) {
        if (cc0) __syncthreads();
        // stage: ih x iw pixels x CC channels
        // batches of SB loads per thread in flight (one load per trip exposes a full memory round trip per 16 bytes)
        constexpr int SB = 4;
        for (int base = tid; base < ih * iw * CQ; base += PMN_BLOCK * SB) {
            float4 v[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int idx = base + u * PMN_BLOCK, pix = idx / CQ, q = idx - pix * CQ;
                const int r = pix / iw, c = pix - r * iw;
                const int gy = iy0 + r, gx = ix0 + c;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < ih * iw * CQ && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
                    v[u] = *reinterpret_cast<const float4*>(in + (((size_t)n * a.H + gy) * a.W + gx) * CIN + cc0 + 4 * q);
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int idx = base + u * PMN_BLOCK, pix = idx / CQ, q = idx - pix * CQ;
                if (idx < ih * iw * CQ) *reinterpret_cast<float4*>(tile + pix * CCP + 4 * q) = v[u];
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int ky = 0; ky < K; ++ky) {
#pragma unroll 1
            for (int kx = 0; kx < K; ++kx) {
                const float* tp = tile + ((ty * S + ky * a.dil) * iw + tx * S + kx * a.dil) * CCP;
#pragma unroll 1
                for (int q = 0; q < CQ; ++q) {
                    const float4 v4 = *reinterpret_cast<const float4*>(tp + 4 * q);
                    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
                    // readfirstlane pins the (wave-uniform) weight offset in an SGPR: in some instantiations hipcc
                    // otherwise strength-reduces it into a VGPR pointer and falls back to per-lane vector loads
                    const cfloat* wq = wt + __builtin_amdgcn_readfirstlane(((ky * K + kx) * CIN + cc0 + 4 * q) * COUTP);
#pragma unroll
                    for (int cog = 0; cog < COUT; cog += 16) {
#pragma unroll
                        for (int ci = 0; ci < 4; ++ci) {
#pragma unroll
                            for (int c = 0; c < (COUT < 16 ? COUT : 16); ++c)
                                acc[cog + c] = fmaf(v[ci], wq[ci * COUTP + cog + c], acc[cog + c]);
                        }
                        // one 64-weight batch (4 x s_load_dwordx16) at a time: unfenced, hipcc hoists all 4*COUT weights
                        // of the step into VGPRs and the kernel drops to one wave per SIMD
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    }

    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= a.Ho || ox >= a.Wo) return;
    const cfloat* sh = (const cfloat*)(shift) + co0;
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] += sh[c];
    if (a.relu) {
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[c] = fmaxf(acc[c], 0.0f);
    }
    if (OUT_NCHW) {
#pragma unroll
        for (int c = 0; c < COUT; ++c)
            if (co0 + c < a.cout) out[(((size_t)n * a.cout + co0 + c) * a.Ho + oy) * a.Wo + ox] = acc[c];
    } else {
        float* o = out + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.cout + co0;
#pragma unroll
        for (int c = 0; c < COUT; c += 4)
            if (co0 + c < a.cout) *reinterpret_cast<float4*>(o + c) = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
    }
}

template <int CIN, int CC, int COUT, int COUTP, int K, int S, bool OUT_NCHW, bool UP>
static int launch_tiled_impl(const float* in, const float* w, const float* shift, const float* up, float* out, ConvArgs a,
                             hipStream_t st) {
    const int iw = 15 * S + (K - 1) * a.dil + 1;
    const size_t lds = (size_t)iw * iw * ((S == 2 && CC == 4) ? 4 : CC + 4) * sizeof(float);
    if (lds > 160 * 1024) return PMN_ERR_SHAPE;
    auto kern = conv_tiled_kernel<CIN, CC, COUT, COUTP, K, S, OUT_NCHW, UP>;
    if (lds > 48 * 1024 && pmn_raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != PMN_OK) return PMN_ERR_LAUNCH;
    const int blocks = a.N * ((a.Wo + 15) / 16) * ((a.Ho + 15) / 16);
    PMN_LAUNCH(kern, dim3(blocks, COUTP / COUT), dim3(PMN_BLOCK), lds, st, in, w, shift, up, out, a);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

template <int CIN, int COUT_T, int K, int S, bool IN_NCHW, bool OUT_NCHW>
static int launch_conv(const float* in, const float* w, const float* shift, const float* up, float* out, ConvArgs a,
                       hipStream_t st) {
    // TP = 4 pixels per thread when that still leaves >= ~8 waves per SIMD-quad of work, else 2
    const long pix = (long)a.N * a.Ho * a.Wo;
    const dim3 gy(1, (a.cout + COUT_T - 1) / COUT_T);
    if (pix >= 1500000L) {
        const int wg = (a.Wo + 3) / 4;
        const long thr = (long)a.N * a.Ho * wg;
        PMN_LAUNCH((conv_kernel<CIN, COUT_T, K, S, 4, IN_NCHW, OUT_NCHW>), dim3((thr + PMN_BLOCK - 1) / PMN_BLOCK, gy.y),
                           dim3(PMN_BLOCK), 0, st, in, w, shift, up, out, a);
    } else {
        const int wg = (a.Wo + 1) / 2;
        const long thr = (long)a.N * a.Ho * wg;
        PMN_LAUNCH((conv_kernel<CIN, COUT_T, K, S, 2, IN_NCHW, OUT_NCHW>), dim3((thr + PMN_BLOCK - 1) / PMN_BLOCK, gy.y),
                           dim3(PMN_BLOCK), 0, st, in, w, shift, up, out, a);
    }
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

template <int CIN, int CC, int COUT, int K, int S, bool OUT_NCHW>
static int launch_tiled(const float* in, const float* w, const float* shift, const float* up, float* out, ConvArgs a,
                        hipStream_t st) {
    if constexpr (K == 1 && COUT == 64 && !OUT_NCHW) {  // the FPN lateral 1x1 convs are the only users of `up`:
        // two blocks of 32 output channels each (64 seeded accumulators + the up-sampling loads would spill)
        if (up) return launch_tiled_impl<CIN, CC, 32, 64, K, S, OUT_NCHW, true>(in, w, shift, up, out, a, st);
    }
    if (up) return PMN_ERR_SHAPE;
    return launch_tiled_impl<CIN, CC, COUT, COUT, K, S, OUT_NCHW, false>(in, w, shift, up, out, a, st);
}

// ---- fused stem: conv0 (3 -> 8) + conv1 (8 -> 8), both 3x3 / BatchNorm / ReLU, at full resolution (net.py:17-19, 51) -----
// The two full-resolution layers are bandwidth-shaped (8 channels): fusing them keeps conv0's 8-channel map (61 MB per
// 1600x1200 image) out of HBM.  Workgroup = 16x16 output pixels: the planar 20x20x3 input patch goes to LDS, conv0 is
// evaluated on the 18x18 halo patch into LDS (positions outside the image are ZERO: conv1 pads conv0's output map, not the
// image), then every thread produces one conv1 pixel from ds_read_b128's and SGPR weights.
__global__ __launch_bounds__(PMN_BLOCK) void stem_kernel(const float* __restrict__ img, const float* __restrict__ w0,
                                                         const float* __restrict__ s0, const float* __restrict__ w1,
                                                         const float* __restrict__ s1, float* __restrict__ out, int N,
                                                         int H, int W) {
    // input patch 20x20 (pitch 21); mid 18x18 pixels x 12 words with a 256-word ROW pitch: 12/4 is odd and the row pitch is a
    // multiple of 64 banks, so the ds_read_b128 of conv1 (16 lanes = 8 + 8 pixels of two tile rows) touches every bank once
    constexpr int TW = 16, TH = 16, IW = 20, IWP = 21, MW = 18, MP = 12, MRP = 256;
    __shared__ float xin[3 * IW * IWP];
    __shared__ float4 mid4[MW * MRP / 4];
    float* mid = reinterpret_cast<float*>(mid4);
    typedef const float __attribute__((address_space(4))) cfloat;
    const cfloat* cw0 = (const cfloat*)w0;  // [3][3][3][8]
    const cfloat* cs0 = (const cfloat*)s0;
    const cfloat* cw1 = (const cfloat*)w1;  // [3][3][8][8]
    const cfloat* cs1 = (const cfloat*)s1;
    const int tid = threadIdx.x, tx = tid % TW, ty = tid / TW;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int bt = pmn_xcd_tile(blockIdx.x, N * tiles_x * tiles_y);
    const int n = bt / (tiles_x * tiles_y), tr = bt - n * tiles_x * tiles_y;
    const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TW;

    {   // all of a thread's patch loads in flight at once (one load per loop trip exposes a memory round trip per 4 bytes)
        constexpr int NL = (3 * IW * IW + PMN_BLOCK - 1) / PMN_BLOCK;
        float v[NL];
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int idx = tid + u * PMN_BLOCK;
            const int c = idx / (IW * IW), r = (idx / IW) % IW, q = idx % IW;
            const int gy = oy0 - 2 + r, gx = ox0 - 2 + q;
            v[u] = 0.0f;
            if (idx < 3 * IW * IW && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                v[u] = img[(((size_t)n * 3 + c) * H + gy) * W + gx];
        }
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int idx = tid + u * PMN_BLOCK;
            const int c = idx / (IW * IW), r = (idx / IW) % IW, q = idx % IW;
            if (idx < 3 * IW * IW) xin[(c * IW + r) * IWP + q] = v[u];
        }
    }
    __syncthreads();
    for (int m = tid; m < MW * MW; m += PMN_BLOCK) {
        const int r = m / MW, q = m - r * MW;
        const int gy = oy0 - 1 + r, gx = ox0 - 1 + q;
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.0f;
        // tap rows ROLLED here and below: fully unrolled, hipcc's SLP pass gives up on the block and emits 792 scalar v_fma_f32
        // (half the packed-FMA rate) for the kernel
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
            const cfloat* wq = cw0 + __builtin_amdgcn_readfirstlane(ky * 72);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const float v = xin[(ci * IW + r + ky) * IWP + q + kx];
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[c] = fmaf(v, wq[(kx * 3 + ci) * 8 + c], acc[c]);
                }
        }
        const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = inside ? fmaxf(acc[c] + cs0[c], 0.0f) : 0.0f;
        *reinterpret_cast<float4*>(mid + r * MRP + q * MP) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(mid + r * MRP + q * MP + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
    __syncthreads();
    float o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = 0.0f;
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll 1
        for (int kx = 0; kx < 3; ++kx) {
            const float* mp = mid + (ty + ky) * MRP + (tx + kx) * MP;
            const float4 a = *reinterpret_cast<const float4*>(mp), b = *reinterpret_cast<const float4*>(mp + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            const cfloat* wq = cw1 + __builtin_amdgcn_readfirstlane((ky * 3 + kx) * 64);
#pragma unroll
            for (int ci = 0; ci < 8; ++ci)
#pragma unroll
                for (int c = 0; c < 8; ++c) o[c] = fmaf(v[ci], wq[ci * 8 + c], o[c]);
        }
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= H || ox >= W) return;
    float* op = out + (((size_t)n * H + oy) * W + ox) * 8;
    *reinterpret_cast<float4*>(op) = make_float4(fmaxf(o[0] + cs1[0], 0.f), fmaxf(o[1] + cs1[1], 0.f), fmaxf(o[2] + cs1[2], 0.f),
                                                 fmaxf(o[3] + cs1[3], 0.f));
    *reinterpret_cast<float4*>(op + 4) = make_float4(fmaxf(o[4] + cs1[4], 0.f), fmaxf(o[5] + cs1[5], 0.f),
                                                     fmaxf(o[6] + cs1[6], 0.f), fmaxf(o[7] + cs1[7], 0.f));
}

// img [N,3,H,W] planar; w0 [3][3][3][8] / s0 [8] and w1 [3][3][8][8] / s1 [8] in pack_conv layout -> out [N,H,W,8]
extern "C" int pmn_stem(const float* img, const float* w0, const float* s0, const float* w1, const float* s1, float* out,
                        int N, int H, int W, void* stream) {
    if (!img || !w0 || !s0 || !w1 || !s1 || !out || N < 1 || H < 1 || W < 1) return PMN_ERR_ARG;
    const int blocks = N * ((W + 15) / 16) * ((H + 15) / 16);
    PMN_LAUNCH(stem_kernel, dim3(blocks), dim3(PMN_BLOCK), 0, (hipStream_t)stream, img, w0, s0, w1, s1, out, N, H, W);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

// ---- fused tail of the FPN (reference models/net.py:64-67) ---------------------------------------------------------------
//   intra = bilinear_x2(top) + inner2(half)        (1x1 conv 16 -> 64, bias)
//   out   = output3(intra)                         (1x1 conv 64 -> 16, no bias)
// `intra` (64 channels at half resolution: 737 MB for six 1600x1200 views) is consumed only by output3, so it never
// leaves registers: a thread owns one pixel, builds the 64 intermediate channels in two halves of 32 (up-sampling taps
// + SGPR-weight FMAs) and folds each half straight into the 16 outputs.
template <int CIN, int CMID, int COUT>
__global__ __launch_bounds__(PMN_BLOCK, 2) void fpn_tail_kernel(const float* __restrict__ x, const float* __restrict__ up,
                                                              const float* __restrict__ w_in,
                                                              const float* __restrict__ b_in,
                                                              const float* __restrict__ w_out, float* __restrict__ out,
                                                              int N, int H, int W) {
    // 16x16 output tile per workgroup; the x tile and the 10x10 patch of `up` it samples are staged in LDS with coalesced
    // float4 loads (per-lane 16-byte reads at a 64/256-byte stride thrash the L1, see conv_tiled_kernel)
    constexpr int TW = 16, TH = 16, XP = CIN + 4, UPW = 10, UPP = CMID + 4;
    __shared__ float4 xs4[TW * TH * XP / 4];
    __shared__ float4 us4[UPW * UPW * UPP / 4];
    float* xs = reinterpret_cast<float*>(xs4);
    float* us = reinterpret_cast<float*>(us4);
    typedef const float __attribute__((address_space(4))) cfloat;
    const cfloat* wi = (const cfloat*)w_in;   // [CIN][CMID]
    const cfloat* bi = (const cfloat*)b_in;   // [CMID]
    const cfloat* wo = (const cfloat*)w_out;  // [CMID][COUT]
    const int tid = threadIdx.x, tx = tid % TW, ty = tid / TW;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int bt = pmn_xcd_tile(blockIdx.x, N * tiles_x * tiles_y);
    const int n = bt / (tiles_x * tiles_y), tr = bt - n * tiles_x * tiles_y;
    const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TW;
    const int uh = H / 2, uw = W / 2;
    const int uy_base = max(oy0 / 2 - 1, 0), ux_base = max(ox0 / 2 - 1, 0);

    for (int idx = tid; idx < TW * TH * (CIN / 4); idx += PMN_BLOCK) {
        const int pix = idx / (CIN / 4), q = idx - pix * (CIN / 4);
        const int gy = min(oy0 + pix / TW, H - 1), gx = min(ox0 + pix % TW, W - 1);
        *reinterpret_cast<float4*>(xs + pix * XP + 4 * q) =
            *reinterpret_cast<const float4*>(x + (((size_t)n * H + gy) * W + gx) * CIN + 4 * q);
    }
    for (int idx = tid; idx < UPW * UPW * (CMID / 4); idx += PMN_BLOCK) {
        const int pix = idx / (CMID / 4), q = idx - pix * (CMID / 4);
        const int gy = min(uy_base + pix / UPW, uh - 1), gx = min(ux_base + pix % UPW, uw - 1);
        *reinterpret_cast<float4*>(us + pix * UPP + 4 * q) =
            *reinterpret_cast<const float4*>(up + (((size_t)n * uh + gy) * uw + gx) * CMID + 4 * q);
    }
    __syncthreads();

    const int oy = min(oy0 + ty, H - 1), ox = min(ox0 + tx, W - 1);
    float xin[CIN];
#pragma unroll
    for (int c = 0; c < CIN; c += 4) {
        const float4 t = *reinterpret_cast<const float4*>(xs + tid * XP + c);
        xin[c] = t.x; xin[c + 1] = t.y; xin[c + 2] = t.z; xin[c + 3] = t.w;
    }
    int uy0, uy1, ux0, ux1;
    float ly, lx;
    up2_taps(oy, uh, uy0, uy1, ly);
    up2_taps(ox, uw, ux0, ux1, lx);
    const float* u00 = us + ((uy0 - uy_base) * UPW + (ux0 - ux_base)) * UPP;
    const float* u01 = us + ((uy0 - uy_base) * UPW + (ux1 - ux_base)) * UPP;
    const float* u10 = us + ((uy1 - uy_base) * UPW + (ux0 - ux_base)) * UPP;
    const float* u11 = us + ((uy1 - uy_base) * UPW + (ux1 - ux_base)) * UPP;
    const float hy = 1.0f - ly, hx = 1.0f - lx;

    float o[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) o[c] = 0.0f;
#pragma unroll 1
    for (int m0 = 0; m0 < CMID; m0 += 32) {
        float t[32];
#pragma unroll
        for (int c = 0; c < 32; c += 4) {
            const float4 p00 = *reinterpret_cast<const float4*>(u00 + m0 + c), p01 = *reinterpret_cast<const float4*>(u01 + m0 + c);
            const float4 p10 = *reinterpret_cast<const float4*>(u10 + m0 + c), p11 = *reinterpret_cast<const float4*>(u11 + m0 + c);
            t[c] = hy * (hx * p00.x + lx * p01.x) + ly * (hx * p10.x + lx * p11.x);
            t[c + 1] = hy * (hx * p00.y + lx * p01.y) + ly * (hx * p10.y + lx * p11.y);
            t[c + 2] = hy * (hx * p00.z + lx * p01.z) + ly * (hx * p10.z + lx * p11.z);
            t[c + 3] = hy * (hx * p00.w + lx * p01.w) + ly * (hx * p10.w + lx * p11.w);
            if ((c & 4) == 4) __builtin_amdgcn_sched_barrier(0);
        }
        // inner conv: t[c] = up + (sum_ci x[ci]*w[ci][m0+c] + b)   (same association as net.py:65: upsample + inner2(x))
#pragma unroll
        for (int cg = 0; cg < 32; cg += 16) {
            float s[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) s[c] = 0.0f;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                for (int c = 0; c < 16; ++c) s[c] = fmaf(xin[ci], wi[ci * CMID + m0 + cg + c], s[c]);
#pragma unroll
            for (int c = 0; c < 16; ++c) t[cg + c] = t[cg + c] + pmn_settle(s[c] + bi[m0 + cg + c]);  // (a scalar sum hipcc packs: lesson 46)
            __builtin_amdgcn_sched_barrier(0);
        }
        // output conv on this half of the intermediate channels
#pragma unroll
        for (int cm = 0; cm < 32; cm += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < COUT; ++c) o[c] = fmaf(t[cm + k], wo[(m0 + cm + k) * COUT + c], o[c]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (oy0 + ty >= H || ox0 + tx >= W) return;
    float* op = out + (((size_t)n * H + oy) * W + ox) * COUT;
#pragma unroll
    for (int c = 0; c < COUT; c += 4) *reinterpret_cast<float4*>(op + c) = make_float4(o[c], o[c + 1], o[c + 2], o[c + 3]);
}

// x [N,H,W,16] (conv4 output), up [N,H/2,W/2,64] (previous FPN level), w_in [16][64] / b_in [64] (inner2, pack_conv layout),
// w_out [64][16] (output3, pack_conv layout) -> out [N,H,W,16].
extern "C" int pmn_fpn_tail(const float* x, const float* up, const float* w_in, const float* b_in, const float* w_out,
                            float* out, int N, int H, int W, int cin, int cmid, int cout, void* stream) {
    if (!x || !up || !w_in || !b_in || !w_out || !out || N < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) return PMN_ERR_ARG;
    if (cin != 16 || cmid != 64 || cout != 16) return PMN_ERR_SHAPE;
    const int blocks = N * ((W + 15) / 16) * ((H + 15) / 16);
    PMN_LAUNCH((fpn_tail_kernel<16, 64, 16>), dim3(blocks), dim3(PMN_BLOCK), 0, (hipStream_t)stream, x, up, w_in, b_in,
                       w_out, out, N, H, W);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

// ---- one level of the FOLDED FPN head ------------------------------------------------------------------------------------
// The FPN head (reference models/net.py:57-67) has no non-linearity: output_k(upsample(intra) + inner_k(conv)) distributes
// over the sum, and a 1x1 convolution commutes with bilinear up-sampling (its taps sum to 1, so biases pass through).  With
//   W8 = [output1; output2; output3]            (112 x 64, applied to conv10 at 1/8)
//   W4 = [output2; output3] . inner1, b4 = [output2; output3] . b_inner1      (48 x 32, applied to conv7 at 1/4)
//   W2 = output3 . inner2,            b2 = output3 . b_inner2                 (16 x 16, applied to conv4 at 1/2)
// (products formed in fp64 on the host, params.fold_fpn) the three feature maps are
//   [f3 | u] = W8 conv10        [f2 | t] = up2(u) + W4 conv7 + b4        f1 = up2(t) + W2 conv4 + b2
// and the 64-channel intermediates (184 MB at 1/4 and 737 MB at 1/2 for six 1600x1200 views) never exist.  One kernel per
// level: out[c] = up2(u)[c] + b[c] + sum_ci x[ci] w[ci][c]; channels [0,CA) go to outA, the rest to outB.  A workgroup owns
// 16x16 pixels, stages x and the 10x10 patch of u it samples in LDS with coalesced float4 loads, then every thread
// produces its pixel 16 channels at a time from ds_read_b128's and wave-uniform SGPR weights.
template <int CIN, int COUT, int CA, bool UP>
__global__ __launch_bounds__(PMN_BLOCK, 2) void fpn_level_kernel(const float* __restrict__ x, const float* __restrict__ u,
                                                               const float* __restrict__ wgt,
                                                               const float* __restrict__ bias, float* __restrict__ outA,
                                                               float* __restrict__ outB, int N, int H, int W) {
    constexpr int TW = 16, TH = 16, XP = CIN + 4, UPW = 10, UPP = COUT + 4, CB = COUT - CA;
    static_assert(CIN % 4 == 0 && COUT % 16 == 0 && CA % 16 == 0, "channel tiling");
    extern __shared__ float4 fl_lds4[];
    float* xs = reinterpret_cast<float*>(fl_lds4);
    float* us = xs + TW * TH * XP;
    typedef const float __attribute__((address_space(4))) cfloat;
    const cfloat* wt = (const cfloat*)wgt;  // [CIN][COUT]
    const cfloat* bs = (const cfloat*)bias;  // [COUT]
    const int tid = threadIdx.x, tx = tid % TW, ty = tid / TW;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int bt = pmn_xcd_tile(blockIdx.x, N * tiles_x * tiles_y);
    const int n = bt / (tiles_x * tiles_y), tr = bt - n * tiles_x * tiles_y;
    const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TW;
    const int uh = H / 2, uw = W / 2;
    const int uy_base = max(oy0 / 2 - 1, 0), ux_base = max(ox0 / 2 - 1, 0);

    {   // every load of the tile in flight before the first LDS write (a load per loop trip exposes a memory round trip each)
        constexpr int NX = CIN / 4, NU = UP ? (UPW * UPW * (COUT / 4) + PMN_BLOCK - 1) / PMN_BLOCK : 0;
        float4 vx[NX], vu[NU > 0 ? NU : 1];
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int idx = tid + k * PMN_BLOCK, pix = idx / NX, q = idx - pix * NX;
            const int gy = min(oy0 + pix / TW, H - 1), gx = min(ox0 + pix % TW, W - 1);
            vx[k] = *reinterpret_cast<const float4*>(x + (((size_t)n * H + gy) * W + gx) * CIN + 4 * q);
        }
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            const int idx = min(tid + k * PMN_BLOCK, UPW * UPW * (COUT / 4) - 1), pix = idx / (COUT / 4), q = idx - pix * (COUT / 4);
            const int gy = min(uy_base + pix / UPW, uh - 1), gx = min(ux_base + pix % UPW, uw - 1);
            vu[k] = *reinterpret_cast<const float4*>(u + (((size_t)n * uh + gy) * uw + gx) * COUT + 4 * q);
        }
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int idx = tid + k * PMN_BLOCK, pix = idx / NX, q = idx - pix * NX;
            *reinterpret_cast<float4*>(xs + pix * XP + 4 * q) = vx[k];
        }
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            const int idx = tid + k * PMN_BLOCK, pix = idx / (COUT / 4), q = idx - pix * (COUT / 4);
            if (idx < UPW * UPW * (COUT / 4)) *reinterpret_cast<float4*>(us + pix * UPP + 4 * q) = vu[k];
        }
    }
    __syncthreads();

    const int oy = min(oy0 + ty, H - 1), ox = min(ox0 + tx, W - 1);
    const bool live = oy0 + ty < H && ox0 + tx < W;
    int uy0 = 0, uy1 = 0, ux0 = 0, ux1 = 0;
    float ly = 0.f, lx = 0.f;
    if constexpr (UP) {
        up2_taps(oy, uh, uy0, uy1, ly);
        up2_taps(ox, uw, ux0, ux1, lx);
    }
    const float* u00 = us + ((uy0 - uy_base) * UPW + (ux0 - ux_base)) * UPP;
    const float* u01 = us + ((uy0 - uy_base) * UPW + (ux1 - ux_base)) * UPP;
    const float* u10 = us + ((uy1 - uy_base) * UPW + (ux0 - ux_base)) * UPP;
    const float* u11 = us + ((uy1 - uy_base) * UPW + (ux1 - ux_base)) * UPP;
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    const float* xp = xs + tid * XP;
    const size_t opix = ((size_t)n * H + oy) * W + ox;

#pragma unroll 1
    for (int c0 = 0; c0 < COUT; c0 += 16) {
        float acc[16];
        if constexpr (UP) {
#pragma unroll
            for (int c = 0; c < 16; c += 4) {
                const float4 p00 = *reinterpret_cast<const float4*>(u00 + c0 + c), p01 = *reinterpret_cast<const float4*>(u01 + c0 + c);
                const float4 p10 = *reinterpret_cast<const float4*>(u10 + c0 + c), p11 = *reinterpret_cast<const float4*>(u11 + c0 + c);
                // ATen upsample_bilinear2d: h0*(w0*a + w1*b) + h1*(w0*c + w1*d)
                acc[c] = hy * (hx * p00.x + lx * p01.x) + ly * (hx * p10.x + lx * p11.x);
                acc[c + 1] = hy * (hx * p00.y + lx * p01.y) + ly * (hx * p10.y + lx * p11.y);
                acc[c + 2] = hy * (hx * p00.z + lx * p01.z) + ly * (hx * p10.z + lx * p11.z);
                acc[c + 3] = hy * (hx * p00.w + lx * p01.w) + ly * (hx * p10.w + lx * p11.w);
            }
            const cfloat* bq = bs + __builtin_amdgcn_readfirstlane(c0);
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] += bq[c];
        } else {
            const cfloat* bq = bs + __builtin_amdgcn_readfirstlane(c0);
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] = bq[c];
        }
#pragma unroll
        for (int ci = 0; ci < CIN; ci += 4) {
            const float4 v4 = *reinterpret_cast<const float4*>(xp + ci);
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
            const cfloat* wq = wt + __builtin_amdgcn_readfirstlane(ci * COUT + c0);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[c] = fmaf(v[k], wq[k * COUT + c], acc[c]);
            __builtin_amdgcn_sched_barrier(0);  // one 64-weight batch in SGPRs at a time
        }
        if (live) {
            float* o = (CB == 0 || c0 < CA) ? outA + opix * CA + c0 : outB + opix * (CB > 0 ? CB : 1) + (c0 - CA);
#pragma unroll
            for (int c = 0; c < 16; c += 4)
                *reinterpret_cast<float4*>(o + c) = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
        }
    }
}

template <int CIN, int COUT, int CA, bool UP>
static int launch_fpn_level(const float* x, const float* u, const float* w, const float* b, float* outA, float* outB, int N,
                            int H, int W, hipStream_t st) {
    const size_t lds = (size_t)(16 * 16 * (CIN + 4) + (UP ? 10 * 10 * (COUT + 4) : 0)) * sizeof(float);
    auto kern = fpn_level_kernel<CIN, COUT, CA, UP>;
    if (lds > 48 * 1024 && pmn_raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != PMN_OK) return PMN_ERR_LAUNCH;
    const int blocks = N * ((W + 15) / 16) * ((H + 15) / 16);
    PMN_LAUNCH(kern, dim3(blocks), dim3(PMN_BLOCK), lds, st, x, u, w, b, outA, outB, N, H, W);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

// x [N,H,W,cin]; u [N,H/2,W/2,cout] or NULL (no up-sampled term); w [cin][cout], b [cout] (params.fold_fpn);
// out_a [N,H,W,ca] receives channels [0,ca), out_b [N,H,W,cout-ca] the rest (NULL when ca == cout).
extern "C" int pmn_fpn_level(const float* x, const float* u, const float* w, const float* b, float* out_a, float* out_b,
                             int N, int H, int W, int cin, int cout, int ca, void* stream) {
    if (!x || !w || !b || !out_a || N < 1 || H < 1 || W < 1 || ca < 1 || ca > cout) return PMN_ERR_ARG;
    if ((ca < cout) != (out_b != nullptr)) return PMN_ERR_ARG;
    if (u && ((H & 1) || (W & 1))) return PMN_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (!u && cin == 64 && cout == 112 && ca == 64) return launch_fpn_level<64, 112, 64, false>(x, u, w, b, out_a, out_b, N, H, W, st);
    if (u && cin == 32 && cout == 48 && ca == 32) return launch_fpn_level<32, 48, 32, true>(x, u, w, b, out_a, out_b, N, H, W, st);
    if (u && cin == 16 && cout == 16 && ca == 16) return launch_fpn_level<16, 16, 16, true>(x, u, w, b, out_a, out_b, N, H, W, st);
    return PMN_ERR_SHAPE;
}

// ---- ConvTranspose2d(k=3, stride=2, padding=1, output_padding=1) + folded BatchNorm + ReLU (Refinement, net.py:86-88,114) ----
// out[oy,ox,co] = sum_{ky,kx,ci} in[(oy+1-ky)/2, (ox+1-kx)/2, ci] * w[ky][kx][ci][co] over the (ky,kx) for which both
// source coordinates are integral and in range: one tap for even coordinates (k=1), two for odd ones (k=0 and k=2).
template <int CIN, int COUT>
__global__ __launch_bounds__(PMN_BLOCK) void deconv3x3s2_kernel(const float* __restrict__ in, const float* __restrict__ wgt,
                                                               const float* __restrict__ shift, float* __restrict__ out,
                                                               int N, int Hi, int Wi, int relu) {
    typedef const float __attribute__((address_space(4))) cfloat;
    const cfloat* wt = (const cfloat*)wgt;
    const cfloat* sh = (const cfloat*)shift;
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    const size_t total = (size_t)N * Ho * Wo;
    const size_t gid = (size_t)blockIdx.x * PMN_BLOCK + threadIdx.x;
    if (gid >= total) return;
    const int ox = (int)(gid % Wo), oy = (int)((gid / Wo) % Ho), n = (int)(gid / ((size_t)Wo * Ho));
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = 0.0f;
    // tap loops rolled: fully unrolled, the 576 wave-uniform weights are hoisted into SGPRs at once and spilled through
    // v_readlane / v_writelane (706 of them)
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {
        const int ty = oy + 1 - ky;
        const bool yok = ty >= 0 && !(ty & 1) && (ty >> 1) < Hi;
#pragma unroll 1
        for (int kx = 0; kx < 3; ++kx) {
            const int tx = ox + 1 - kx;
            const bool ok = yok && tx >= 0 && !(tx & 1) && (tx >> 1) < Wi;
            const float* ip = in + (((size_t)n * Hi + (ok ? (ty >> 1) : 0)) * Wi + (ok ? (tx >> 1) : 0)) * CIN;
            const cfloat* wq = wt + __builtin_amdgcn_readfirstlane((ky * 3 + kx) * CIN * COUT);
            float v[CIN];
#pragma unroll
            for (int c = 0; c < CIN; c += 4) {
                const float4 t = ok ? *reinterpret_cast<const float4*>(ip + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                v[c] = t.x; v[c + 1] = t.y; v[c + 2] = t.z; v[c + 3] = t.w;
            }
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                for (int c = 0; c < COUT; ++c) acc[c] = fmaf(v[ci], wq[ci * COUT + c], acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
        acc[c] += sh[c];
        if (relu) acc[c] = fmaxf(acc[c], 0.0f);
    }
#pragma unroll
    for (int c = 0; c < COUT; c += 4)
        *reinterpret_cast<float4*>(out + gid * COUT + c) = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
}

// in [N,Hi,Wi,8], weights [3][3][8][8] (ConvTranspose2d weight [ci][co][ky][kx] re-ordered, BatchNorm scale folded in),
// shift [8] -> out [N,2*Hi,2*Wi,8]
extern "C" int pmn_deconv3x3s2(const float* in, const float* weights, const float* shift, float* out, int N, int Hi, int Wi,
                               int cin, int cout, int relu, void* stream) {
    if (!in || !weights || !shift || !out || N < 1 || Hi < 1 || Wi < 1) return PMN_ERR_ARG;
    if (cin != 8 || cout != 8) return PMN_ERR_SHAPE;
    const size_t total = (size_t)N * Hi * Wi * 4;
    PMN_LAUNCH((deconv3x3s2_kernel<8, 8>), dim3((unsigned)((total + PMN_BLOCK - 1) / PMN_BLOCK)), dim3(PMN_BLOCK), 0,
                       (hipStream_t)stream, in, weights, shift, out, N, Hi, Wi, relu);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

// in: [N,H,W,CIN] (or [N,CIN,H,W] when in_nchw); weights packed [K][K][CIN][COUTP] with COUTP = cout rounded up to the
// channel tile (8 if cout <= 8 else 16), BatchNorm scale folded in; shift[COUTP]; up: optional [N,up_h,up_w,cout] map that
// is bilinearly up-sampled x2 and added before the ReLU; out: [N,Ho,Wo,cout] (or [N,cout,Ho,Wo] when out_nchw).
extern "C" int pmn_conv2d(const float* in, const float* weights, const float* shift, const float* up, float* out, int N,
                          int H, int W, int cin, int cout, int K, int stride, int pad, int dil, int relu, int in_nchw,
                          int out_nchw, int up_h, int up_w, void* stream) {
    if (!in || !weights || !shift || !out) return PMN_ERR_ARG;
    if (N < 1 || H < 1 || W < 1 || cin < 1 || cout < 1 || stride < 1 || dil < 1 || pad < 0) return PMN_ERR_ARG;
    ConvArgs a;
    a.N = N; a.H = H; a.W = W;
    a.Ho = (H + 2 * pad - dil * (K - 1) - 1) / stride + 1;
    a.Wo = (W + 2 * pad - dil * (K - 1) - 1) / stride + 1;
    const int tile = cout <= 8 ? 8 : 16;
    a.COUTP = (cout + tile - 1) / tile * tile;
    a.cout = cout; a.pad = pad; a.dil = dil; a.relu = relu; a.up_h = up_h; a.up_w = up_w;
    if (a.Ho < 1 || a.Wo < 1) return PMN_ERR_ARG;
    if (up && (out_nchw || a.COUTP != cout || up_h * 2 != a.Ho || up_w * 2 != a.Wo)) return PMN_ERR_ARG;
    if (!out_nchw && (cout % 4) != 0) return PMN_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
#define PMN_CONV(CI, T, KK, SS, INN, OUTN) return launch_conv<CI, T, KK, SS, INN, OUTN>(in, weights, shift, up, out, a, st)
    if (in_nchw) {
        if (out_nchw) return PMN_ERR_SHAPE;
        if (cin == 3 && K == 3 && stride == 1 && tile == 8) PMN_CONV(3, 8, 3, 1, true, false);
        if (cin == 1 && K == 3 && stride == 1 && tile == 8) PMN_CONV(1, 8, 3, 1, true, false);
        return PMN_ERR_SHAPE;
    }
#define PMN_TILED(CI, CCH, CO, KK, SS, OUTN) return launch_tiled<CI, CCH, CO, KK, SS, OUTN>(in, weights, shift, up, out, a, st)
    if (out_nchw) {  // offset heads: channels-last feature in, planar [B,2K,h,w] offsets out (cout padded to 16 / 32)
        if (K != 3 || stride != 1) return PMN_ERR_SHAPE;
        // 16 output channels per workgroup (grid.y = coutp/16): the coarsest stage has only ~120 tiles of 16x16 pixels,
        // so the channel split is what fills the chip
#define PMN_HEAD(CI, CP) return launch_tiled_impl<CI, 8, (CP < 16 ? CP : 16), CP, 3, 1, true, false>(in, weights, shift, up, out, a, st)
        if (up) return PMN_ERR_SHAPE;
        if (a.COUTP == 8) {
            if (cin == 8) PMN_HEAD(8, 8);
            if (cin == 64) PMN_HEAD(64, 8);
            if (cin == 32) PMN_HEAD(32, 8);
            if (cin == 16) PMN_HEAD(16, 8);
        } else if (a.COUTP == 16) {
            if (cin == 64) PMN_HEAD(64, 16);
            if (cin == 32) PMN_HEAD(32, 16);
            if (cin == 16) PMN_HEAD(16, 16);
        } else if (a.COUTP == 32) {
            if (cin == 64) PMN_HEAD(64, 32);
            if (cin == 32) PMN_HEAD(32, 32);
            if (cin == 16) PMN_HEAD(16, 32);
        } else if (a.COUTP == 48) {
            if (cin == 64) PMN_HEAD(64, 48);
            if (cin == 32) PMN_HEAD(32, 48);
            if (cin == 16) PMN_HEAD(16, 48);
        }
#undef PMN_HEAD
        return PMN_ERR_SHAPE;
    }
    if (a.COUTP != cout) return PMN_ERR_SHAPE;
    static const int split64 = getenv("PMN_CONV_SPLIT") ? atoi(getenv("PMN_CONV_SPLIT")) : 32;
    static const int cc5 = getenv("PMN_CONV_CC5") ? atoi(getenv("PMN_CONV_CC5")) : 4;
    if (!up && split64 == 32) {
        if (K == 3 && stride == 1 && dil == 1 && cin == 64 && cout == 64)
            return launch_tiled_impl<64, 16, 32, 64, 3, 1, false, false>(in, weights, shift, up, out, a, st);
        if (K == 5 && stride == 2 && cin == 32 && cout == 64) {
            if (cc5 == 4) return launch_tiled_impl<32, 4, 32, 64, 5, 2, false, false>(in, weights, shift, up, out, a, st);
            return launch_tiled_impl<32, 8, 32, 64, 5, 2, false, false>(in, weights, shift, up, out, a, st);
        }
        if (K == 1 && cin == 64 && cout == 64)
            return launch_tiled_impl<64, 16, 32, 64, 1, 1, false, false>(in, weights, shift, up, out, a, st);
    }
    if (!up && cc5 == 4 && K == 5 && stride == 2) {
        if (cin == 8 && cout == 16) return launch_tiled_impl<8, 4, 16, 16, 5, 2, false, false>(in, weights, shift, up, out, a, st);
        if (cin == 16 && cout == 32) return launch_tiled_impl<16, 4, 32, 32, 5, 2, false, false>(in, weights, shift, up, out, a, st);
        if (cin == 32 && cout == 64) return launch_tiled_impl<32, 4, 64, 64, 5, 2, false, false>(in, weights, shift, up, out, a, st);
    }
    if (K == 3 && stride == 1 && dil == 1) {
        if (cin == 8 && cout == 8) PMN_TILED(8, 8, 8, 3, 1, false);
        if (cin == 16 && cout == 8) PMN_TILED(16, 16, 8, 3, 1, false);
        if (cin == 16 && cout == 16) PMN_TILED(16, 16, 16, 3, 1, false);
        if (cin == 32 && cout == 32) PMN_TILED(32, 16, 32, 3, 1, false);
        if (cin == 64 && cout == 64) PMN_TILED(64, 16, 64, 3, 1, false);
    } else if (K == 5 && stride == 2 && dil == 1) {
        if (cin == 8 && cout == 16) PMN_TILED(8, 8, 16, 5, 2, false);
        if (cin == 16 && cout == 32) PMN_TILED(16, 8, 32, 5, 2, false);
        if (cin == 32 && cout == 64) PMN_TILED(32, 8, 64, 5, 2, false);
    } else if (K == 1 && stride == 1) {
        if (cin == 16 && cout == 64) PMN_TILED(16, 16, 64, 1, 1, false);
        if (cin == 32 && cout == 64) PMN_TILED(32, 16, 64, 1, 1, false);
        if (cin == 64 && cout == 64) PMN_TILED(64, 16, 64, 1, 1, false);
        if (cin == 64 && cout == 32) PMN_TILED(64, 16, 32, 1, 1, false);
        if (cin == 64 && cout == 16) PMN_TILED(64, 16, 16, 1, 1, false);
    }
#undef PMN_TILED
#undef PMN_CONV
    return PMN_ERR_SHAPE;
}
