// conv_f16s.hip -- FeatureNet's 16..64-channel ConvBnReLU layers (reference models/net.py:20-31: conv2..conv10, 3x3 stride 1 and
// 5x5 stride 2) on the FP16 matrix cores with SPLIT operands: fp32-convolution accuracy at a multiple of the fp32 MFMA rate.
//
// Why.  gfx950's fp32 MFMA (v_mfma_f32_16x16x4_f32) runs at the fp32 VECTOR rate, 1/16 of v_mfma_f32_16x16x32_f16, and there is
// no xf32 form.  Every value is split on the fly into two fp16 numbers,
//       x = hi + lo / 2048,   hi = fp16(x),   lo = fp16((x - hi) * 2048)            (22 significant bits for 6.1e-5 <= |x| < 65504;
//       below that hi is a subnormal and the ABSOLUTE error stays <= 3e-8 per factor; at |x| >= 65504 hi = inf and the result is
//       inf / NaN -- the kernel does not check: weights are checked when packed, activations are the caller's contract, see the
//       "fp16-split entry points" paragraph of include/pmn_hip.h)
// and the convolution is evaluated as
//       sum x*w  ~=  sum hi_x*hi_w  +  ( sum hi_x*lo_w + sum lo_x*hi_w ) / 2048        (the lo*lo term, 2^-22, is dropped)
// Products of two fp16 numbers are exact in fp32 and the MFMA accumulates in fp32, so the result carries 2-4e-7 of the output scale
// -- what an fp32 direct convolution of the same layer carries (scripts/fp16_split_study.py on the checkpoint's own layers and
// activations; tests/test_f16s_emulation.py).  Three fp16 MFMAs per k-step instead of one fp32 MFMA over a K 8 times shorter:
// 16/3 = 5.3x the fp32 matrix rate, which turns these layers from matrix-pipe-bound into HBM-bound.
//
// Mapping.  Implicit GEMM, rows = output pixels, columns = output channels, k = (tap, input channel).  Workgroup = 4 waves =
// a 16 x (4 MT) block of output pixels; wave w owns output rows [w MT, (w+1) MT) -- MT M-tiles of 16 consecutive pixels of one row --
// and ALL cout/16 N-tiles.  Per chunk of CC input channels the input patch ((4MT-1) S + K rows x (15 S + K) columns) is loaded
// once (coalesced float4), split, and kept in LDS as two fp16 planes [row][col][CCP] (CCP = CC + padding, chosen so that the
// ds_read_b128 lane groups {0-3,12-15,20-27}.. touch 16 distinct 16-byte slots; brute-forced per shape).  The k axis of a chunk is
// cut into blocks of 8 channels, block q = tap * (CC/8) + cb; a k-step = 4 blocks = one v_mfma_f32_16x16x32_f16:
//   A  lane (i = lane&15, kb = lane>>4): ONE ds_read_b128 = channels [8 cb, +8) of pixel (row + dy, 16-tile column i*S + dx) for
//      block q = 4 ks + kb -- hi plane and lo plane
//   B  host-packed to [chunk][k-step][N-tile][hi|lo][64 lanes][8] fp16 (params.pack_conv_f16s): one contiguous 1 KB load per wave,
//      L2-resident, one k-step ahead in registers
//   D  lane holds column n = lane&15 (output channel) of rows 4 (lane>>4) + r (pixels of the tile): two accumulators per tile,
//      main (hi*hi) and low (hi*lo + lo*hi); epilogue: main + low/2048 + shift (folded BatchNorm), ReLU, channels-last store.
#include "pmn_common.hpp"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

struct F16sArgs {
    int N, H, W, Ho, Wo, relu;
    int cout, ca;   // OUTMODE 1 / 2: valid output channels (the operands are padded to COUT = a multiple of 16), channels of `out`
    float* out_b;   // OUTMODE 1 / 2: channels [ca, cout) go here
};

#define PMN_F16S_LO_SCALE 2048.0f

// DIL = dilation (padding DIL * (KS / 2)); OUTMODE 0: out [N,Ho,Wo,COUT] channels-last; 1: planar, split: out [N,ca,Ho,Wo] | out_b
// [N,cout-ca,Ho,Wo] (the offset heads of a PatchMatch stage: propa_conv rows, then eval_conv rows); 2: channels-last, split.
template <int CIN, int COUT, int KS, int S, int CC, int CCP, int MT, int WPS, int DIL = 1, int OUTMODE = 0, bool SWAP = (OUTMODE == 0)>
__global__ __launch_bounds__(256, WPS) void conv_f16s_kernel(const float* __restrict__ in, const f16x8* __restrict__ wB,
                                                             const float* __restrict__ shift, float* __restrict__ out,
                                                             const F16sArgs a) {
    constexpr int NT = COUT / 16, NCB = CC / 8, CHUNKS = CIN / CC, NQ = KS * KS * NCB, KSTEPS = (NQ + 3) / 4;
    constexpr int TH = 4 * MT, TW = 16, PH = (TH - 1) * S + (KS - 1) * DIL + 1, PW = (TW - 1) * S + (KS - 1) * DIL + 1;
    constexpr int PAD = DIL * (KS / 2);
    constexpr int PLANE = PH * PW * CCP;  // halves per plane
    constexpr int NTHR = 256;
    static_assert(!SWAP || OUTMODE == 0, "swapped operand roles (see the epilogue) serve the channels-last output");
    static_assert(CCP % 8 == 0 && CC % 8 == 0 && CIN % CC == 0, "16-byte aligned channel blocks");
    extern __shared__ float4 f16s_lds4[];
    _Float16* Phi = reinterpret_cast<_Float16*>(f16s_lds4);
    _Float16* Plo = Phi + PLANE;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 15, kb = lane >> 4;
    const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
    const int bt = pmn_xcd_tile(blockIdx.x, a.N * tiles_x * tiles_y);
    const int n = bt / (tiles_x * tiles_y), tr = bt - n * tiles_x * tiles_y;
    const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TW;
    const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;

    f32x4_t accM[MT][NT], accL[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            accM[t][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            accL[t][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }

    const f16x8* bl = wB + lane;  // element (((ch * KSTEPS + ks) * NT + nt) * 2 + split) * 64 + lane
    f16x8 bq[2][NT][2];           // B operands of two consecutive k-steps (compile-time indexed: the k loop is fully unrolled)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        bq[0][nt][0] = bl[(size_t)((0 * NT + nt) * 2 + 0) * 64];
        bq[0][nt][1] = bl[(size_t)((0 * NT + nt) * 2 + 1) * 64];
    }

    // staging position of the thread's first unit (see the patch block): pixel tid / QP of the patch, channel quad tid % QP
    const int q4x4 = 4 * (tid % (CC / 4)), pix_first = tid / (CC / 4);
    const int py_first = pix_first / PW, px_first = pix_first - py_first * PW, lds_first = pix_first * CCP + q4x4;

#pragma unroll 1
    for (int ch = 0; ch < CHUNKS; ++ch) {
        if (ch) __syncthreads();  // every wave is done with the previous chunk's patch
        {   // ---- patch: PH x PW pixels x CC/4 channel quads -> split -> two fp16 planes; SB loads of a thread in flight per batch.
            // Written for few instructions (these layers issue about as many VALU cycles in this block as MFMA cycles in the k loop):
            // unit idx = tid + k NTHR is pixel pix0 + k DPIX, quad q4 -- the pixel's (row, column) advances by a constant step with
            // one wrap instead of a division per load, the LDS slot is a compile-time offset from the thread's first one, the global
            // address is a uniform base + a 32-bit offset (host: H W CIN < 2^30), the split runs on 2-vectors.
            constexpr int QP = CC / 4, TOT = PH * PW * QP, NL = (TOT + NTHR - 1) / NTHR, SB = 6;
            constexpr int DPIX = NTHR / QP, DY = DPIX / PW, DX = DPIX - DY * PW;
            static_assert(NTHR % QP == 0, "a step of NTHR units is a whole number of pixels");
            const float* src = in + (size_t)n * a.H * a.W * CIN + ch * CC;  // uniform
            int py = py_first, px = px_first;
#pragma unroll
            for (int k0 = 0; k0 < NL; k0 += SB) {
                float4 v[SB];
#pragma unroll
                for (int k = 0; k < SB; ++k) {
                    const int gy = iy0 + py, gx = ix0 + px;
                    v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (k0 + k < NL && tid < TOT - (k0 + k) * NTHR && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
                        v[k] = *reinterpret_cast<const float4*>(src + ((unsigned)(gy * a.W + gx) * CIN + q4x4));
                    px += DX;
                    py += DY;
                    if (px >= PW) {
                        px -= PW;
                        ++py;
                    }
                }
#pragma unroll
                for (int k = 0; k < SB; ++k) {
                    if (k0 + k < NL && tid < TOT - (k0 + k) * NTHR) {
                        const f32x2_t x01 = {v[k].x, v[k].y}, x23 = {v[k].z, v[k].w};
                        const f16x2_t h01 = __builtin_convertvector(x01, f16x2_t), h23 = __builtin_convertvector(x23, f16x2_t);  // RNE
                        // x - hi is exact in fp32
                        const f16x2_t l01 = __builtin_convertvector((x01 - __builtin_convertvector(h01, f32x2_t)) * PMN_F16S_LO_SCALE, f16x2_t);
                        const f16x2_t l23 = __builtin_convertvector((x23 - __builtin_convertvector(h23, f32x2_t)) * PMN_F16S_LO_SCALE, f16x2_t);
                        const f16x4 hi = {h01[0], h01[1], h23[0], h23[1]}, lo = {l01[0], l01[1], l23[0], l23[1]};
                        *reinterpret_cast<f16x4*>(Phi + lds_first + (k0 + k) * DPIX * CCP) = hi;
                        *reinterpret_cast<f16x4*>(Plo + lds_first + (k0 + k) * DPIX * CCP) = lo;
                    }
                }
            }
        }
        __syncthreads();

        // ---- k loop of the chunk (fully unrolled; the fences keep hipcc from hoisting every B load to the top) ---------------
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            constexpr int dummy = 0;
            (void)dummy;
            const int cur = ks & 1, nxt = cur ^ 1;
            // next k-step's B operands (runs on into the next chunk; the very last step re-reads itself)
            {
                const int step = ch * KSTEPS + ks + 1;
                const int lim = CHUNKS * KSTEPS - 1;
                const int sidx = step < lim ? step : lim;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    bq[nxt][nt][0] = bl[(size_t)((sidx * NT + nt) * 2 + 0) * 64];
                    bq[nxt][nt][1] = bl[(size_t)((sidx * NT + nt) * 2 + 1) * 64];
                }
            }
            // this lane's k-block: q = 4 ks + kb -> (tap, channel block); padding blocks (zero weights) read block NQ - 1
            int q = 4 * ks + kb;
            if (4 * ks + 3 >= NQ) q = q < NQ - 1 ? q : NQ - 1;
            const int tap = q / NCB, cb = q - tap * NCB;
            const int dy = tap / KS, dx = tap - dy * KS;
            const _Float16* pa = Phi + ((wave * MT * S + dy * DIL) * PW + li * S + dx * DIL) * CCP + cb * 8;
            f16x8 ah[MT], al[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                ah[t] = *reinterpret_cast<const f16x8*>(pa + t * S * PW * CCP);
                al[t] = *reinterpret_cast<const f16x8*>(pa + t * S * PW * CCP + PLANE);
            }
            // three passes, every accumulator once per pass: no back-to-back dependent MFMAs
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    accM[t][nt] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_f16(bq[cur][nt][0], ah[t], accM[t][nt], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bq[cur][nt][0], accM[t][nt], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    accL[t][nt] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_f16(bq[cur][nt][1], ah[t], accL[t][nt], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bq[cur][nt][1], accL[t][nt], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    accL[t][nt] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_f16(bq[cur][nt][0], al[t], accL[t][nt], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bq[cur][nt][0], accL[t][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KSTEPS & 1) {  // an odd number of k-steps leaves the prefetched operands in slot 1: the next chunk starts from slot 0
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bq[0][nt][0] = bq[1][nt][0];
                bq[0][nt][1] = bq[1][nt][1];
            }
        }
    }

    // ---- epilogue: main + low / 2048 + shift (folded BatchNorm / bias), ReLU.
    if constexpr (SWAP) {
        // channels-last output: the MFMAs ran with the operand roles swapped (rows = output channels, columns = pixels), so lane
        // (li, kb) holds channels 16 nt + 4 kb .. + 3 of pixel ox0 + li -- one 16-byte store per tile, 64 contiguous bytes per pixel
        // and N-tile (with the roles as staged it is four 4-byte stores per tile and their address arithmetic)
        const int ox = ox0 + li, oyw = oy0 + wave * MT;
        if (ox < a.Wo) {
            float* po = out + (((size_t)n * a.Ho + oyw) * a.Wo + ox) * COUT + 4 * kb;
            const size_t rs = (size_t)a.Wo * COUT;
            f32x4_t sh[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) sh[nt] = *reinterpret_cast<const f32x4_t*>(shift + 16 * nt + 4 * kb);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                if (oyw + t < a.Ho) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        f32x4_t v = accM[t][nt] + accL[t][nt] * (1.0f / PMN_F16S_LO_SCALE) + sh[nt];
                        if (a.relu) v = __builtin_elementwise_max(v, f32x4_t{0.f, 0.f, 0.f, 0.f});
                        *reinterpret_cast<f32x4_t*>(po + t * rs + 16 * nt) = v;
                    }
                }
            }
        }
        return;
    }
    // planar / split outputs (the offset heads): lane = output channel 16 nt + li, rows 4 kb + r = pixels
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int c = nt * 16 + li;
        const float sh = shift[c];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int oy = oy0 + wave * MT + t;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = accM[t][nt][r] + accL[t][nt][r] * (1.0f / PMN_F16S_LO_SCALE) + sh;
                if (a.relu) v[r] = fmaxf(v[r], 0.0f);
            }
            if constexpr (OUTMODE == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ox = ox0 + 4 * kb + r;
                    if (oy < a.Ho && ox < a.Wo) out[(((size_t)n * a.Ho + oy) * a.Wo + ox) * COUT + c] = v[r];
                }
            } else {
                if (c < a.cout && oy < a.Ho) {
                    const bool first = c < a.ca;
                    float* dst = first ? out : a.out_b;
                    const int cx = first ? c : c - a.ca, nc = first ? a.ca : a.cout - a.ca;
                    if constexpr (OUTMODE == 1) {  // planar: the lane's four pixels are consecutive in x
                        float* row = dst + (((size_t)n * nc + cx) * a.Ho + oy) * a.Wo;
                        const int ox = ox0 + 4 * kb;
                        if ((a.Wo & 3) == 0 && ox + 3 < a.Wo) {
                            *reinterpret_cast<float4*>(row + ox) = make_float4(v[0], v[1], v[2], v[3]);
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (ox + r < a.Wo) row[ox + r] = v[r];
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int ox = ox0 + 4 * kb + r;
                            if (ox < a.Wo) dst[(((size_t)n * a.Ho + oy) * a.Wo + ox) * nc + cx] = v[r];
                        }
                    }
                }
            }
        }
    }
}

template <int CIN, int COUT, int KS, int S, int CC, int CCP, int MT, int WPS, int DIL = 1, int OUTMODE = 0, bool SWAP = (OUTMODE == 0)>
static int launch_f16s(const float* in, const void* w, const float* shift, float* out, F16sArgs a, hipStream_t st) {
    constexpr int TH = 4 * MT, PH = (TH - 1) * S + (KS - 1) * DIL + 1, PW = 15 * S + (KS - 1) * DIL + 1;
    const size_t lds = (size_t)2 * PH * PW * CCP * sizeof(_Float16);
    auto kern = conv_f16s_kernel<CIN, COUT, KS, S, CC, CCP, MT, WPS, DIL, OUTMODE, SWAP>;
    if ((size_t)a.H * a.W * CIN >= ((size_t)1 << 30)) return PMN_ERR_SHAPE;  // the kernel addresses one image with a 32-bit offset
    if (lds > 48 * 1024 && pmn_raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != PMN_OK) return PMN_ERR_LAUNCH;
    const int blocks = a.N * ((a.Wo + 15) / 16) * ((a.Ho + TH - 1) / TH);
    PMN_LAUNCH(kern, dim3(blocks), dim3(256), lds, st, in, reinterpret_cast<const f16x8*>(w), shift, out, a);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

// in [N,H,W,cin] channels-last float32; weights DEVICE fp16 [cin/CC][k-steps][cout/16][2][64][8] (params.pack_conv_f16s); shift
// DEVICE float[cout]; out [N,Ho,Wo,cout] float32 with Ho = (H-1)/stride + 1 (padding k/2).  Supported (k, stride, cin, cout):
// (3,1,16,16), (3,1,32,32), (3,1,64,64), (5,2,8,16), (5,2,16,32), (5,2,32,64) -- FeatureNet's conv2..conv10.
extern "C" int pmn_conv2d_f16s(const float* in, const void* weights, const float* shift, float* out, int N, int H, int W, int cin,
                               int cout, int k, int stride, int relu, void* stream) {
    if (!in || !weights || !shift || !out || N < 1 || H < 1 || W < 1) return PMN_ERR_ARG;
    F16sArgs a;
    a.N = N; a.H = H; a.W = W; a.relu = relu;
    a.cout = cout; a.ca = cout; a.out_b = nullptr;
    a.Ho = (H - 1) / (stride > 0 ? stride : 1) + 1;
    a.Wo = (W - 1) / (stride > 0 ? stride : 1) + 1;
    hipStream_t st = (hipStream_t)stream;
    // Tile / chunk shapes from same-box A/Bs of build variants (profiles/r03_f16s_ab.log): small chunks (more workgroups per CU) and
    // M = 32 pixels per wave (MT = 2) win on the small layers; M = 16 loses (the B operands are re-read per wave), and so does a
    // persistent variant that prefetches the next (tile, chunk) behind the MFMAs with the B operands staged in LDS
    // (scripts/experiments/conv_f16s_persistent_prefetch.hip.txt: correct, 5-10 % slower) -- these layers are bound by operand delivery
    // (8 KB of B operands per 384 MFMA cycles and wave = 85 B/clk/CU against the L1's 64), not by the latency of the patch loads.
    // Operand roles (last template argument; scripts/experiments/f16s_ab/, same box, bit-identical either way): swapped (one 16-byte
    // store per tile) wins for one and four N-tiles -- 16->16 96 -> 83 us, 64->64 58 -> 53, 8->16 138 -> 131, 32->64 80 -> 75 per six
    // views -- and loses for two (32->32 55.6 as staged vs 61.2 swapped, 16->32 88.5 vs 91.1), which therefore keep the staged roles.
    //                                                            CIN COUT KS S  CC CCP MT WPS
    if (k == 3 && stride == 1 && cin == 16 && cout == 16) return launch_f16s<16, 16, 3, 1, 16, 16, 4, 4>(in, weights, shift, out, a, st);
    if (k == 3 && stride == 1 && cin == 32 && cout == 32) return launch_f16s<32, 32, 3, 1, 16, 16, 2, 4, 1, 0, false>(in, weights, shift, out, a, st);
    if (k == 3 && stride == 1 && cin == 64 && cout == 64) return launch_f16s<64, 64, 3, 1, 16, 16, 2, 3>(in, weights, shift, out, a, st);
    if (k == 5 && stride == 2 && cin == 8 && cout == 16) return launch_f16s<8, 16, 5, 2, 8, 8, 2, 4>(in, weights, shift, out, a, st);
    if (k == 5 && stride == 2 && cin == 16 && cout == 32) return launch_f16s<16, 32, 5, 2, 8, 8, 2, 4, 1, 0, false>(in, weights, shift, out, a, st);
    if (k == 5 && stride == 2 && cin == 32 && cout == 64) return launch_f16s<32, 64, 5, 2, 16, 24, 2, 2>(in, weights, shift, out, a, st);
    return PMN_ERR_SHAPE;
}


// =================================================================================================================================
// TWO consecutive 3x3 / stride-1 / 16 -> 16 ConvBnReLU layers in one launch (FeatureNet conv3 + conv4 at half resolution, reference
// models/net.py:21-22, 52) -- round 6.
//
// Why these two.  Ablation builds of the kernel above (stores predicated off / patch loads from an L2-resident window:
// profiles/r06_conv_fusion_bound.log) price what the intermediate map's trip through HBM costs per layer pair: 8 us for conv6+conv7,
// 9 us for conv9+conv10 (those layers are bound by operand delivery, as their tile A/Bs said) -- and 60-70 us for conv3+conv4, which
// move 184 MB out and back in at half resolution and are HBM-bound.  So the pair that is fused is this one, not the ones the reviews
// of rounds 4 and 5 asked for.
//
// Mapping.  Workgroup = 14 x 14 output pixels.  (1) the 18 x 18 input patch is staged as hi / lo fp16 planes exactly as above.
// (2) layer A on the 16 x 16 region the second layer reads (one 16-pixel M-tile per row, four rows per wave), operand roles swapped
// as in the single-layer kernel: same k order, same MFMA sequence per accumulator, same epilogue expression, so the region holds the
// BITS pmn_conv2d_f16s would have written; positions outside the image are ZERO (layer B pads layer A's output, not its input).
// (3) after a barrier the region is split into hi / lo planes IN PLACE of the input patch (same conversion as the staging code).
// (4) layer B on the 14 x 14 tile from those planes, epilogue and channels-last stores as above.  Halo recompute: layer A does
// 256 / 196 = 1.31x the MFMAs of the unfused layer and layer B runs 16-wide tiles for 14 valid columns; the matrix pipe has that slack
// (these layers ran at 20 % of it).  HBM: 1.65x the input once + 1x the output instead of 2 x (1.27 + 1): 498 instead of 835 MB per six
// 1600x1200 views.  Measured: 160 us instead of 92 + 91 (same box, scripts/call_ab.py; 4 / 5 / 6 / 8 waves per SIMD: 160 / 163 / 162 /
// 189) -- the HBM time alone would be ~100 us: a workgroup's four phases run one after the other behind three barriers and four to
// six workgroups per CU do not hide that; a wider tile or a producer / consumer wave split is what is left to try.
template <int WPS>
__global__ __launch_bounds__(256, WPS) void conv_f16s_pair16_kernel(const float* __restrict__ in, const f16x8* __restrict__ wA,
                                                                   const float* __restrict__ shiftA, const f16x8* __restrict__ wBq,
                                                                   const float* __restrict__ shiftB, float* __restrict__ out,
                                                                   const F16sArgs a) {
    constexpr int C = 16, CCP = 16, NCB = 2, NQ = 18, KSTEPS = 5, TO = 14, RA = 16, PI = 18;
    constexpr int PLANE = PI * PI * CCP;  // halves per plane (layer A's 16 x 16 region re-uses the same two planes at pitch RA)
    extern __shared__ float4 f16s_pair_lds4[];
    _Float16* Phi = reinterpret_cast<_Float16*>(f16s_pair_lds4);
    _Float16* Plo = Phi + PLANE;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 15, kb = lane >> 4;
    const int tiles_x = (a.W + TO - 1) / TO, tiles_y = (a.H + TO - 1) / TO;
    const int bt = pmn_xcd_tile(blockIdx.x, a.N * tiles_x * tiles_y);
    const int n = bt / (tiles_x * tiles_y), tr = bt - n * tiles_x * tiles_y;
    const int oy0 = (tr / tiles_x) * TO, ox0 = (tr % tiles_x) * TO;

    // ---- (1) input patch, origin (oy0 - 2, ox0 - 2): 324 pixels x 4 channel quads, all of a thread's loads in flight before the split
    {
        constexpr int QP = C / 4, TOT = PI * PI * QP, NL = (TOT + 255) / 256;
        const float* src = in + (size_t)n * a.H * a.W * C;
        float4 v[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int idx = tid + k * 256, pix = idx / QP, q = idx - pix * QP;
            const int py = pix / PI, px = pix - py * PI, gy = oy0 - 2 + py, gx = ox0 - 2 + px;
            v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < TOT && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
                v[k] = *reinterpret_cast<const float4*>(src + ((unsigned)(gy * a.W + gx) * C + 4 * q));
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int idx = tid + k * 256, pix = idx / QP, q = idx - pix * QP;
            if (idx < TOT) {
                const f32x2_t x01 = {v[k].x, v[k].y}, x23 = {v[k].z, v[k].w};
                const f16x2_t h01 = __builtin_convertvector(x01, f16x2_t), h23 = __builtin_convertvector(x23, f16x2_t);  // RNE
                const f16x2_t l01 = __builtin_convertvector((x01 - __builtin_convertvector(h01, f32x2_t)) * PMN_F16S_LO_SCALE, f16x2_t);
                const f16x2_t l23 = __builtin_convertvector((x23 - __builtin_convertvector(h23, f32x2_t)) * PMN_F16S_LO_SCALE, f16x2_t);
                const f16x4 hi = {h01[0], h01[1], h23[0], h23[1]}, lo = {l01[0], l01[1], l23[0], l23[1]};
                *reinterpret_cast<f16x4*>(Phi + pix * CCP + 4 * q) = hi;
                *reinterpret_cast<f16x4*>(Plo + pix * CCP + 4 * q) = lo;
            }
        }
    }
    __syncthreads();

    // one 3x3 / 16 -> 16 layer over four rows of 16 pixels per wave from planes of row pitch `pitch` pixels: the k loop of
    // conv_f16s_kernel<16, 16, 3, 1, 16, 16, 4, ..., SWAP = true> (B operands one k-step ahead in registers)
    f32x4_t accM[4], accL[4];
    auto layer = [&](const f16x8* __restrict__ wq, const int pitch) {
        const f16x8* bl = wq + lane;
        f16x8 bq[2][2];
        bq[0][0] = bl[0];
        bq[0][1] = bl[64];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            accM[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            accL[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            const int sidx = ks + 1 < KSTEPS ? ks + 1 : KSTEPS - 1;
            bq[nxt][0] = bl[(size_t)(sidx * 2 + 0) * 64];
            bq[nxt][1] = bl[(size_t)(sidx * 2 + 1) * 64];
            int q = 4 * ks + kb;
            if (4 * ks + 3 >= NQ) q = q < NQ - 1 ? q : NQ - 1;
            const int tap = q / NCB, cb = q - tap * NCB;
            const int dy = tap / 3, dx = tap - dy * 3;
            const _Float16* pa = Phi + ((wave * 4 + dy) * pitch + li + dx) * CCP + cb * 8;
            f16x8 ah[4], al[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                ah[t] = *reinterpret_cast<const f16x8*>(pa + t * pitch * CCP);
                al[t] = *reinterpret_cast<const f16x8*>(pa + t * pitch * CCP + PLANE);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) accM[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bq[cur][0], ah[t], accM[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) accL[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bq[cur][1], ah[t], accL[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) accL[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bq[cur][0], al[t], accL[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- (2) layer A on the 16 x 16 region, origin (oy0 - 1, ox0 - 1): lane (li, kb) ends up with channels 4 kb .. + 3 of column li
    layer(wA, PI);
    f16x4 rh[4], rl[4];
    {
        const f32x4_t sh = *reinterpret_cast<const f32x4_t*>(shiftA + 4 * kb);
        const int gx = ox0 - 1 + li;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int gy = oy0 - 1 + wave * 4 + t;
            f32x4_t v = accM[t] + accL[t] * (1.0f / PMN_F16S_LO_SCALE) + sh;
            if (a.relu) v = __builtin_elementwise_max(v, f32x4_t{0.f, 0.f, 0.f, 0.f});
            if (!((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)) v = f32x4_t{0.f, 0.f, 0.f, 0.f};  // layer B's zero padding
            const f32x2_t x01 = {v[0], v[1]}, x23 = {v[2], v[3]};
            const f16x2_t h01 = __builtin_convertvector(x01, f16x2_t), h23 = __builtin_convertvector(x23, f16x2_t);
            const f16x2_t l01 = __builtin_convertvector((x01 - __builtin_convertvector(h01, f32x2_t)) * PMN_F16S_LO_SCALE, f16x2_t);
            const f16x2_t l23 = __builtin_convertvector((x23 - __builtin_convertvector(h23, f32x2_t)) * PMN_F16S_LO_SCALE, f16x2_t);
            rh[t] = f16x4{h01[0], h01[1], h23[0], h23[1]};
            rl[t] = f16x4{l01[0], l01[1], l23[0], l23[1]};
        }
    }
    __syncthreads();  // every wave is done reading the input patch: the region takes its place
    // ---- (3) the region as hi / lo planes [16 rows][16 px][16 halves]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int pix = (wave * 4 + t) * RA + li;
        *reinterpret_cast<f16x4*>(Phi + pix * CCP + 4 * kb) = rh[t];
        *reinterpret_cast<f16x4*>(Plo + pix * CCP + 4 * kb) = rl[t];
    }
    __syncthreads();

    // ---- (4) layer B on the 14 x 14 tile (rows / columns 14, 15 of the 16-wide tiles read past the region: computed, never stored)
    layer(wBq, RA);
    {
        const int ox = ox0 + li;
        if (li < TO && ox < a.W) {
            const f32x4_t sh = *reinterpret_cast<const f32x4_t*>(shiftB + 4 * kb);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int r = wave * 4 + t, oy = oy0 + r;
                if (r < TO && oy < a.H) {
                    f32x4_t v = accM[t] + accL[t] * (1.0f / PMN_F16S_LO_SCALE) + sh;
                    if (a.relu) v = __builtin_elementwise_max(v, f32x4_t{0.f, 0.f, 0.f, 0.f});
                    *reinterpret_cast<f32x4_t*>(out + (((size_t)n * a.H + oy) * a.W + ox) * C + 4 * kb) = v;
                }
            }
        }
    }
}

// in [N,H,W,16] channels-last float32 -> out [N,H,W,16] = relu(convB(relu(convA(in)) ...)): weights_a / weights_b, shift_a / shift_b as
// pmn_conv2d_f16s takes them for a (3, 1, 16, 16) layer (params.pack_conv_f16s).  Bit-identical to two pmn_conv2d_f16s calls.
extern "C" int pmn_conv2d_f16s_pair(const float* in, const void* weights_a, const float* shift_a, const void* weights_b,
                                    const float* shift_b, float* out, int N, int H, int W, int channels, int relu, void* stream) {
    if (!in || !weights_a || !shift_a || !weights_b || !shift_b || !out || N < 1 || H < 1 || W < 1) return PMN_ERR_ARG;
    if (channels != 16) return PMN_ERR_SHAPE;
    if ((size_t)H * W * 16 >= ((size_t)1 << 30)) return PMN_ERR_SHAPE;
    F16sArgs a;
    a.N = N; a.H = H; a.W = W; a.Ho = H; a.Wo = W; a.relu = relu;
    a.cout = 16; a.ca = 16; a.out_b = nullptr;
    const size_t lds = (size_t)2 * 18 * 18 * 16 * sizeof(_Float16);
    const int blocks = N * ((W + 13) / 14) * ((H + 13) / 14);
    PMN_LAUNCH(conv_f16s_pair16_kernel<4>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, in, reinterpret_cast<const f16x8*>(weights_a),
               shift_a, reinterpret_cast<const f16x8*>(weights_b), shift_b, out, a);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}


// The offset heads of one PatchMatch stage (propa_conv rows first, then eval_conv: reference models/patchmatch.py:288-311) as ONE dilated
// 3x3 convolution with bias on the fp16 matrix cores (split operands), planar outputs.  in [N,H,W,cin] channels-last; weights DEVICE
// fp16 [cin/16][k-steps][coutp/16][2][64][8] (params.pack_conv_f16s_general, rows padded with zeros to coutp = a multiple of 16); shift
// DEVICE float[coutp]; out_a [N,ca,H,W], out_b [N,cout-ca,H,W] (may be NULL when ca == cout).  Supported: (cin, dilation) in
// {(64,2), (32,4), (16,6)} = the reference's stages 3, 2, 1 with coutp in {32, 48, 64}; everything else: PMN_ERR_SHAPE (the caller
// falls back to pmn_conv2d_mfma / pmn_conv2d).
template <int CIN, int DIL, int MT, int WPS>
static int dispatch_heads(const float* in, const void* w, const float* shift, float* out_a, int coutp, F16sArgs a, hipStream_t st) {
    if (coutp == 32) return launch_f16s<CIN, 32, 3, 1, 16, 16, MT, WPS, DIL, 1>(in, w, shift, out_a, a, st);
    if (coutp == 48) return launch_f16s<CIN, 48, 3, 1, 16, 16, MT, WPS, DIL, 1>(in, w, shift, out_a, a, st);
    if (coutp == 64) return launch_f16s<CIN, 64, 3, 1, 16, 16, MT, (WPS > 3 ? 3 : WPS), DIL, 1>(in, w, shift, out_a, a, st);
    return PMN_ERR_SHAPE;
}

extern "C" int pmn_offset_heads_f16s(const float* in, const void* weights, const float* shift, float* out_a, float* out_b, int N, int H,
                                     int W, int cin, int cout, int ca, int dil, void* stream) {
    if (!in || !weights || !shift || !out_a || N < 1 || H < 1 || W < 1 || cout < 1 || ca < 1 || ca > cout) return PMN_ERR_ARG;
    if (ca < cout && !out_b) return PMN_ERR_ARG;
    F16sArgs a;
    a.N = N; a.H = H; a.W = W; a.Ho = H; a.Wo = W; a.relu = 0;
    a.cout = cout; a.ca = ca; a.out_b = out_b;
    const int coutp = (cout + 15) / 16 * 16;
    hipStream_t st = (hipStream_t)stream;
    if (cin == 64 && dil == 2) return dispatch_heads<64, 2, 2, 3>(in, weights, shift, out_a, coutp, a, st);
    if (cin == 32 && dil == 4) return dispatch_heads<32, 4, 2, 4>(in, weights, shift, out_a, coutp, a, st);
    if (cin == 16 && dil == 6) return dispatch_heads<16, 6, 2, 4>(in, weights, shift, out_a, coutp, a, st);
    return PMN_ERR_SHAPE;
}

// =================================================================================================================================
// Fused stem on the fp16 matrix cores: conv0 (3 -> 8, fp32 VALU, as pmn_stem) feeds conv1 (8 -> 8: 72 % of the stem's multiplies)
// as split-operand MFMAs.  reference models/net.py:17-19, 51.
//
// Workgroup = 16 x 16 output pixels.  (1) input patch -> LDS as 3 x 20 rows of 24 floats, x = ox0 - 4 .. ox0 + 19: with W % 4 == 0
// (VEC4; every size the network itself produces) a row is six ALIGNED float4 loads, each entirely inside or outside the image, and
// the index arithmetic is shifts (the scalar staging it replaces spent 165 VALU instructions per thread on / and %); (2) every thread
// evaluates conv0 + BatchNorm + ReLU for its pixel of the 18 x 18 halo patch (zero outside the image: conv1 pads conv0's OUTPUT map),
// splits the 8 channels into hi / lo fp16 and writes two planes [18][18][8 halves] (16 B per pixel: a ds_read_b128 lane group {pixels
// i} x {tap q, tap q + 1} touches 16 distinct slots or the same address -- brute-forced); (3) conv1 with the ROLES SWAPPED: the MFMA's
// A operand (rows) = the 8 output channels (rows 8..15 zero), the B operand (columns) = 16 consecutive pixels of one output row,
// k = (tap, channel): 9 blocks of 8 channels = 3 k-steps (3 padding blocks).  D then holds, in lane (pixel i, kb), output channels
// 4 kb .. 4 kb + 3 of pixel i for kb = 0, 1: one float4 store per lane, 512 contiguous bytes per 16-pixel row of the channels-last
// output.  Wave w owns output rows 4 w .. 4 w + 3.
//
// What it waits for (scripts/experiments/README.md, phase ablation): the phases' costs add up -- it is bound by instruction issue, not
// by HBM (84 MB per 1600 x 1200 view = 10.5 us) -- so the kernel is written for few instructions: 404 -> 235 static VALU instructions
// against the first version (vector staging, halo pixels outside the image skip conv0, split and epilogue on 2- / 4-vectors, one
// output pointer per lane), 44 -> 38 us per view with bit-identical results.  Tried and measured equal: a 14 x 14 tile (halo patch =
// one pass of 256 threads), alternating the second conv0 pass between wave pairs, 8 instead of 6 workgroups per CU.
// =================================================================================================================================
#ifndef PMN_STEM_TH
#define PMN_STEM_TH 32  // output rows per workgroup tile (16 x TH pixels): 32 amortises the per-tile skeleton (round 5); 16 = rounds 3-4
#endif
template <bool VEC4, int TH>
__global__ __launch_bounds__(256, 4) void stem_f16s_kernel(const float* __restrict__ img, const float* __restrict__ w0,
                                                          const float* __restrict__ s0, const f16x8* __restrict__ w1A,
                                                          const float* __restrict__ s1, float* __restrict__ out, const int N,
                                                          const int H, const int W, const float* const* __restrict__ img_tab,
                                                          const int BV) {
    constexpr int TS = 16, IW = TH + 4, XS = 24, MW = TS + 2, MRP = 18, MROWS = TH + 2, NTHR = 256;  // IW / MROWS: patch / halo ROWS
    static_assert(TH % 16 == 0, "a wave owns TH / 4 rows, four at a time");
    __shared__ float4 xin4[3 * IW * (XS / 4)];
    __shared__ float4 mid4[2 * MROWS * MRP];  // two planes of 18 x 18 pixel slots x 8 halves (16 B)
    float* xin = reinterpret_cast<float*>(xin4);
    _Float16* midh = reinterpret_cast<_Float16*>(mid4);
    _Float16* midl = midh + MROWS * MRP * 8;
    typedef const float __attribute__((address_space(4))) cfloat;
    const cfloat* cw0 = (const cfloat*)w0;  // [3][3][3][8]
    const cfloat* cs0 = (const cfloat*)s0;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 15, kb = lane >> 4;
    const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + TH - 1) / TH;
    const int bt = pmn_xcd_tile(blockIdx.x, N * tiles_x * tiles_y);
    const int n = bt / (tiles_x * tiles_y), tr = bt - n * tiles_x * tiles_y;
    const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TS;
    // image n of the launch: a slice of one [N,3,H,W] tensor, or -- pmn_stem_f16s_views -- batch element n % BV of the view n / BV
    // whose [BV,3,H,W] tensor sits wherever the device table says (wave-uniform: one scalar load)
    if (img_tab) img = img_tab[n / BV] + (ptrdiff_t)((n % BV) - n) * 3 * H * W;  // (the indexing below adds n * 3 * H * W)

    // conv1's weights (A operands of the three k-steps, hi | lo): six 1 KB loads per wave, in flight across the staging below
    f16x8 wa[3][2];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        wa[ks][0] = w1A[(ks * 2 + 0) * 64 + lane];
        wa[ks][1] = w1A[(ks * 2 + 1) * 64 + lane];
    }
    // (1) input patch, every load of the thread in flight before the first LDS write.  Row R of 60 = (channel, patch row).
    if constexpr (VEC4) {
        constexpr int NU = (3 * IW + 31) / 32;  // passes of 32 rows
        float4 v[NU];
        const int j = tid & 7, gx = ox0 - 4 + 4 * j;  // float4 j of the row (6 used)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int R = (tid >> 3) + 32 * u;
            const int c = (R >= IW) + (R >= 2 * IW), r = R - IW * c, gy = oy0 - 2 + r;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < 6 && R < 3 * IW && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                v[u] = *reinterpret_cast<const float4*>(img + (((size_t)n * 3 + c) * H + gy) * W + gx);
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int R = (tid >> 3) + 32 * u;
            if (j < 6 && R < 3 * IW) xin4[R * (XS / 4) + j] = v[u];
        }
    } else {  // any width: one float per thread, 8 rows of 32 columns (24 used) per pass
        constexpr int NU = (3 * IW + 7) / 8;  // passes of 8 rows
        float v[NU];
        const int q = tid & 31, gx = ox0 - 4 + q;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int R = (tid >> 5) + 8 * u;
            const int c = (R >= IW) + (R >= 2 * IW), r = R - IW * c, gy = oy0 - 2 + r;
            v[u] = 0.0f;
            if (q < XS && R < 3 * IW && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                v[u] = img[(((size_t)n * 3 + c) * H + gy) * W + gx];
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int R = (tid >> 5) + 8 * u;
            if (q < XS && R < 3 * IW) xin[R * XS + q] = v[u];
        }
    }
    __syncthreads();
    // (2) conv0 + BN + ReLU on the halo patch (fp32, same order of operations as stem_kernel) -> hi / lo planes
    for (int m = tid; m < MW * MROWS; m += NTHR) {
        const int r = m / MW, q = m - r * MW;
        const int gy = oy0 - 1 + r, gx = ox0 - 1 + q;
        f16x8 hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = {0, 0, 0, 0, 0, 0, 0, 0};
        if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
            float acc[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = 0.0f;
            const float* xp = xin + r * XS + q + 2;  // input x = gx - 1 + kx = (ox0 - 4) + q + 2 + kx
#pragma unroll 1
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
#pragma unroll
            for (int c = 0; c < 8; c += 2) {
                const f32x2_t x = {fmaxf(acc[c] + cs0[c], 0.0f), fmaxf(acc[c + 1] + cs0[c + 1], 0.0f)};
                const f16x2_t h = __builtin_convertvector(x, f16x2_t);
                const f32x2_t d = (x - __builtin_convertvector(h, f32x2_t)) * PMN_F16S_LO_SCALE;
                const f16x2_t l = __builtin_convertvector(d, f16x2_t);
                hi[c] = h[0];
                hi[c + 1] = h[1];
                lo[c] = l[0];
                lo[c + 1] = l[1];
            }
        }
        *reinterpret_cast<f16x8*>(midh + (r * MRP + q) * 8) = hi;
        *reinterpret_cast<f16x8*>(midl + (r * MRP + q) * 8) = lo;
    }
    __syncthreads();
    // (3) conv1: D[cout][pixel] += W[cout][k] * mid[k][pixel]; a wave owns TH / 4 output rows, four at a time
#pragma unroll 1
    for (int rg = 0; rg < TH / 16; ++rg) {
    const int wrow = wave * (TH / 4) + 4 * rg;
    f32x4_t accM[4], accL[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        accM[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        accL[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        int q = 4 * ks + kb;
        q = q < 8 ? q : 8;  // padding blocks 9..11 (zero weights) read tap 8
        const int dy = q / 3, dx = q - dy * 3;
        // (rows up to 4*3 + 3 + 2 = 17 and columns up to 15 + 2 = 17: inside the 18 x 18 slots)
        const _Float16* pb = midh + ((wrow + dy) * MRP + li + dx) * 8;
        f16x8 bh[4], blo[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            bh[t] = *reinterpret_cast<const f16x8*>(pb + t * MRP * 8);
            blo[t] = *reinterpret_cast<const f16x8*>(pb + t * MRP * 8 + MROWS * MRP * 8);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) accM[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ks][0], bh[t], accM[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) accL[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ks][0], blo[t], accL[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) accL[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ks][1], bh[t], accL[t], 0, 0, 0);
    }
    // D rows 4 kb + r = output channels: lanes with kb < 2 hold channels [4 kb, 4 kb + 4) of pixel (ox0 + li, oy0 + 4 wave + t)
    const int ox = ox0 + li, oyw = oy0 + wrow;
    if (kb < 2 && ox < W) {
        const f32x4_t sh = *reinterpret_cast<const f32x4_t*>(s1 + 4 * kb);
        float* po = out + (((size_t)n * H + oyw) * W + ox) * 8 + 4 * kb;
        const size_t rs = (size_t)W * 8;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (oyw + t < H) {
                f32x4_t v = accM[t] + accL[t] * (1.0f / PMN_F16S_LO_SCALE) + sh;
                v = __builtin_elementwise_max(v, f32x4_t{0.f, 0.f, 0.f, 0.f});
                *reinterpret_cast<f32x4_t*>(po + t * rs) = v;
            }
        }
    }
    }
}

// img [N,3,H,W] planar; w0 [3][3][3][8] / s0 [8] (pack_conv layout, fp32); w1a DEVICE fp16 [3][2][64][8] (params.pack_stem_conv1_f16s:
// conv1's weights as MFMA A operands, hi | lo, BatchNorm scale folded in float64); s1 [8] -> out [N,H,W,8] channels-last float32.
extern "C" int pmn_stem_f16s(const float* img, const float* w0, const float* s0, const void* w1a, const float* s1, float* out,
                             int N, int H, int W, void* stream) {
    if (!img || !w0 || !s0 || !w1a || !s1 || !out || N < 1 || H < 1 || W < 1) return PMN_ERR_ARG;
    constexpr int TS = 16, TH = PMN_STEM_TH;
    const int blocks = N * ((W + TS - 1) / TS) * ((H + TH - 1) / TH);
    // aligned float4 staging needs 16-byte aligned image rows: W % 4 == 0 and a 16-byte aligned base
    if (W % 4 == 0 && (reinterpret_cast<uintptr_t>(img) & 15) == 0)
        PMN_LAUNCH((stem_f16s_kernel<true, PMN_STEM_TH>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, w0, s0,
                           reinterpret_cast<const f16x8*>(w1a), s1, out, N, H, W, nullptr, 1);
    else
        PMN_LAUNCH((stem_f16s_kernel<false, PMN_STEM_TH>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, w0, s0,
                           reinterpret_cast<const f16x8*>(w1a), s1, out, N, H, W, nullptr, 1);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

// The same for `views` separately allocated images of one size: img_table DEVICE array of `views` addresses, entry v = a dense
// [B,3,H,W] float32 tensor, every address 16-byte aligned (the caller's contract: the library cannot see the entries when it
// launches); out [views*B,H,W,8] view-major, as the stacked call would write it.  The table is read when the kernel RUNS: a captured
// launch (HIP graph) follows whatever the table holds at replay time -- patchmatchnet_amd/graph.py reads a sample's images where
// the caller left them instead of copying them into static buffers (6 x 23 MB per 1600x1200 sample).
extern "C" int pmn_stem_f16s_views(const float* const* img_table, int views, const float* w0, const float* s0, const void* w1a,
                                   const float* s1, float* out, int B, int H, int W, void* stream) {
    if (!img_table || !w0 || !s0 || !w1a || !s1 || !out || views < 1 || B < 1 || H < 1 || W < 1) return PMN_ERR_ARG;
    constexpr int TS = 16, TH = PMN_STEM_TH;
    const int N = views * B;
    const int blocks = N * ((W + TS - 1) / TS) * ((H + TH - 1) / TH);
    if (W % 4 == 0)
        PMN_LAUNCH((stem_f16s_kernel<true, PMN_STEM_TH>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, nullptr, w0, s0,
                           reinterpret_cast<const f16x8*>(w1a), s1, out, N, H, W, img_table, B);
    else
        PMN_LAUNCH((stem_f16s_kernel<false, PMN_STEM_TH>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, nullptr, w0, s0,
                           reinterpret_cast<const f16x8*>(w1a), s1, out, N, H, W, img_table, B);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}
