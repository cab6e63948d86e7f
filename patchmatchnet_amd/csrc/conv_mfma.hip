// conv_mfma.hip -- fp32 implicit-GEMM convolution on the matrix cores for FeatureNet's wide layers (reference
// models/net.py:25-34: conv5..conv10, 16/32/64 channels), conv + folded BatchNorm shift + ReLU, channels-last in and out.
//
// Why: the VALU kernel of conv.hip feeds every FMA a wave-uniform SGPR weight.  From 32x32x3x3 weights up the filter bank no
// longer fits the 16 KB scalar cache, each 64-weight batch is an L2 round trip, and the layers run at 43-75 TFLOP/s
// (profiles/r01_final_*).  v_mfma_f32_32x32x2_f32 is exact fp32 (bitwise a k-ordered fmaf chain) at the same 157 TFLOP/s
// peak, takes ONE VGPR per operand per lane and leaves the scalar path out of the loop.
//
// GEMM view: rows = output pixels, columns = output channels, k = (tap, input channel).
//   workgroup  NW waves, a TH x 16 tile of output pixels (TH = 2*PG*NW rows), all COUT channels
//   wave       PG groups of 32 pixels (2 tile rows) x NT = COUT/32 column blocks -> PG*NT accumulators of 16 VGPRs
//   A operand  the (TH*S + K-1) x (16*S + K-1) x CC input patch staged in LDS (zero-filled outside the image); lane
//              (i = lane&31, h = lane>>5) reads ONE ds_read_b128 = channels [8*c8 + 4h, +4) of its pixel at the current tap
//   B operand  weights repacked on the host (params.pack_conv_mfma) as [tap][cin/8][NT][64 lanes][4]: a wave's
//              global_load_dwordx4 is one contiguous 1 KB line set, shared by every wave on the chip (L1/L2 resident)
//   k-step j   (0..3) multiplies A.j by B.j: k pairs (cin 8*c8+j, cin 8*c8+4+j) -- any k order is valid as long as A and B
//              agree, and the sum stays a plain fp32 fmaf chain.
// C/D layout of the 32x32 MFMA: lane holds column (lane&31), rows (reg&3) + 8*(reg>>2) + 4*(lane>>5): for a fixed register
// the 32 lanes of a half-wave store 32 consecutive output channels of one pixel (128 contiguous bytes).
#include "pmn_common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct MfmaConvArgs {
    int N, H, W, Ho, Wo, pad, relu, cout, ca;  // cout = real output channels (<= COUT), ca = channels that go to out (PLANAR)
};

// PLANAR (the offset heads propa_conv / eval_conv, reference models/patchmatch.py:288-311, dilated 3x3, bias, no ReLU): the two
// MFMA operands swap roles (rows = output channels, columns = pixels), so a register holds one output channel of 32
// pixels = two 64-byte runs of a [B,cout,h,w] plane; channels [0,ca) go to `out`, [ca,cout) to `out_b` (both heads of a stage
// are ONE convolution over the shared reference feature), channels >= cout are padding.
template <int CIN, int CC, int COUT, int K, int S, int DIL, int NW, int PG, int D, bool PLANAR>
__global__ __launch_bounds__(64 * NW, 3) void conv_mfma_kernel(const float* __restrict__ in, const float4* __restrict__ wB,
                                                          const float* __restrict__ shift, float* __restrict__ out,
                                                          float* __restrict__ out_b, const MfmaConvArgs a) {
    constexpr int TW = 16, TH = 2 * PG * NW, NT = COUT / 32, C8 = CC / 8, CCP = CC + 4, CQ = CC / 4, NTHR = 64 * NW;
    constexpr int IW = (TW - 1) * S + (K - 1) * DIL + 1, IH = (TH - 1) * S + (K - 1) * DIL + 1;
    constexpr int STEPS = K * K * C8;  // k-steps of 8 input channels per staged chunk: (tap, c8) flattened, c8 fastest
    constexpr int SB = 4;              // staging loads in flight per thread
    static_assert(CIN % CC == 0 && CC % 8 == 0 && COUT % 32 == 0 && STEPS % D == 0, "channel tiling");
    extern __shared__ float4 mf_lds4[];
    float* tile = reinterpret_cast<float*>(mf_lds4);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, h = lane >> 5, li = lane & 31;
    const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
    const int bt = pmn_xcd_tile(blockIdx.x, a.N * tiles_x * tiles_y);
    const int n = bt / (tiles_x * tiles_y), tr = bt - n * tiles_x * tiles_y;
    const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TW;
    const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;

    f32x16 acc[PG][NT];
#pragma unroll
    for (int g = 0; g < PG; ++g)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][t][r] = 0.0f;

    // A-side LDS offsets (words) of this lane's pixel in each pixel group, tap (0,0), first channel chunk
    int aoff[PG];
#pragma unroll
    for (int g = 0; g < PG; ++g) {
        const int ty = (wave * PG + g) * 2 + (li >> 4), tx = li & 15;
        aoff[g] = (ty * S * IW + tx * S) * CCP + 4 * h;
    }

    // B operands run D k-steps (D x 16*PG*NT/2... MFMA issue slots) ahead of their use in a register ring: the loads are L2
    // round trips (every wave streams the whole filter bank), far longer than the 4*PG*NT MFMAs of one step
    float4 bq[D][NT];
    const float4* bl = wB + lane;
    auto b_index = [&](int s, int cc0) { return ((s / C8) * (CIN / 8) + cc0 / 8 + (s % C8)) * NT * 64; };

#pragma unroll 1
    for (int cc0 = 0; cc0 < CIN; cc0 += CC) {
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int t = 0; t < NT; ++t) bq[d][t] = bl[b_index(d, cc0) + t * 64];
        if (cc0) __syncthreads();
        // staging in batches of SB loads per thread: one load per trip would expose a full HBM/L2 round trip per 16 bytes
        for (int base = tid; base < IH * IW * CQ; base += NTHR * SB) {
            float4 v[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int idx = base + u * NTHR, pix = idx / CQ, q = idx - pix * CQ;
                const int r = pix / IW, c = pix - r * IW;
                const int gy = iy0 + r, gx = ix0 + c;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < IH * IW * CQ && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
                    v[u] = *reinterpret_cast<const float4*>(in + (((size_t)n * a.H + gy) * a.W + gx) * CIN + cc0 + 4 * q);
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int idx = base + u * NTHR, pix = idx / CQ, q = idx - pix * CQ;
                if (idx < IH * IW * CQ) *reinterpret_cast<float4*>(tile + pix * CCP + 4 * q) = v[u];
            }
        }
        __syncthreads();
        // B operands run D-1 steps ahead in a D-deep register ring (L2 latency); A operands one step ahead inside a group of D
        // steps (ds_read latency).  A ring slot is re-loaded only AFTER the MFMAs that read it have issued (no copies, the
        // registers stay put across the back-edge); sched_barriers pin that order -- unfenced, hipcc sinks the loads to the
        // end of the group and waits for all of them
        auto a_offset = [&](int s) { return ((((s / C8) / K) * IW + (s / C8) % K) * DIL) * CCP + 8 * (s % C8); };
#pragma unroll 1
        for (int s0 = 0; s0 < STEPS; s0 += D) {
            float4 ar[2][PG];
#pragma unroll
            for (int g = 0; g < PG; ++g) ar[0][g] = *reinterpret_cast<const float4*>(tile + aoff[g] + a_offset(s0));
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int s = s0 + d;
                if (d + 1 < D) {
#pragma unroll
                    for (int g = 0; g < PG; ++g)
                        ar[(d + 1) & 1][g] = *reinterpret_cast<const float4*>(tile + aoff[g] + a_offset(s + 1));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int g = 0; g < PG; ++g) {
                        const float4 a4 = ar[d & 1][g];
                        const float af = j == 0 ? a4.x : j == 1 ? a4.y : j == 2 ? a4.z : a4.w;
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const float4 b4 = bq[d][t];
                            const float bf = j == 0 ? b4.x : j == 1 ? b4.y : j == 2 ? b4.z : b4.w;
                            acc[g][t] = PLANAR ? __builtin_amdgcn_mfma_f32_32x32x2f32(bf, af, acc[g][t], 0, 0, 0)
                                               : __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[g][t], 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                const int sp = min(s + D, STEPS - 1);  // the last D steps re-load the final step: no branch, no overrun
#pragma unroll
                for (int t = 0; t < NT; ++t) bq[d][t] = bl[b_index(sp, cc0) + t * 64];
            }
        }
    }

    // epilogue: + shift (folded BatchNorm / bias), ReLU, store
    if constexpr (PLANAR) {
        const int cb = a.cout - a.ca;
#pragma unroll
        for (int g = 0; g < PG; ++g) {
            const int oy = oy0 + (wave * PG + g) * 2 + (li >> 4), ox = ox0 + (li & 15);
            const bool inside = oy < a.Ho && ox < a.Wo;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (inside && co < a.cout) {
                        float v = acc[g][t][r] + shift[co];
                        if (a.relu) v = fmaxf(v, 0.0f);
                        if (co < a.ca) out[(((size_t)n * a.ca + co) * a.Ho + oy) * a.Wo + ox] = v;
                        else out_b[(((size_t)n * cb + (co - a.ca)) * a.Ho + oy) * a.Wo + ox] = v;
                    }
                }
            }
        }
    } else {
        // channels-last: channels [0,ca) -> out (pitch ca), [ca,cout) -> out_b (pitch cout-ca); ca == cout == COUT for the plain layers
        const int cb = a.cout - a.ca;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int co = t * 32 + li;
            const float sh = shift[co];
            float* dst = co < a.ca ? out + co : out_b + (co - a.ca);
            const int pitch = co < a.ca ? a.ca : cb;
            const bool live = co < a.cout;
#pragma unroll
            for (int g = 0; g < PG; ++g) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;  // pixel index inside the group of 32
                    const int oy = oy0 + (wave * PG + g) * 2 + (row >> 4), ox = ox0 + (row & 15);
                    float v = acc[g][t][r] + sh;
                    if (a.relu) v = fmaxf(v, 0.0f);
                    if (live && oy < a.Ho && ox < a.Wo) dst[(((size_t)n * a.Ho + oy) * a.Wo + ox) * pitch] = v;
                }
            }
        }
    }
}

template <int CIN, int CC, int COUT, int K, int S, int DIL, int NW, int PG, int D, bool PLANAR>
static int launch_mfma(const float* in, const float* w, const float* shift, float* out, float* out_b, MfmaConvArgs a,
                       hipStream_t st) {
    constexpr int TH = 2 * PG * NW, IW = 15 * S + (K - 1) * DIL + 1, IH = (TH - 1) * S + (K - 1) * DIL + 1;
    const size_t lds = (size_t)IH * IW * (CC + 4) * sizeof(float);
    auto kern = conv_mfma_kernel<CIN, CC, COUT, K, S, DIL, NW, PG, D, PLANAR>;
    if (lds > 48 * 1024 && pmn_raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != PMN_OK) return PMN_ERR_LAUNCH;
    const int blocks = a.N * ((a.Wo + 15) / 16) * ((a.Ho + TH - 1) / TH);
    PMN_LAUNCH(kern, dim3(blocks), dim3(64 * NW), lds, st, in, reinterpret_cast<const float4*>(w), shift, out, out_b,
                       a);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

// in [N,H,W,cin] channels-last; weights DEVICE float [K*K][cin/8][coutp/32][64][4] (params.pack_conv_mfma; coutp = cout rounded
// up to 32, BatchNorm scale folded in); shift DEVICE float[coutp].
//   planar == 0: (64, <=128, 1, 1) with the channels split between out [N,H,W,ca] and out_b [N,H,W,cout-ca] -- the only form of the
//                product library; research build (PMN_EXPERIMENTAL) also: out [N,Ho,Wo,cout] channels-last (out_b NULL, ca == cout ==
//                coutp, dil == 1) for (cin,cout,K,stride) = (64,64,3,1), (32,32,3,1), (32,64,5,2), (16,32,5,2)
//   planar == 1: research build only: out [N,ca,Ho,Wo] and out_b [N,cout-ca,Ho,Wo] (NULL when ca == cout) planar; K = 3, stride 1,
//                pad == dil; (cin,dil): (64,2), (32,4), (16,6) with cout <= 64 -- the offset heads of the default cascade
extern "C" int pmn_conv2d_mfma(const float* in, const float* weights, const float* shift, float* out, float* out_b, int N,
                               int H, int W, int cin, int cout, int ca, int K, int stride, int pad, int dil, int relu,
                               int planar, void* stream) {
    if (!in || !weights || !shift || !out || N < 1 || H < 1 || W < 1 || pad < 0 || stride < 1 || dil < 1 || cout < 1)
        return PMN_ERR_ARG;
    if (ca < 1 || ca > cout || (ca < cout) != (out_b != nullptr)) return PMN_ERR_ARG;
    MfmaConvArgs a;
    a.N = N; a.H = H; a.W = W; a.pad = pad; a.relu = relu; a.cout = cout; a.ca = ca;
    a.Ho = (H + 2 * pad - dil * (K - 1) - 1) / stride + 1;
    a.Wo = (W + 2 * pad - dil * (K - 1) - 1) / stride + 1;
    if (a.Ho < 1 || a.Wo < 1) return PMN_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (!planar) {
        if (dil != 1) return PMN_ERR_SHAPE;
        // 1x1, 64 -> (ca | cout-ca) <= 128 channels: the 1/8-resolution level of the folded FPN head (pmn_fpn_level's arithmetic)
        if (cin == 64 && K == 1 && stride == 1 && pad == 0 && cout <= 128)
            // (the 64 input channels staged in two chunks of 32: 58 us instead of 64 per six 150 x 200 maps, same bits --
            //  profiles/r06_fpn8_variants.log; 2 or 8 waves per workgroup, chunks of 16, two pixel groups per wave: slower)
            return launch_mfma<64, 32, 128, 1, 1, 1, 4, 1, 4, false>(in, weights, shift, out, out_b, a, st);
#ifndef PMN_EXPERIMENTAL
        return PMN_ERR_SHAPE;  // the product library carries the 1x1 form alone (the folded FPN head's 1/8 level)
    }
    return PMN_ERR_SHAPE;
}
#else  // research build: rounds 1-2's fp32 matrix-core forms of FeatureNet's wide layers and of the offset heads (superseded by
       // pmn_conv2d_f16s / pmn_offset_heads_f16s in round 3; tests/test_hip_parity.py runs them under PMN_EXPERIMENTAL=1)
        if (ca != cout) return PMN_ERR_SHAPE;
#define PMN_MFMA(CI, CCH, CO, KK, SS, NWV, PGV, DD) return launch_mfma<CI, CCH, CO, KK, SS, 1, NWV, PGV, DD, false>(in, weights, shift, out, out_b, a, st)
        // NW = 4 waves x one 32-pixel group each (8x16-pixel tiles): one wave per SIMD per workgroup, 80-128 VGPRs; measured
        // 13-20 % faster than 2 waves x 2 groups on the 64-column layers and the stride-2 layers (profiles/README.md)
        if (cin == 64 && cout == 64 && K == 3 && stride == 1) PMN_MFMA(64, 64, 64, 3, 1, 4, 1, 4);
        if (cin == 32 && cout == 32 && K == 3 && stride == 1) PMN_MFMA(32, 32, 32, 3, 1, 4, 1, 4);
        if (cin == 32 && cout == 64 && K == 5 && stride == 2) PMN_MFMA(32, 16, 64, 5, 2, 4, 1, 5);
        if (cin == 16 && cout == 32 && K == 5 && stride == 2) PMN_MFMA(16, 8, 32, 5, 2, 4, 1, 5);
#undef PMN_MFMA
        return PMN_ERR_SHAPE;
    }
    if (K != 3 || stride != 1 || pad != dil || cout > 64) return PMN_ERR_SHAPE;
#define PMN_HEAD(CI, CCH, DL, NWV, PGV, DD)                                                                              \
    do {                                                                                                                 \
        if (cout <= 32) return launch_mfma<CI, CCH, 32, 3, 1, DL, NWV, PGV, DD, true>(in, weights, shift, out, out_b, a, st); \
        return launch_mfma<CI, CCH, 64, 3, 1, DL, NWV, PGV, DD, true>(in, weights, shift, out, out_b, a, st);             \
    } while (0)
    if (cin == 64 && dil == 2) PMN_HEAD(64, 32, 2, 2, 1, 4);  // 1/8 resolution: 4x16-pixel tiles, or the grid is < 1 wave per SIMD
    if (cin == 32 && dil == 4) PMN_HEAD(32, 16, 4, 4, 1, 3);
    if (cin == 16 && dil == 6) PMN_HEAD(16, 16, 6, 4, 2, 3);
#undef PMN_HEAD
    return PMN_ERR_SHAPE;
}
#endif
