// conv_wino.hip -- Winograd F(2x2,3x3) fp32 convolution on the matrix cores for FeatureNet's 3x3 stride-1 layers with
// cin == cout == C in {16, 32, 64} (reference models/net.py:21-23, 26-27, 30-31: conv3/4, conv6/7, conv9/10), conv + folded
// BatchNorm shift + ReLU, channels-last in and out.
//
// Why: these six layers are 80 of the 156 GFLOP of a six-view FeatureNet, and every convolution kernel of this library is
// bound by its multiply rate (VALU 81-83 % busy, matrix pipe ~60 % busy).  F(2x2,3x3) spends 16 multiplies per 2x2 output
// tile and input channel instead of 36: Y = A^T [ (G g G^T) . (B^T d B) ] A.  The filter transform U = G g G^T is done on the
// host in float64 (params.pack_conv_wino); in fp32 the result carries the same 2-3e-7 relative error as a direct fp32
// convolution on this network's own layers and activations (scripts/winograd_study.py).
//
// Mapping (C = 32, 64; the 16-channel kernel further down keeps the transform in registers).  A workgroup owns 32 tiles = 8 x 16 output pixels (4 x 8 tiles) and all C output channels.  Per chunk of 16 input
// channels: (1) the 10 x 18 input patch goes to LDS (zero-filled outside the image); (2) thread (tile, channel quad) forms the
// 16 transformed values B^T d B of its tile -- adds only -- and writes V[pos][tile][16] to LDS; (3) for each of the 16
// transform positions the waves run  M_pos[16 tiles x 16 couts] += V_pos[16 tiles x 16 cin] . U_pos[16 cin x 16 couts]
// as four v_mfma_f32_16x16x4_f32 (exact fp32).  A wave owns one block of 16 output channels and one or two groups of 16
// tiles, so the inverse transform never crosses lanes: it is linear, and every finished M_pos is folded straight into the
// wave's four output accumulators Y[a][b] with its +-1/0 coefficient A^T[a][p] A^T[b][q] (no 16-position accumulator file).
//   A operand  lane (i = lane&15, kq = lane>>4): ONE ds_read_b128 = channels [4 kq, +4) of tile i at the position; element m
//              feeds MFMA m (k index kq <-> input channel 4 kq + m)
//   B operand  U repacked on the host to [chunk][pos][cout block][64 lanes][4]: one contiguous 1 KB load per wave, positions
//              run 4 deep ahead in a register ring
//   C/D        16x16 MFMA: lane holds column (lane&15) = output channel, rows 4*(lane>>4) + r = tiles
#include "../pmn_common.hpp"
#include "../../../include/pmn_hip_experimental.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WinoArgs {
    int N, H, W, relu;
};

__device__ __forceinline__ float4 f4sub(const float4 a, const float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4add(const float4 a, const float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

template <int C>
__global__ __launch_bounds__(256, 3) void conv_wino_kernel(const float* __restrict__ in,
                                                                             const float4* __restrict__ wU,
                                                                             const float* __restrict__ shift,
                                                                             float* __restrict__ out, const WinoArgs a) {
    static_assert(C == 32 || C == 64, "16 channels: conv_wino16_kernel");
    constexpr int NCB = C / 16, TG = (C == 64 ? 2 : 1), NTHR = 256;
    // LDS banking (ds_read_b128 / ds_write_b128: a 256-B bank row = 16 slots of 16 B; a read is served in four fixed 16-lane
    // groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32; MI355X_MICROARCH.md, LDS): round 2 measured 39-42 % of this kernel's LDS
    // cycles as bank conflicts.  (a) patch pitch 24 words (6 slots) per pixel: the transform phase's lanes (tile = tid >> 2,
    // quad = tid & 3) of one group cover four tiles {0,3,5,6} / {1,2,4,7} two pixels apart -> base slots 12 tx mod 16 =
    // {0,4,12,8} / {12,8,0,4}: conflict-free (pitch 20: 10 tx mod 16 = {0,14,2,12}: two-way).  (b) V keeps 16 words (4 slots)
    // per tile, but a tile's four channel quads are stored at sub-slot quad ^ 2*((tile >> 3) & 1): the MFMA phase's groups hold
    // tiles j in {0-3,12-15} with one k-quad and j in {4-11} with the next, and 4 j mod 16 alone puts j and j + 4 on one slot.
    constexpr int CCP = 24, PH = 10, PW = 18, VP = 16, NT = 32;
    __shared__ float4 P4[PH * PW * CCP / 4];
    __shared__ float4 V4[16 * NT * VP / 4];
    float* P = reinterpret_cast<float*>(P4);
    float* V = reinterpret_cast<float*>(V4);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, j = lane & 15, kq = lane >> 4;
    const int cb = wave % NCB, g0 = (C == 64) ? 0 : wave / NCB;
    const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + 7) / 8;
    const int bt = pmn_xcd_tile(blockIdx.x, a.N * tiles_x * tiles_y);
    const int n = bt / (tiles_x * tiles_y), tr = bt - n * tiles_x * tiles_y;
    const int oy0 = (tr / tiles_x) * 8, ox0 = (tr % tiles_x) * 16;

    f32x4 Y[TG][2][2];
#pragma unroll
    for (int g = 0; g < TG; ++g)
#pragma unroll
        for (int p = 0; p < 4; ++p) Y[g][p >> 1][p & 1] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float4* bl = wU + (size_t)cb * 64 + lane;
#pragma unroll 1
    for (int cc = 0; cc < NCB; ++cc) {
        float4 bq[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) bq[d] = bl[(size_t)((cc * 16 + d) * NCB) * 64];
        if (cc) __syncthreads();  // the previous chunk's MFMA phase is done with V (and P is long free)
        {   // (1) patch: PH x PW pixels x 4 channel quads, every load of the thread in flight before the first LDS write
            constexpr int TOT = PH * PW * 4, NL = (TOT + NTHR - 1) / NTHR;
            float4 v[NL];
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const int idx = tid + k * NTHR, pix = idx >> 2, q = idx & 3;
                const int gy = oy0 - 1 + pix / PW, gx = ox0 - 1 + pix % PW;
                v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < TOT && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
                    v[k] = *reinterpret_cast<const float4*>(in + (((size_t)n * a.H + gy) * a.W + gx) * C + cc * 16 + 4 * q);
            }
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const int idx = tid + k * NTHR, pix = idx >> 2, q = idx & 3;
                if (idx < TOT) *reinterpret_cast<float4*>(P + pix * CCP + 4 * q) = v[k];
            }
        }
        __syncthreads();
        if (tid < NT * 4) {  // (2) input transform B^T d B of (tile, channel quad)
            const int t = tid >> 2, c4 = tid & 3, ty = t >> 3, tx = t & 7;
            const float* pp = P + ((2 * ty) * PW + 2 * tx) * CCP + 4 * c4;
            float4 w[4][4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float4 d0 = *reinterpret_cast<const float4*>(pp + (0 * PW + b) * CCP);
                const float4 d1 = *reinterpret_cast<const float4*>(pp + (1 * PW + b) * CCP);
                const float4 d2 = *reinterpret_cast<const float4*>(pp + (2 * PW + b) * CCP);
                const float4 d3 = *reinterpret_cast<const float4*>(pp + (3 * PW + b) * CCP);
                w[0][b] = f4sub(d0, d2);
                w[1][b] = f4add(d1, d2);
                w[2][b] = f4sub(d2, d1);
                w[3][b] = f4sub(d1, d3);
            }
            float* vp = V + t * VP + 4 * (c4 ^ (2 * ((t >> 3) & 1)));
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                *reinterpret_cast<float4*>(vp + ((4 * p + 0) * NT) * VP) = f4sub(w[p][0], w[p][2]);
                *reinterpret_cast<float4*>(vp + ((4 * p + 1) * NT) * VP) = f4add(w[p][1], w[p][2]);
                *reinterpret_cast<float4*>(vp + ((4 * p + 2) * NT) * VP) = f4sub(w[p][2], w[p][1]);
                *reinterpret_cast<float4*>(vp + ((4 * p + 3) * NT) * VP) = f4sub(w[p][1], w[p][3]);
            }
        }
        __syncthreads();
        // (3) 16 positions: M_pos = V_pos . U_pos over this chunk's 16 input channels, folded into Y with the inverse-transform
        // coefficient of the position.  Two positions at a time (independent MFMA chains: the 16x16x4 result latency is 40 cycles
        // against a 32-cycle issue); sched_barriers keep the B ring reloads behind the MFMAs that read the slot.
        const float* va = V + (g0 * 16 + j) * VP + 4 * (kq ^ (2 * (j >> 3)));
#pragma unroll
        for (int pos = 0; pos < 16; pos += 2) {
            f32x4 M[2][TG];
            float4 av[2][TG];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int g = 0; g < TG; ++g) {
                    M[u][g] = f32x4{0.f, 0.f, 0.f, 0.f};
                    av[u][g] = *reinterpret_cast<const float4*>(va + ((pos + u) * NT + g * 16) * VP);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float4 b4 = bq[(pos + u) & 3];
                    const float bf = m == 0 ? b4.x : m == 1 ? b4.y : m == 2 ? b4.z : b4.w;
#pragma unroll
                    for (int g = 0; g < TG; ++g) {
                        const float4 a4 = av[u][g];
                        const float af = m == 0 ? a4.x : m == 1 ? a4.y : m == 2 ? a4.z : a4.w;
                        M[u][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, M[u][g], 0, 0, 0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (pos + u + 4 < 16) bq[(pos + u) & 3] = bl[(size_t)((cc * 16 + pos + u + 4) * NCB) * 64];
                // inverse transform A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]]: coefficient of position (p,q) in output (a,b)
                const int p = (pos + u) >> 2, q = (pos + u) & 3;
#pragma unroll
                for (int oa = 0; oa < 2; ++oa)
#pragma unroll
                    for (int ob = 0; ob < 2; ++ob) {
                        const int ca = oa == 0 ? (p < 3 ? 1 : 0) : (p == 0 ? 0 : p == 1 ? 1 : -1);
                        const int cbq = ob == 0 ? (q < 3 ? 1 : 0) : (q == 0 ? 0 : q == 1 ? 1 : -1);
                        const int cf = ca * cbq;
#pragma unroll
                        for (int g = 0; g < TG; ++g) {
                            if (cf == 1) Y[g][oa][ob] += M[u][g];
                            if (cf == -1) Y[g][oa][ob] -= M[u][g];
                        }
                    }
            }
        }
    }

    // epilogue: + shift (folded BatchNorm), ReLU, channels-last store: lane = output channel 16 cb + j, rows r = tiles 4 kq + r
    const float sh = shift[cb * 16 + j];
#pragma unroll
    for (int g = 0; g < TG; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = (g0 + g) * 16 + kq * 4 + r, ty = t >> 3, tx = t & 7;
#pragma unroll
            for (int oa = 0; oa < 2; ++oa)
#pragma unroll
                for (int ob = 0; ob < 2; ++ob) {
                    const int oy = oy0 + 2 * ty + oa, ox = ox0 + 2 * tx + ob;
                    float v = Y[g][oa][ob][r] + sh;
                    if (a.relu) v = fmaxf(v, 0.0f);
                    if (oy < a.H && ox < a.W) out[(((size_t)n * a.H + oy) * a.W + ox) * C + cb * 16 + j] = v;
                }
        }
}

// ---- C = 16 (conv3 / conv4 at half resolution): register-resident input transform ------------------------------------------
// With a single block of 16 output channels nothing shares the transformed tile between waves, so the V buffer of the general
// kernel (32 KB of LDS per 32 tiles: 6 waves per CU, transforms and barriers dominating the few MFMAs) is dropped: lane (tile i,
// channel quad kq) reads the 16 patch values of ITS operand (16 ds_read_b128), transforms them in registers and feeds the 64
// MFMAs of its wave directly.  Workgroup = 4 waves x 16 tiles (16 x 16 pixels), LDS = the 18 x 18 patch only (26 KB).
__device__ __forceinline__ constexpr int wino16_tile_row(int j) { return (j >= 4 && j < 12) ? 1 : 0; }
__device__ __forceinline__ constexpr int wino16_tile_col(int j) { return (j >= 4 && j < 12) ? j - 4 : (j < 4 ? j : j - 8); }

__global__ __launch_bounds__(256, 3) void conv_wino16_kernel(const float* __restrict__ in, const float4* __restrict__ wU,
                                                            const float* __restrict__ shift, float* __restrict__ out,
                                                            const WinoArgs a) {
    constexpr int C = 16, CCP = 20, PH = 18, PW = 18, NTHR = 256;
    __shared__ float4 P4[PH * PW * CCP / 4];
    float* P = reinterpret_cast<float*>(P4);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, j = lane & 15, kq = lane >> 4;
    const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + 15) / 16;
    const int bt = pmn_xcd_tile(blockIdx.x, a.N * tiles_x * tiles_y);
    const int n = bt / (tiles_x * tiles_y), tr = bt - n * tiles_x * tiles_y;
    const int oy0 = (tr / tiles_x) * 16, ox0 = (tr % tiles_x) * 16;

    float4 bq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) bq[d] = wU[(size_t)d * 64 + lane];
    {
        constexpr int TOT = PH * PW * 4, NL = (TOT + NTHR - 1) / NTHR;
        float4 v[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int idx = tid + k * NTHR, pix = idx >> 2, q = idx & 3;
            const int gy = oy0 - 1 + pix / PW, gx = ox0 - 1 + pix % PW;
            v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < TOT && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
                v[k] = *reinterpret_cast<const float4*>(in + (((size_t)n * a.H + gy) * a.W + gx) * C + 4 * q);
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int idx = tid + k * NTHR, pix = idx >> 2, q = idx & 3;
            if (idx < TOT) *reinterpret_cast<float4*>(P + pix * CCP + 4 * q) = v[k];
        }
    }
    __syncthreads();

    // this lane's A operands: one of the wave's 16 tiles (2 rows x 8 columns), channels [4 kq, +4), all 16 positions.  Which tile
    // MFMA row j stands for is free, and it decides the LDS bank conflicts: a ds_read_b128 is served in the lane groups
    // {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32), i.e. rows j in {0-3,12-15} with quad kq and j in {4-11} with quad kq + 1.  The slot
    // (16 B) of a tile is 180 trow + 10 tcol + kq mod 16 = an even number that takes all 8 even values along a tile row, and the
    // SAME 8 values in the other row -- so rows {0-3,12-15} take tile row 0 and rows {4-11} tile row 1: 16 distinct slots per
    // group (round 2's j -> (j >> 3, j & 7) put j and j + 4 ... on one slot: 48 % of the LDS cycles were conflicts).
    const int ty = 2 * wave + wino16_tile_row(j), tx = wino16_tile_col(j);
    const float* pp = P + ((2 * ty) * PW + 2 * tx) * CCP + 4 * kq;
    float4 V[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const float4 d0 = *reinterpret_cast<const float4*>(pp + (0 * PW + b) * CCP);
        const float4 d1 = *reinterpret_cast<const float4*>(pp + (1 * PW + b) * CCP);
        const float4 d2 = *reinterpret_cast<const float4*>(pp + (2 * PW + b) * CCP);
        const float4 d3 = *reinterpret_cast<const float4*>(pp + (3 * PW + b) * CCP);
        V[0][b] = f4sub(d0, d2);
        V[1][b] = f4add(d1, d2);
        V[2][b] = f4sub(d2, d1);
        V[3][b] = f4sub(d1, d3);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float4 w0 = V[p][0], w1 = V[p][1], w2 = V[p][2], w3 = V[p][3];
        V[p][0] = f4sub(w0, w2);
        V[p][1] = f4add(w1, w2);
        V[p][2] = f4sub(w2, w1);
        V[p][3] = f4sub(w1, w3);
    }

    f32x4 Y[2][2];
#pragma unroll
    for (int p = 0; p < 4; ++p) Y[p >> 1][p & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pos = 0; pos < 16; pos += 2) {
        f32x4 M[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float4 b4 = bq[(pos + u) & 3], a4 = V[(pos + u) >> 2][(pos + u) & 3];
                const float bf = m == 0 ? b4.x : m == 1 ? b4.y : m == 2 ? b4.z : b4.w;
                const float af = m == 0 ? a4.x : m == 1 ? a4.y : m == 2 ? a4.z : a4.w;
                M[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, M[u], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (pos + u + 4 < 16) bq[(pos + u) & 3] = wU[(size_t)(pos + u + 4) * 64 + lane];
            const int p = (pos + u) >> 2, q = (pos + u) & 3;
#pragma unroll
            for (int oa = 0; oa < 2; ++oa)
#pragma unroll
                for (int ob = 0; ob < 2; ++ob) {
                    const int ca = oa == 0 ? (p < 3 ? 1 : 0) : (p == 0 ? 0 : p == 1 ? 1 : -1);
                    const int cq = ob == 0 ? (q < 3 ? 1 : 0) : (q == 0 ? 0 : q == 1 ? 1 : -1);
                    if (ca * cq == 1) Y[oa][ob] += M[u];
                    if (ca * cq == -1) Y[oa][ob] -= M[u];
                }
        }
    }
    const float sh = shift[j];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = kq * 4 + r, oty = 2 * wave + wino16_tile_row(t), otx = wino16_tile_col(t);  // D row t = MFMA row t's tile
#pragma unroll
        for (int oa = 0; oa < 2; ++oa)
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                const int oy = oy0 + 2 * oty + oa, ox = ox0 + 2 * otx + ob;
                float v = Y[oa][ob][r] + sh;
                if (a.relu) v = fmaxf(v, 0.0f);
                if (oy < a.H && ox < a.W) out[(((size_t)n * a.H + oy) * a.W + ox) * C + j] = v;
            }
    }
}

template <int C>
static int launch_wino(const float* in, const float* w, const float* shift, float* out, WinoArgs a, hipStream_t st) {
    const int blocks = a.N * ((a.W + 15) / 16) * ((a.H + 7) / 8);
    PMN_LAUNCH(conv_wino_kernel<C>, dim3(blocks), dim3(256), 0, st, in,
                       reinterpret_cast<const float4*>(w), shift, out, a);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

// in [N,H,W,C] channels-last; weights DEVICE float [C/16][16][C/16][64][4] (params.pack_conv_wino: U = G g G^T with the
// BatchNorm scale folded in, in float64, laid out in B-operand lane order); shift DEVICE float[C]; out [N,H,W,C].
// 3x3, stride 1, padding 1, cin == cout == C in {16, 32, 64}.
extern "C" int pmn_conv3x3_wino(const float* in, const float* weights, const float* shift, float* out, int N, int H, int W, int C,
                                int relu, void* stream) {
    if (!in || !weights || !shift || !out || N < 1 || H < 1 || W < 1) return PMN_ERR_ARG;
    WinoArgs a;
    a.N = N; a.H = H; a.W = W; a.relu = relu;
    hipStream_t st = (hipStream_t)stream;
    if (C == 64) return launch_wino<64>(in, weights, shift, out, a, st);
    if (C == 32) return launch_wino<32>(in, weights, shift, out, a, st);
    if (C == 16) {
        const int blocks = N * ((W + 15) / 16) * ((H + 15) / 16);
        PMN_LAUNCH(conv_wino16_kernel, dim3(blocks), dim3(256), 0, st, in, reinterpret_cast<const float4*>(weights), shift, out, a);
        PMN_CHECK_LAUNCH();
        return PMN_OK;
    }
    return PMN_ERR_SHAPE;
}

// =================================================================================================================================
// 5x5 stride-2 ConvBnReLU layers (conv2 8->16, conv5 16->32, conv8 32->64; reference models/net.py:20, 24, 28) in Winograd form.
//
// A stride-2 convolution splits into four stride-1 convolutions on the four parity sub-images of the input:
//   out = sum over (r,s) in {0,1}^2 of  X_rs (*) W_rs,   X_rs[i,j] = in[2i+r, 2j+s],   W_rs[a,b] = w[2a+r, 2b+s]
// with 3x3, 3x2, 2x3 and 2x2 taps.  Each runs as minimal filtering per dimension -- F(2,3): 4 products, F(2,2): 3 products for 2
// outputs -- so a 2x2 output tile costs 16 + 12 + 12 + 9 = 49 multiplies per (cin, cout) instead of 100, and all four share the
// same output tile grid, i.e. the same accumulators.  In fp32 the error is 1-2e-7 relative (below the direct form's 4-5e-7 on this
// network's layers, scripts/winograd_study.py).
//
// Kernel = conv_wino16_kernel's scheme: a wave owns 16 tiles (2 x 8 tiles = 4 x 16 output pixels) x 16 output channels and
// transforms ITS OWN operands in registers straight from the staged input patch (7 x 7 input pixels per tile, tiles 4 apart),
// phase by phase (16 / 12 / 12 / 9 values live at a time), 8 input channels per chunk: lane (tile i = lane&15, kq = lane>>4) holds
// channels 2 kq + {0,1} of its tile (ds_read_b64), element m feeds MFMA m of the position (k index kq <-> channel 2 kq + m).
// Filter transforms G_r W_rs G_s^T are done on the host in float64 (params.pack_conv5x5s2_wino) and stored [chunk][49 positions]
// [cout block][64 lanes][2]; positions run 7 deep ahead in a register ring.  The inverse transforms are linear, so every finished
// position is folded into the four output accumulators with its coefficient (0, +1, -1).
// =================================================================================================================================
typedef float f32x2 __attribute__((ext_vector_type(2)));

// coefficient of transformed position p (of a dimension in phase `odd`) in output o (0/1): A^T = [[1,1,1,0],[0,1,-1,-1]] for the
// 3-tap (even) phase, [[1,1,0],[0,1,-1]] for the 2-tap (odd) phase
__device__ __forceinline__ constexpr int w5_at(int odd, int o, int p) {
    return odd ? (o == 0 ? (p < 2 ? 1 : 0) : (p == 0 ? 0 : p == 1 ? 1 : -1))
               : (o == 0 ? (p < 3 ? 1 : 0) : (p == 0 ? 0 : p == 1 ? 1 : -1));
}

// in-place input transform of one dimension: even phase (4 points) B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],
// odd phase (3 points) [[1,-1,0],[0,1,0],[0,1,-1]]
template <int ODD>
__device__ __forceinline__ void w5_bt(f32x2& d0, f32x2& d1, f32x2& d2, f32x2& d3) {
    if (ODD) {
        const f32x2 t0 = d0 - d1, t2 = d1 - d2;
        d0 = t0; d2 = t2;
    } else {
        const f32x2 t0 = d0 - d2, t1 = d1 + d2, t2 = d2 - d1, t3 = d1 - d3;
        d0 = t0; d1 = t1; d2 = t2; d3 = t3;
    }
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256, 3) void conv5x5s2_wino_kernel(const float* __restrict__ in, const f32x2* __restrict__ wU,
                                                               const float* __restrict__ shift, float* __restrict__ out,
                                                               const WinoArgs a, const int Ho, const int Wo, const int nch) {
    // nch = CIN / 8 chunks of 8 input channels, passed at run time on purpose: with a compile-time single trip (conv2) hipcc
    // flattens the chunk loop, hoists all 49 weight loads to the top and spills
    constexpr int NCB = COUT / 16, NG = 4 / NCB;                      // cout blocks, tile groups per workgroup
    constexpr int PH = 8 * NG + 3, PW = 35, PP = 10;                  // patch rows / cols / words per pixel (8 channels + 2 pad)
    // Row pitch RP = PW * PP + 4 words, and the pixels of patch row y are shifted by 4 * ((y >> 2) & 1) words.  A ds_read_b64 is
    // served in the two 32-lane halves {tile j = 0..15} x {kq, kq + 1}; the lane's word is 4 RP trow + 40 tcol + 2 kq, 40 tcol mod 64
    // runs through the 8 multiples of 8 and 4 RP = 8 mod 64 -- so without the shift tile rows 0 and 1 (4 patch rows apart) sit on
    // the SAME 8 bank groups and every read is a two-way conflict (round 2: 70-72 % of this kernel's LDS cycles).  Patch rows 4
    // apart always differ in (y >> 2) & 1, so one of the two tile rows is shifted by half a bank group: 64 distinct banks.
    constexpr int RP = PW * PP + 4;
    constexpr int NPOS = 49, RING = 7, NTHR = 256;  // 49 % RING == 0: a position keeps its ring slot from chunk to chunk
    extern __shared__ float4 w5_lds4[];
    float* P = reinterpret_cast<float*>(w5_lds4);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, j = lane & 15, kq = lane >> 4;
    const int cb = wave % NCB, g = wave / NCB;
    const int tiles_x = (Wo + 15) / 16, tiles_y = (Ho + 4 * NG - 1) / (4 * NG);
    const int bt = pmn_xcd_tile(blockIdx.x, a.N * tiles_x * tiles_y);
    const int n = bt / (tiles_x * tiles_y), tr = bt - n * tiles_x * tiles_y;
    const int oy0 = (tr / tiles_x) * (4 * NG), ox0 = (tr % tiles_x) * 16;
    const int iy0 = 2 * oy0 - 2, ix0 = 2 * ox0 - 2;

    f32x4 Y[2][2];
#pragma unroll
    for (int p = 0; p < 4; ++p) Y[p >> 1][p & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x2* bl = wU + (size_t)cb * 64 + lane;  // element ((cc*49 + pos)*NCB + cb)*64 + lane
    f32x2 bq[RING];
#pragma unroll
    for (int d = 0; d < RING; ++d) bq[d] = bl[(size_t)(d * NCB) * 64];

    // this lane's tile inside the group: 2 x 8 tiles, 4 input pixels apart; patch rows of the group start at 8 g
    const int ty = j >> 3, tx = j & 7;
    const float* pl = P + (8 * g + 4 * ty) * RP + 4 * tx * PP + 2 * kq;
    const int sw_lo = 4 * ty, sw_hi = 4 - 4 * ty;  // shift of this tile's patch rows 0..3 / 4..6 (rows 8 g + 4 ty + [0, 7))

#pragma unroll 1
    for (int cc = 0; cc < nch; ++cc) {
        if (cc) __syncthreads();  // every wave is done with the previous chunk's patch
        {   // patch: PH x 35 pixels x 2 channel quads; at most SB loads of a thread in flight per batch (register budget)
            constexpr int TOT = PH * PW * 2, NL = (TOT + NTHR - 1) / NTHR, SB = 5;
#pragma unroll
            for (int k0 = 0; k0 < NL; k0 += SB) {
                float4 v[SB];
#pragma unroll
                for (int k = 0; k < SB; ++k) {
                    const int idx = tid + (k0 + k) * NTHR, pix = idx >> 1, q = idx & 1;
                    const int gy = iy0 + pix / PW, gx = ix0 + pix % PW;
                    v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (k0 + k < NL && idx < TOT && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
                        v[k] = *reinterpret_cast<const float4*>(in + (((size_t)n * a.H + gy) * a.W + gx) * CIN + cc * 8 + 4 * q);
                }
#pragma unroll
                for (int k = 0; k < SB; ++k) {
                    const int idx = tid + (k0 + k) * NTHR, pix = idx >> 1, q = idx & 1;
                    if (k0 + k < NL && idx < TOT) {  // pixel pitch 10 words: 8-byte aligned only
                        const int py = pix / PW, px = pix - py * PW;
                        float* dst = P + py * RP + px * PP + 4 * ((py >> 2) & 1) + 4 * q;
                        *reinterpret_cast<f32x2*>(dst) = f32x2{v[k].x, v[k].y};
                        *reinterpret_cast<f32x2*>(dst + 2) = f32x2{v[k].z, v[k].w};
                    }
                }
            }
        }
        __syncthreads();

        int pos = 0;  // compile-time after unrolling: position index 0..48 in the host's order (r, s, p, q)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                constexpr int dummy = 0;
                (void)dummy;
                const int nr = r ? 3 : 4, ns = s ? 3 : 4;
                // the phase's sub-grid of the tile's 7 x 7 patch: rows r, r+2, ..; columns s, s+2, ..
                f32x2 d[4][4];
#pragma unroll
                for (int ia = 0; ia < 4; ++ia)
#pragma unroll
                    for (int ib = 0; ib < 4; ++ib) {
                        d[ia][ib] = f32x2{0.f, 0.f};
                        if (ia < nr && ib < ns)
                            d[ia][ib] = *reinterpret_cast<const f32x2*>(pl + (r + 2 * ia) * RP + (s + 2 * ib) * PP +
                                                                        ((r + 2 * ia) >= 4 ? sw_hi : sw_lo));
                    }
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) {
                    if (r) w5_bt<1>(d[0][ib], d[1][ib], d[2][ib], d[3][ib]);
                    else w5_bt<0>(d[0][ib], d[1][ib], d[2][ib], d[3][ib]);
                }
#pragma unroll
                for (int ia = 0; ia < 4; ++ia) {
                    if (s) w5_bt<1>(d[ia][0], d[ia][1], d[ia][2], d[ia][3]);
                    else w5_bt<0>(d[ia][0], d[ia][1], d[ia][2], d[ia][3]);
                }
                // positions of the phase, two at a time (independent MFMA chains)
#pragma unroll
                for (int q0 = 0; q0 < nr * ns; q0 += 2) {
                    f32x4 M[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            if (q0 + u < nr * ns) {
                                const f32x2 a2 = d[(q0 + u) / ns][(q0 + u) % ns], b2 = bq[(pos + u) % RING];
                                M[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(m ? a2.y : a2.x, m ? b2.y : b2.x, M[u], 0, 0, 0);
                            }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        if (q0 + u < nr * ns) {
                            // ring slot free: fetch the position RING ahead (runs on into the next chunk; the tail re-reads the last)
                            const int nxt = min(cc * NPOS + pos + u + RING, nch * NPOS - 1);
                            bq[(pos + u) % RING] = bl[(size_t)(nxt * NCB) * 64];
                            const int p = (q0 + u) / ns, q = (q0 + u) % ns;
#pragma unroll
                            for (int oa = 0; oa < 2; ++oa)
#pragma unroll
                                for (int ob = 0; ob < 2; ++ob) {
                                    const int cf = w5_at(r, oa, p) * w5_at(s, ob, q);
                                    if (cf == 1) Y[oa][ob] += M[u];
                                    if (cf == -1) Y[oa][ob] -= M[u];
                                }
                        }
                    pos += (q0 + 1 < nr * ns) ? 2 : 1;
                }
            }
    }

    const float sh = shift[cb * 16 + j];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int t = kq * 4 + rr, oty = t >> 3, otx = t & 7;  // D rows = tiles 4 kq + rr of this wave's 16
#pragma unroll
        for (int oa = 0; oa < 2; ++oa)
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                const int oy = oy0 + 4 * g + 2 * oty + oa, ox = ox0 + 2 * otx + ob;
                float v = Y[oa][ob][rr] + sh;
                if (a.relu) v = fmaxf(v, 0.0f);
                if (oy < Ho && ox < Wo) out[(((size_t)n * Ho + oy) * Wo + ox) * COUT + cb * 16 + j] = v;
            }
    }
}

template <int CIN, int COUT>
static int launch_w5(const float* in, const float* w, const float* shift, float* out, WinoArgs a, hipStream_t st) {
    constexpr int NG = 4 / (COUT / 16);
    const int Ho = (a.H - 1) / 2 + 1, Wo = (a.W - 1) / 2 + 1;
    const size_t lds = (size_t)(8 * NG + 3) * (35 * 10 + 4) * sizeof(float);  // PH rows of RP words (kernel constants)
    auto kern = conv5x5s2_wino_kernel<CIN, COUT>;
    if (lds > 48 * 1024 && pmn_raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != PMN_OK) return PMN_ERR_LAUNCH;
    const int blocks = a.N * ((Wo + 15) / 16) * ((Ho + 4 * NG - 1) / (4 * NG));
    PMN_LAUNCH(kern, dim3(blocks), dim3(256), lds, st, in, reinterpret_cast<const f32x2*>(w), shift, out, a, Ho, Wo,
                       CIN / 8);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

// in [N,H,W,cin] channels-last; weights DEVICE float [cin/8][49][cout/16][64][2] (params.pack_conv5x5s2_wino); shift DEVICE
// float[cout]; out [N,Ho,Wo,cout] with Ho = (H-1)/2 + 1, Wo = (W-1)/2 + 1 (5x5, stride 2, padding 2).
// Supported (cin, cout): (8,16), (16,32), (32,64).
extern "C" int pmn_conv5x5s2_wino(const float* in, const float* weights, const float* shift, float* out, int N, int H, int W,
                                  int cin, int cout, int relu, void* stream) {
    if (!in || !weights || !shift || !out || N < 1 || H < 1 || W < 1) return PMN_ERR_ARG;
    WinoArgs a;
    a.N = N; a.H = H; a.W = W; a.relu = relu;
    hipStream_t st = (hipStream_t)stream;
    if (cin == 8 && cout == 16) return launch_w5<8, 16>(in, weights, shift, out, a, st);
    if (cin == 16 && cout == 32) return launch_w5<16, 32>(in, weights, shift, out, a, st);
    if (cin == 32 && cout == 64) return launch_w5<32, 64>(in, weights, shift, out, a, st);
    return PMN_ERR_SHAPE;
}
