// gather_common.hpp -- pieces shared by the two implementations of the fused warp + gather + group-correlation kernel
// (gather_corr.hip: lane groups gather straight from HBM/L1; gather_win.hip: wave-private LDS windows of the source map).
#pragma once
#include <cstdlib>
#include <cstring>

#include "pmn_common.hpp"

typedef float pmn_f2 __attribute__((ext_vector_type(2)));

enum { MODE_VIEWS = 0, MODE_PIXELWISE = 1, MODE_NEIGHBOR = 2 };

struct GatherArgs {
    const float* ref;      // [B,h,w,C]
    const float* src;      // [N,B,hs,ws,C]  (stacked source views), or
    const unsigned long long* src_tab;  // DEVICE table of N addresses of [B,hs,ws,C] maps (pmn_warp_correlate_views); src then null
    const float* proj;     // [B,N,4,4]
    const float* depth;    // [B,D,h,w]
    const float* offsets;  // [B,2K,h,w]   (MODE_NEIGHBOR)
    const float* vw_in;    // [B,N,h>>s,w>>s]
    const float* mlp_a;    // device float[PMN_MLP_FLOATS]: similarity_net | feature_weight_net
    const float* mlp_b;    // device float[PMN_MLP_FLOATS]: pixel_wise_net
    float* vw_out;         // [B,N,h,w]
    int* vw_argmax;        // [B,N,h,w] or null
    float* sim_out;        // [B,G,D,h,w] or null
    float* out;            // [B,D,h,w]
    int B, N, D, h, w, hs, ws, vw_shift, ntiles;
    int table[2 * PMN_MAX_NEIGHBORS];
};

#define MLP_LDS_FLOATS PMN_MLP_FLOATS  // 340: a float4 multiple

__device__ __forceinline__ float pmn_pair_swap(float v) {
    // lane l <-> lane l^1 through DPP quad_perm [1,0,3,2]
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
}

// (pmn_settle / pmn_settle4 -- lesson 46's pin for scalars that are broadcast into packed math -- live in pmn_common.hpp)

// Pointwise MLP G -> 16 -> 8 -> 1 for NI items at once, weights read from LDS (uniform address = broadcast read).
// Layers 1 and 2 are fused in a ROLLED loop over the 16 hidden units: unit j of every item is produced from weight row
// j and immediately scattered into the 8 layer-2 accumulators with column j of the second weight matrix, so no array
// of hidden activations or of weights stays live (a fully unrolled form makes hipcc hoist all 73 row loads and spill).
// The packed block (params.py) is laid out for exactly this walk: per unit j one 20-float record
//   [0..7] w0[j][g] (BN folded, g < G used) | [8..15] w1[k][j] (BN folded) | [16] t0[j] | pad
// followed by t1[8] | w2[8] | b2.  Summation orders: layer 1 over g ascending, layer 2 over j ascending, layer 3 over
// k ascending, bias added last -- the same as the oracle's.
template <int G, int NI>
__device__ __forceinline__ void mlp_from_lds(const float* __restrict__ W, const float (&x)[NI][G], float (&out)[NI]) {
    float a1[NI][8];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) a1[i][k] = 0.0f;
#pragma unroll 1
    for (int j = 0; j < 16; ++j) {
        const float4* rp = reinterpret_cast<const float4*>(W + 20 * j);
        float w0[8], w1c[8];
        {
            const float4 r0 = rp[0];
            w0[0] = r0.x; w0[1] = r0.y; w0[2] = r0.z; w0[3] = r0.w;
            if (G == 8) {
                const float4 r1 = rp[1];
                w0[4] = r1.x; w0[5] = r1.y; w0[6] = r1.z; w0[7] = r1.w;
            }
            const float4 c0 = rp[2], c1 = rp[3];
            w1c[0] = c0.x; w1c[1] = c0.y; w1c[2] = c0.z; w1c[3] = c0.w;
            w1c[4] = c1.x; w1c[5] = c1.y; w1c[6] = c1.z; w1c[7] = c1.w;
        }
        const float t0 = W[20 * j + 16];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float acc = w0[0] * x[i][0];
#pragma unroll
            for (int g = 1; g < G; ++g) acc = fmaf(w0[g], x[i][g], acc);
            const float hj = fmaxf(acc + t0, 0.0f);
#pragma unroll
            for (int k = 0; k < 8; ++k) a1[i][k] = fmaf(w1c[k], hj, a1[i][k]);
        }
    }
    const float4 ta = reinterpret_cast<const float4*>(W + 320)[0], tb = reinterpret_cast<const float4*>(W + 320)[1];
    const float4 wa = reinterpret_cast<const float4*>(W + 328)[0], wb = reinterpret_cast<const float4*>(W + 328)[1];
    const float t1[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
    const float w2[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
    const float b2 = W[336];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        float acc = w2[0] * fmaxf(a1[i][0] + t1[0], 0.0f);
#pragma unroll
        for (int k = 1; k < 8; ++k) acc = fmaf(w2[k], fmaxf(a1[i][k] + t1[k], 0.0f), acc);
        out[i] = acc + b2;
    }
}

// The same network for NP PAIRS of items: the two items of a pair ride in the halves of v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32 (weights broadcast to both halves through op_sel), which halves the VALU instructions of the MLPs -- they are
// 24 % of the kernel's instructions at stage 1.  Every component goes through exactly the scalar version's operations in the
// same order (a packed fma is two IEEE fmas), so the results are bit-identical to mlp_from_lds.
template <int G, int NP>
__device__ __forceinline__ void mlp_pairs_from_lds(const float* __restrict__ W, const pmn_f2 (&x)[NP][G], pmn_f2 (&out)[NP]) {
    pmn_f2 a1[NP][8];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int k = 0; k < 8; ++k) a1[p][k] = pmn_f2{0.0f, 0.0f};
#pragma unroll 1
    for (int j = 0; j < 16; ++j) {
        const float4* rp = reinterpret_cast<const float4*>(W + 20 * j);
        float w0[8], w1c[8];
        {
            const float4 r0 = rp[0];
            w0[0] = r0.x; w0[1] = r0.y; w0[2] = r0.z; w0[3] = r0.w;
            if (G == 8) {
                const float4 r1 = rp[1];
                w0[4] = r1.x; w0[5] = r1.y; w0[6] = r1.z; w0[7] = r1.w;
            }
            const float4 c0 = rp[2], c1 = rp[3];
            w1c[0] = c0.x; w1c[1] = c0.y; w1c[2] = c0.z; w1c[3] = c0.w;
            w1c[4] = c1.x; w1c[5] = c1.y; w1c[6] = c1.z; w1c[7] = c1.w;
        }
        float t0 = W[20 * j + 16];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            pmn_f2 acc = pmn_f2{w0[0], w0[0]} * x[p][0];
#pragma unroll
            for (int g = 1; g < G; ++g) acc = __builtin_elementwise_fma(pmn_f2{w0[g], w0[g]}, x[p][g], acc);
            pmn_f2 hj = acc + pmn_f2{t0, t0};
            hj.x = fmaxf(hj.x, 0.0f);
            hj.y = fmaxf(hj.y, 0.0f);
#pragma unroll
            for (int k = 0; k < 8; ++k) a1[p][k] = __builtin_elementwise_fma(pmn_f2{w1c[k], w1c[k]}, hj, a1[p][k]);
        }
    }
    const float4 ta = reinterpret_cast<const float4*>(W + 320)[0], tb = reinterpret_cast<const float4*>(W + 320)[1];
    const float4 wa = reinterpret_cast<const float4*>(W + 328)[0], wb = reinterpret_cast<const float4*>(W + 328)[1];
    // (scalars out of two float4 loads, broadcast into packed math: lesson 46)
    const float t1[8] = {pmn_settle(ta.x), pmn_settle(ta.y), pmn_settle(ta.z), pmn_settle(ta.w), pmn_settle(tb.x), pmn_settle(tb.y),
                         pmn_settle(tb.z), pmn_settle(tb.w)};
    const float w2[8] = {pmn_settle(wa.x), pmn_settle(wa.y), pmn_settle(wa.z), pmn_settle(wa.w), pmn_settle(wb.x), pmn_settle(wb.y),
                         pmn_settle(wb.z), pmn_settle(wb.w)};
    const float b2 = pmn_settle(W[336]);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        pmn_f2 h = a1[p][0] + pmn_f2{t1[0], t1[0]};
        h.x = fmaxf(h.x, 0.0f);
        h.y = fmaxf(h.y, 0.0f);
        pmn_f2 acc = pmn_f2{w2[0], w2[0]} * h;
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            pmn_f2 hk = a1[p][k] + pmn_f2{t1[k], t1[k]};
            hk.x = fmaxf(hk.x, 0.0f);
            hk.y = fmaxf(hk.y, 0.0f);
            acc = __builtin_elementwise_fma(pmn_f2{w2[k], w2[k]}, hk, acc);
        }
        out[p] = acc + pmn_f2{b2, b2};
    }
}

// NIT items in chunks of NI
template <int G, int NIT, int NI>
__device__ __forceinline__ void mlp_items(const float* __restrict__ W, const float (&x)[NIT][G], float (&out)[NIT]) {
    static_assert(NIT % NI == 0, "chunk must divide the item count");
#pragma unroll
    for (int c = 0; c < NIT / NI; ++c) {
        if constexpr (NI % 2 == 0) {
            pmn_f2 xp[NI / 2][G], op[NI / 2];
#pragma unroll
            for (int p = 0; p < NI / 2; ++p)
#pragma unroll
                for (int g = 0; g < G; ++g) xp[p][g] = pmn_f2{x[c * NI + 2 * p][g], x[c * NI + 2 * p + 1][g]};
            mlp_pairs_from_lds<G, NI / 2>(W, xp, op);
#pragma unroll
            for (int p = 0; p < NI / 2; ++p) {
                out[c * NI + 2 * p] = op[p].x;
                out[c * NI + 2 * p + 1] = op[p].y;
            }
        } else {
            float xc[NI][G], oc[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int g = 0; g < G; ++g) xc[i][g] = x[c * NI + i][g];
            mlp_from_lds<G, NI>(W, xc, oc);
#pragma unroll
            for (int i = 0; i < NI; ++i) out[c * NI + i] = oc[i];
        }
    }
}

__device__ __forceinline__ float mul_add_unfused(float acc, float a, float b) {
#pragma clang fp contract(off)
    return acc + a * b;  // two roundings, like the reference's separate mul and add kernels
}


#ifdef PMN_EXPERIMENTAL  // `make EXPERIMENTAL=1`: the three LDS-window research families (csrc/experimental/, DESIGN.md lessons 17, 23)
// gather_win.hip: windowed implementation of MODE_VIEWS / MODE_PIXELWISE.  Returns PMN_ERR_SHAPE when the shape is not covered
// (the caller then uses the streaming kernel of gather_corr.hip).
int pmn_launch_gather_win(GatherArgs& a, int C, int G, bool pixelwise, hipStream_t stream);
// gather_lane.hip: lane = item, wave-autonomous implementation (the default family); same contract.
int pmn_launch_gather_lane(GatherArgs& a, int C, int G, bool pixelwise, hipStream_t stream);
int pmn_lane_set_tuning(int key, int value);
// gather_tile.hip: tile-window implementation of MODE_VIEWS (one LDS window per workgroup tile and view); same contract.
int pmn_launch_gather_tile(GatherArgs& a, int C, int G, hipStream_t stream);
int pmn_tile_set_tuning(int key, int value);
int pmn_gather_flags();  // pmn_set_tuning key 1
// corr_mfma.hip: correlate-then-interpolate on the fp32 matrix cores (round 4).  Returns PMN_ERR_SHAPE when the shape is not covered.
int pmn_launch_corr_mfma(GatherArgs& a, int C, int G, bool pixelwise, hipStream_t stream);
#endif
