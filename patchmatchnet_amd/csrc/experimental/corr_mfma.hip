// corr_mfma.hip -- pmn_warp_correlate as CORRELATE-THEN-INTERPOLATE on the fp32 matrix cores (round 4).
//
// RESEARCH BUILD ONLY (`make EXPERIMENTAL=1`, pmn_set_tuning key 1 bit 6): parity-green against the streaming kernel and the
// golden tensors (tests/test_corr_mfma.py), and measured SLOWER than it on every launch of the cascade on the same box
// (1517 vs 1108 us per depth map, profiles/r04_corr_mfma.md): the formulation moves the bound from the vector L1 to VALU
// issue -- projecting every (pixel, hypothesis, view) and managing the data-dependent windows costs as many VALU
// instructions per item as the streaming kernel's blend, and the MFMA pipe sits at 11-17 %.
//
// Reference: models/patchmatch.py:192-217 + :570 (Evaluation / SimilarityNet MLP), :695-702 (PixelwiseNet);
// models/module.py:130-181 (differentiable_warping).  The reference warps first (bilinear blend of C channels at every
// (pixel, hypothesis, view)) and correlates second.  Bilinear interpolation is linear, so it commutes with the group dot product:
//
//     sum_{c in g} ref[p,c] * ( sum_k w_k * src[q_k,c] )  =  sum_k w_k * ( sum_{c in g} ref[p,c] * src[q_k,c] )
//                                                          =  sum_k w_k * R[p, q_k, g]
//
// R[p,q,g] is a DENSE contraction between the pixels of a tile and the texels of the source-map window their taps fall into:
// it runs on v_mfma_f32_16x16x4_f32 (exact fp32, K = 4 = one correlation group's channels per k-step), from ONE coalesced read
// of the window, and every (pixel, hypothesis) then gathers 4 taps x G floats from LDS instead of 4 taps x C floats from global
// memory.  Against the streaming kernel (gather_corr.hip) the gathered bytes fall by C/G, the source bytes through the vector
// L1 by the window re-use (5-30x), and the VALU work per item from ~5 C flops to 4 G FMAs.
//
// Mapping (wave64; a wave owns one UNIT = 16 pixels of one image row x 8 consecutive hypotheses and runs on its own -- with known
// view weights there is no workgroup barrier after the prologue; with PixelwiseNet the D/8 waves of a pixel tile form one
// workgroup and meet once per view for the max over D):
//   * lane = (n = lane & 15, k = lane >> 4); lane (n, k) owns the two hypotheses d = 8 chunk + 2 k + {0, 1} of pixel n (its
//     `items`, carried in the halves of packed registers).  Hypotheses are sorted along d, so a chunk's taps sit on a short
//     piece of the epipolar line; tiles never straddle image rows.
//   * per view: every lane projects its items (same arithmetic as the streaming kernel: pmn_pose_position); a
//     wave reduction gives the bounding box of the live taps = the window (Wd x Hd texels, flattened row-major: texel q).
//   * software pipeline: the window of the NEXT fill (next pass / next view: projected first) is requested before the
//     current one is consumed, so its L2 latency hides behind the MFMAs and the gather of the current one.
//   * R for 4 groups at a time (`pass`; G = 8 takes two): for every N-tile t of 16 texels lane (n, k) loads channels
//     [16 jb + 4 k, +4) of texel 16 t + n (one dwordx4: the wave reads 16 texels x 64 B), a 4x4 transpose across the four 16-lane rows
//     (v_permlane32_swap + v_permlane16_swap) turns that into the MFMA operand layout (lane (n, k) <-> channel 4 g' + k), and
//     D[texel][pixel] += A[texel][k] * B[k][pixel]  with A = source, B = reference fragments (loaded and transposed once per
//     tile).  Lane (n, k) ends up with texels 16 t + 4 k .. +3 of pixel n: ONE ds_write_b128 per group into R[n][g][q].
//   * gather: item (n, d) reads R[n][g][q00 + {0, 1, Wd, Wd + 1}], blends with its 4 corner weights, and the per-view group
//     similarity goes into the view sum (known weights) or through PixelwiseNet (first iteration of the coarsest stage).
//   * windows larger than the wave's LDS buffer (QP texels) are walked in rectangular pieces with per-tap predicates
//     (correct for any geometry, slower; the sums of such an item are taken in piece order).
//   * epilogue as the streaming kernel: view normalisation, SimilarityNet MLP from LDS-staged weights, hypothesis-last cost.
// Numerics: same tap positions and weights as the streaming kernel; the channel sum and the 4-tap blend are re-associated
// (the MFMA is an exact fp32 fmaf chain over the group's channels), a rounding-level (1e-7 relative) difference.
#include <type_traits>

#include "../gather_common.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef PMN_CM_ABL
#define PMN_CM_ABL 0  // development ablations (wrong results): 1 = every window at texel 0 (cache-hot loads), 2 = no MFMA / LDS stores,
#endif                //   4 = no LDS gather, 8 = no window loads at all
#define CM_GP 2           // groups per sub-pass: a wave's R block is 16 px x 2 groups x QP texels
#define CM_MAX_VIEWS 16  // source views whose projections a workgroup stages in LDS (more: the streaming kernel)

// 4x4 transpose across the four 16-lane rows of a wave.  In: lane (n, k) register i holds E[k][i]; out: E[i][k].
__device__ __forceinline__ void pmn_row_transpose4(float (&v)[4]) {
#ifndef PMN_TRANSPOSE_SHFL
    // v_permlane32_swap a, b: a.lanes[32..63] <-> b.lanes[0..31];  v_permlane16_swap a, b: odd rows of a <-> even rows of b
    const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[2]), false, false);
    const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[1]), __float_as_uint(v[3]), false, false);
    const auto t01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
    const auto t23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
    v[0] = __uint_as_float(t01[0]);
    v[1] = __uint_as_float(t01[1]);
    v[2] = __uint_as_float(t23[0]);
    v[3] = __uint_as_float(t23[1]);
#else  // ds_bpermute form (build variant for checking the permlane form on hardware)
    const int lane = threadIdx.x & 63, n = lane & 15, k = lane >> 4;
    float o[4];
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) {
        float got = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x = __shfl(v[i], n + 16 * kp, 64);
            got = (i == k) ? x : got;
        }
        o[kp] = got;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = o[i];
#endif
}

// wave-wide min / max of a 32-bit int: four DPP steps inside the 16-lane rows, then the four rows through SGPRs
template <bool IS_MAX>
__device__ __forceinline__ int pmn_wave_minmax(int v) {
    auto op = [](int a, int b) { return IS_MAX ? max(a, b) : min(a, b); };
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));  // row_half_mirror
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));  // row_mirror
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    return op(op(r0, r1), op(r2, r3));
}

__device__ __forceinline__ pmn_f2 mul_add_unfused2(pmn_f2 acc, pmn_f2 a, float b) {
#pragma clang fp contract(off)
    return acc + a * pmn_f2{b, b};  // two roundings per half, like the reference's separate mul and add kernels
}

// lanes of one wave hand data to each other through LDS: the LDS queue is in order, the compiler must be too
__device__ __forceinline__ void pmn_wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The four wave-wide reductions of the window in one block: v_min/v_max with the DPP operand folded in (hipcc emits v_mov_dpp +
// v_min for the builtin form), the four chains interleaved so that no instruction reads a register written less than three
// instructions earlier (a DPP read needs two wait states after a VALU write), the four rows combined with row_bcast:15 / :31 --
// 24 VALU instructions + 4 v_readlane instead of ~64 + s_nops.  All 64 lanes must be active.
__device__ __forceinline__ void cm_wave_bbox(int& lox, int& hix, int& loy, int& hiy) {
#define CM_STEP(ctrl)                                                             \
    "v_min_i32_dpp %0, %0, %0 " ctrl "\n\t"                                       \
    "v_max_i32_dpp %1, %1, %1 " ctrl "\n\t"                                       \
    "v_min_i32_dpp %2, %2, %2 " ctrl "\n\t"                                       \
    "v_max_i32_dpp %3, %3, %3 " ctrl "\n\t"
    asm volatile("s_nop 1\n\t"
                 CM_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 CM_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 CM_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
                 CM_STEP("row_mirror row_mask:0xf bank_mask:0xf")
                 CM_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 CM_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 1"
                 : "+v"(lox), "+v"(hix), "+v"(loy), "+v"(hiy));
#undef CM_STEP
    lox = __builtin_amdgcn_readlane(lox, 63);
    hix = __builtin_amdgcn_readlane(hix, 63);
    loy = __builtin_amdgcn_readlane(loy, 63);
    hiy = __builtin_amdgcn_readlane(hiy, 63);
}

// pmn_make_taps_xy with its common case first: both corners of both axes addressable (pmn_axis: i0 == i0c), where the general
// rule's clamps and selects are no-ops -- the same operations on the same values, so the same bits.
__device__ __forceinline__ PmnTapsXY cm_taps_xy(float ix, float iy, int hs, int ws) {
#pragma clang fp contract(off)
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;  // (v_cvt saturates, NaN -> 0: such an item ends up with NaN weights = not live)
    if ((unsigned)x0 <= (unsigned)(ws - 2) && (unsigned)y0 <= (unsigned)(hs - 2)) {
        const float ax = (fx + 1.0f) - ix, bx = ix - fx, ay = (fy + 1.0f) - iy, by = iy - fy;
        PmnTapsXY t;
        t.x0 = x0; t.y0 = y0;
        t.w00 = ax * ay; t.w01 = bx * ay; t.w10 = ax * by; t.w11 = bx * by;
        return t;
    }
    return pmn_make_taps_xy(ix, iy, hs, ws);
}

// one view of a unit: the projected items and the window of their taps
struct CmTask {
    int x0[2], y0[2];
    float w00[2], w01[2], w10[2], w11[2];
    bool live[2];
    int xmin, ymin, Wd, Hd;  // wave-uniform
    bool any;                // wave-uniform: some item has a tap inside the source map
    float vw;                // view weight of the lane's pixel (known-weights launches)
};

// C channels, G groups, QP = texel capacity of a wave's LDS window, NW units (waves) per workgroup of a known-weights launch,
// PIXELWISE = view weights computed here by PixelwiseNet (workgroup = the ceil(D/8) units of one pixel tile)
template <int C, int G, int QP, int NW, bool PIXELWISE>
__global__ __launch_bounds__(PIXELWISE ? 512 : 64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void corr_mfma_kernel(const GatherArgs a) {
    constexpr int CG = C / G;            // channels per group: 4 or 8
    constexpr int KS = CG / 4;           // k-steps per group
    constexpr int NP16 = C / 16;         // passes per view: one 16-channel block of the features each (one dwordx4 per lane and N-tile)
    constexpr int GPB = 16 / CG;         // groups per pass: 4 or 2
    constexpr int GP = CM_GP;            // groups per sub-pass: R holds GP groups of the window at a time
    constexpr int NSUB = GPB / GP;       // sub-passes per pass (the pass's texels stay in registers across them)
    constexpr int PP = GP * QP + 4;      // floats per pixel block of R (the +4 spreads the pixels' ds_write_b128 over the banks)
    constexpr int TMAX = QP / 16;
    static_assert(CG == 4 || CG == 8, "group size must be 4 or 8 channels");
    static_assert(G % 4 == 0 && QP % 16 == 0 && QP <= 256 && NSUB >= 1 && NSUB <= 2 && (NP16 == 1 || NP16 % 2 == 0), "shape");

    extern __shared__ float4 smem4[];
    float* wlds_a = reinterpret_cast<float*>(smem4);                 // SimilarityNet
    float* wlds_b = wlds_a + MLP_LDS_FLOATS;                         // PixelwiseNet            (PIXELWISE)
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(wlds_b + MLP_LDS_FLOATS);  // [2][8][16]  (PIXELWISE)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float* plds = wlds_a + (PIXELWISE ? 2 * MLP_LDS_FLOATS + 2 * 8 * 16 * 2 : MLP_LDS_FLOATS);  // [CM_MAX_VIEWS][16] relative projections
    float* R = plds + CM_MAX_VIEWS * 16 + wave * (16 * PP);
    const int n = lane & 15, k = lane >> 4;
    const int nthreads = PIXELWISE ? (int)blockDim.x : 64 * NW;

    const int b = blockIdx.y;
    for (int i = tid; i < PMN_MLP_FLOATS; i += nthreads) {
        wlds_a[i] = a.mlp_a[i];
        if (PIXELWISE) wlds_b[i] = a.mlp_b[i];
    }
    // the views' projections: a per-view global load would sit at the head of every view's dependency chain
    for (int i = tid; i < a.N * 16; i += nthreads) plds[i] = a.proj[(size_t)b * a.N * 16 + i];
    __syncthreads();

    const int D = a.D, N = a.N, h = a.h, w = a.w, hs = a.hs, ws = a.ws;
    const int hw = h * w;
    const int nch = (D + 7) >> 3;
    const int tpr = (w + 15) >> 4;       // tiles per image row
    int tile, chunk;
    if (PIXELWISE) {
        tile = pmn_xcd_tile(blockIdx.x, a.ntiles);
        chunk = wave;
    } else {
        const int unit = pmn_xcd_tile(blockIdx.x, a.ntiles) * NW + wave;
        if (unit >= tpr * h * nch) return;  // (whole wave; no barrier follows on this path)
        tile = unit / nch;
        chunk = unit - tile * nch;
    }
    const int y = tile / tpr;
    const int x = (tile - y * tpr) * 16 + n;
    const bool ok = x < w;
    const int p = y * w + min(x, w - 1);
    const float xf = (float)x, yf = (float)y;

    // reference fragments: refT[jb][i] = channel 16 jb + 4 i + k of pixel n
    float refT[C / 16][4];
#pragma unroll
    for (int jb = 0; jb < C / 16; ++jb) {
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) f = reinterpret_cast<const float4*>(a.ref)[((size_t)b * hw + p) * (C / 4) + jb * 4 + k];
        refT[jb][0] = f.x; refT[jb][1] = f.y; refT[jb][2] = f.z; refT[jb][3] = f.w;
        pmn_row_transpose4(refT[jb]);
    }
    // the lane's two hypotheses
    const int d0 = 8 * chunk + 2 * k;
    bool dok[2];
    float dep[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        dok[i] = ok && (d0 + i < D);
        dep[i] = dok[i] ? a.depth[((size_t)b * D + d0 + i) * hw + p] : 0.0f;
    }

    const int wv = w >> a.vw_shift, hwv = (h >> a.vw_shift) * wv;
    const int vw_idx = (y >> a.vw_shift) * wv + (min(x, w - 1) >> a.vw_shift);
    const unsigned pixbase = (unsigned)n * PP;

    // ---- projection of the lane's items into view v + the window of the wave's live taps ---------------------------------
    auto prepare = [&](CmTask& T, const int v) __attribute__((always_inline)) {
        const PmnPose pose = pmn_make_pose(plds + v * 16, xf, yf, h, w);  // the reference's own warp chain (pmn_common.hpp)
        int lox = 0x7fffffff, loy = 0x7fffffff, hix = -0x7fffffff, hiy = -0x7fffffff;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            PmnTapsXY t;
            t.x0 = t.y0 = 0;
            t.w00 = t.w01 = t.w10 = t.w11 = 0.0f;
            float ix, iy;
            // behind-camera hypotheses sample nothing (reference sentinel, module.py:166-169)
            if (pmn_pose_position(pose, dep[i], h, w, hs, ws, ix, iy) && dok[i]) t = cm_taps_xy(ix, iy, hs, ws);
            T.x0[i] = t.x0; T.y0[i] = t.y0;
            T.w00[i] = t.w00; T.w01[i] = t.w01; T.w10[i] = t.w10; T.w11[i] = t.w11;
            T.live[i] = (t.w00 + t.w01) + (t.w10 + t.w11) > 0.0f;  // items without an in-range corner stay out of the window
            if (T.live[i]) {
                lox = min(lox, t.x0); hix = max(hix, t.x0);
                loy = min(loy, t.y0); hiy = max(hiy, t.y0);
            }
        }
        cm_wave_bbox(lox, hix, loy, hiy);
        const int xmin = lox, xmax = hix, ymin = loy, ymax = hiy;
        T.any = xmin <= xmax;
        T.xmin = T.any ? xmin : 0;
        T.ymin = T.any ? ymin : 0;
        T.Wd = T.any ? xmax - xmin + 2 : 2;
        T.Hd = T.any ? ymax - ymin + 2 : 2;
        T.vw = 0.0f;
        if (!PIXELWISE) T.vw = ok ? a.vw_in[((size_t)b * N + v) * hwv + vw_idx] : 0.0f;
    };

    // ---- request the source texels of the window piece [ox, ox+cw) x [oy, oy+ch) (window coordinates), pass ps ---------------
    auto request = [&](const CmTask& T, const int v, auto psc, const int ox, const int oy, const int cw, const int ch,
                       float4 (&raw)[TMAX]) __attribute__((always_inline)) {
        constexpr int ps = decltype(psc)::value;
        const char* sbase = reinterpret_cast<const char*>(a.src) + ((size_t)(v * a.B + b) * hs * ws) * (C * 4);
        const int Qc = cw * ch;
        // q -> (qy, qx) = divmod(q, cw) by a 20-bit reciprocal: exact for q < 1024, cw <= 512 (QP <= 256 here)
        const unsigned magic = (1u << 20) / (unsigned)cw + 1u;
        const unsigned tex0 = (PMN_CM_ABL & 1) ? 0u : (unsigned)((T.ymin + oy) * ws + (T.xmin + ox));  // first texel of the piece (wave-uniform)
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            const unsigned q = (unsigned)min(16 * t + n, Qc - 1);
            const unsigned qy = __umul24(q, magic) >> 20;
            const unsigned qx = q - __umul24(qy, (unsigned)cw);
            const unsigned tex = tex0 + __umul24(qy, (unsigned)ws) + qx;
            const unsigned bo = tex * (C * 4u) + (ps * 16 + 4 * k) * 4u;
            if (PMN_CM_ABL & 8) raw[t] = make_float4((float)bo, 1.0f, 2.0f, (float)t);
            else raw[t] = *reinterpret_cast<const float4*>(sbase + bo);
        }
    };
    // ---- R of a requested piece (Qc texels) for the four groups of pass ps: transposes, MFMAs, LDS stores -------------------
    auto fill = [&](auto psc, auto subc, const int Qc, float4 (&raw)[TMAX]) __attribute__((always_inline)) {
        constexpr int ps = decltype(psc)::value, sub = decltype(subc)::value;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            if (16 * t < Qc && !(PMN_CM_ABL & 2)) {
                if constexpr (sub == 0) {  // the first sub-pass turns the loaded texels into MFMA operands, in place
                    float e[4] = {raw[t].x, raw[t].y, raw[t].z, raw[t].w};
                    pmn_row_transpose4(e);
                    raw[t] = make_float4(e[0], e[1], e[2], e[3]);
                }
                f32x4 dd[GP];
#pragma unroll
                for (int gq = 0; gq < GP; ++gq) {
                    dd[gq] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        const int e = ((sub * GP + gq) * CG + 4 * s) / 4;  // element of the lane's dwordx4 = channel quad of the block
                        const float sv_ = e == 0 ? raw[t].x : e == 1 ? raw[t].y : e == 2 ? raw[t].z : raw[t].w;
                        dd[gq] = __builtin_amdgcn_mfma_f32_16x16x4f32(sv_, refT[ps][e], dd[gq], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int gq = 0; gq < GP; ++gq) *reinterpret_cast<f32x4*>(R + pixbase + gq * QP + 16 * t + 4 * k) = dd[gq];
            }
        }
    };

    // ---- one pass over one piece [ox, ox+cw) x [oy, oy+ch) of a view's window: sv[ps*GPB ..] += the blended group correlations ---------
    auto compute = [&](const CmTask& T, auto psc, const int ox, const int oy, const int cw, const int ch, const bool whole,
                       float4 (&raw)[TMAX], pmn_f2 (&sv)[G]) __attribute__((always_inline)) {
        constexpr int ps = decltype(psc)::value;
        if (!T.any) return;
        auto sub_pass = [&](auto subc) __attribute__((always_inline)) {
            constexpr int sub = decltype(subc)::value;
            fill(psc, subc, cw * ch, raw);
            pmn_wave_lds_fence();
            if (whole) {
                // the piece is the whole window: every tap of every live item is inside, fixed nw, ne, sw, se blend order
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (T.live[i] && !(PMN_CM_ABL & 4)) {
                        const float* rp0 = R + pixbase + (unsigned)((T.y0[i] - T.ymin) * cw + (T.x0[i] - T.xmin));
#pragma unroll
                        for (int gq = 0; gq < GP; ++gq) {
                            const float* rp = rp0 + gq * QP;
                            const float r00 = rp[0], r01 = rp[1], r10 = rp[cw], r11 = rp[cw + 1];
                            sv[ps * GPB + sub * GP + gq][i] = fmaf(r11, T.w11[i], fmaf(r10, T.w10[i], fmaf(r01, T.w01[i], r00 * T.w00[i])));
                        }
                    }
                }
            } else {
                // a window larger than the buffer is walked in pieces (rare): every tap is taken in the piece it falls into
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ux = T.x0[i] - T.xmin - ox, uy = T.y0[i] - T.ymin - oy;
#pragma unroll
                    for (int tp = 0; tp < 4; ++tp) {
                        const int qx = ux + (tp & 1), qy = uy + (tp >> 1);
                        const float wt = tp == 0 ? T.w00[i] : tp == 1 ? T.w01[i] : tp == 2 ? T.w10[i] : T.w11[i];
                        if (T.live[i] && (unsigned)qx < (unsigned)cw && (unsigned)qy < (unsigned)ch) {
                            const float* rp = R + pixbase + (unsigned)(qy * cw + qx);
#pragma unroll
                            for (int gq = 0; gq < GP; ++gq)
                                sv[ps * GPB + sub * GP + gq][i] = fmaf(rp[gq * QP], wt, sv[ps * GPB + sub * GP + gq][i]);
                        }
                    }
                }
            }
            pmn_wave_lds_fence();
        };
        sub_pass(std::integral_constant<int, 0>{});
        if constexpr (NSUB > 1) sub_pass(std::integral_constant<int, 1>{});
    };

    // items (d0, d0 + 1) of the lane ride in the halves of packed registers: the view sum and the MLPs run on v_pk_* as they are
    pmn_f2 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = pmn_f2{0.0f, 0.0f};
    float wsum = 1e-5f;

    // ---- end of a view: the view sum (known weights), or PixelwiseNet + max over D + the view sum --------------------------------
    auto finish = [&](const CmTask& T, const int v, pmn_f2 (&sv)[G]) __attribute__((always_inline)) {
        if constexpr (!PIXELWISE) {
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = mul_add_unfused2(acc[g], sv[g] * (1.0f / CG), T.vw);
            wsum += T.vw;
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) sv[g] = sv[g] * (1.0f / CG);
            pmn_f2 xq[1][G], rq[1];
#pragma unroll
            for (int g = 0; g < G; ++g) xq[0][g] = sv[g];
            mlp_pairs_from_lds<G, 1>(wlds_b, xq, rq);
            // max over D, first arg-max on ties through the ~d low word; hypotheses past D do not take part
            unsigned long long best = 0ull;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned long long key = ((unsigned long long)__float_as_uint(pmn_sigmoid(rq[0][i])) << 32) |
                                               (unsigned long long)(0xFFFFFFFFu - (unsigned)(d0 + i));
                if (d0 + i < D) best = key > best ? key : best;
            }
            {   // the pixel's four lanes (n, 0..3), then the pixel's other chunks through LDS (double-buffered by view parity:
                // one barrier per view)
                unsigned long long o = __shfl_xor(best, 16, 64);
                best = o > best ? o : best;
                o = __shfl_xor(best, 32, 64);
                best = o > best ? o : best;
            }
            unsigned long long* kb = keys + (v & 1) * (8 * 16);
            if (k == 0) kb[chunk * 16 + n] = best;
            __syncthreads();
            for (int c = 0; c < nch; ++c) {
                const unsigned long long o = kb[c * 16 + n];
                best = o > best ? o : best;
            }
            const float vwp = __uint_as_float((unsigned)(best >> 32));
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = mul_add_unfused2(acc[g], sv[g], vwp);
            wsum += vwp;
            if (ok && k == 0 && chunk == 0) {
                const size_t o = ((size_t)b * N + v) * hw + p;
                a.vw_out[o] = vwp;
                if (a.vw_argmax) a.vw_argmax[o] = (int)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull));
            }
        }
    };

    // ---- the views, software-pipelined: the next fill's texels are requested before the current one is consumed ----------------
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    // Fill units run in the order view -> piece of its window (one, unless the window is larger than the buffer) -> pass; the texels of
    // unit u + 1 are requested into the other buffer before unit u is consumed, so their L2 / HBM latency hides behind unit u's MFMAs
    // and gather.  With an even pass count the buffers alternate with the pass; with one pass the next unit's texels are moved over.
    CmTask cur, nxt;
    float4 rawA[TMAX], rawB[TMAX];
    auto piece_dims = [&](const CmTask& T, int& cwm, int& chm, int& npx, int& npy) __attribute__((always_inline)) {
        cwm = min(T.Wd, QP);
        chm = min(T.Hd, QP / cwm);
        npx = (T.Wd + cwm - 1) / cwm;
        npy = (T.Hd + chm - 1) / chm;
        if (PMN_CM_ABL & 16) npx = npy = 1;
    };
    prepare(cur, 0);
    {
        int cwm, chm, npx, npy;
        piece_dims(cur, cwm, chm, npx, npy);
        request(cur, 0, P0{}, 0, 0, cwm, chm, rawA);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int v = 0; v < N; ++v) {
        pmn_f2 sv[G];
#pragma unroll
        for (int g = 0; g < G; ++g) sv[g] = pmn_f2{0.0f, 0.0f};
        int cwm, chm, npx, npy;
        piece_dims(cur, cwm, chm, npx, npy);
        const bool whole = npx * npy == 1;
        for (int py = 0; py < npy; ++py) {
            const int oy = py * chm, ch = min(chm, cur.Hd - oy);
            for (int px = 0; px < npx; ++px) {
                const int ox = px * cwm, cw = min(cwm, cur.Wd - ox);
                const bool last_piece = py == npy - 1 && px == npx - 1;
                // what follows this piece's last pass: the next piece of the view, or the first piece of the next view
                auto request_after = [&](float4 (&rawn)[TMAX]) __attribute__((always_inline)) {
                    if (!last_piece) {
                        const int nx = px + 1 < npx ? px + 1 : 0, ny = px + 1 < npx ? py : py + 1;
                        request(cur, v, P0{}, nx * cwm, ny * chm, min(cwm, cur.Wd - nx * cwm), min(chm, cur.Hd - ny * chm), rawn);
                    } else if (v + 1 < N) {
                        prepare(nxt, v + 1);
                        int cwn, chn, nxn, nyn;
                        piece_dims(nxt, cwn, chn, nxn, nyn);
                        request(nxt, v + 1, P0{}, 0, 0, cwn, chn, rawn);
                    }
                    __builtin_amdgcn_sched_barrier(0);  // keep the loads up here: hipcc otherwise sinks them to their first use
                };
                auto pass_step = [&](auto psc, float4 (&rawc)[TMAX], float4 (&rawn)[TMAX]) __attribute__((always_inline)) {
                    constexpr int ps = decltype(psc)::value;
                    if constexpr (ps + 1 < NP16) {
                        request(cur, v, std::integral_constant<int, ps + 1>{}, ox, oy, cw, ch, rawn);
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        request_after(rawn);
                    }
                    compute(cur, psc, ox, oy, cw, ch, whole, rawc, sv);
                };
                pass_step(P0{}, rawA, rawB);
                if constexpr (NP16 >= 2) pass_step(P1{}, rawB, rawA);
                if constexpr (NP16 >= 4) {
                    pass_step(std::integral_constant<int, 2>{}, rawA, rawB);
                    pass_step(std::integral_constant<int, 3>{}, rawB, rawA);
                }
                if constexpr (NP16 == 1) {
#pragma unroll
                    for (int t = 0; t < TMAX; ++t) rawA[t] = rawB[t];
                }
            }
        }
        finish(cur, v, sv);
        cur = nxt;
    }

    // ---- epilogue: view normalisation, SimilarityNet MLP, hypothesis-last cost ---------------------------------------------------------
    if (!ok) return;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        acc[g].x = acc[g].x / wsum;
        acc[g].y = acc[g].y / wsum;
    }
    pmn_f2 xq[1][G], oq[1];
#pragma unroll
    for (int g = 0; g < G; ++g) xq[0][g] = acc[g];
    mlp_pairs_from_lds<G, 1>(wlds_a, xq, oq);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int d = d0 + i;
        if (d < D) {
            if (a.sim_out) {
#pragma unroll
                for (int g = 0; g < G; ++g) a.sim_out[(((size_t)b * G + g) * D + d) * hw + p] = acc[g][i];
            }
            a.out[((size_t)b * hw + p) * D + d] = oq[0][i];  // cost is hypothesis-last [B,h,w,D]
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------

template <int C, int G, int QP, int NW, bool PIXELWISE>
static int launch_corr_mfma(GatherArgs& a, hipStream_t stream) {
    constexpr int PP = CM_GP * QP + 4;
    const int nch = (a.D + 7) / 8;
    const int tiles = ((a.w + 15) / 16) * a.h;
    auto kern = corr_mfma_kernel<C, G, QP, NW, PIXELWISE>;
    size_t lds;
    int threads;
    if (PIXELWISE) {
        a.ntiles = tiles;  // one workgroup per pixel tile: its ceil(D/8) units
        threads = 64 * nch;
        lds = (size_t)(2 * MLP_LDS_FLOATS + 2 * 8 * 16 * 2 + CM_MAX_VIEWS * 16 + nch * 16 * PP) * 4;
    } else {
        a.ntiles = (tiles * nch + NW - 1) / NW;
        threads = 64 * NW;
        lds = (size_t)(MLP_LDS_FLOATS + CM_MAX_VIEWS * 16 + NW * 16 * PP) * 4;
    }
    if (lds > 160 * 1024) return PMN_ERR_SHAPE;
    if (lds > 48 * 1024) {
        const int rc = pmn_raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
        if (rc != PMN_OK) return rc;
    }
    PMN_LAUNCH(kern, dim3(a.ntiles, a.B), dim3(threads), lds, stream, a);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

#ifndef PMN_CM_QP
#define PMN_CM_QP 128     // window capacity of a known-weights unit at 16 / 32 channels
#endif
#ifndef PMN_CM_QP64
#define PMN_CM_QP64 128   // ... at 64 channels
#endif
#ifndef PMN_CM_QPX
#define PMN_CM_QPX 128    // ... of a PixelwiseNet unit (eight of them share a workgroup's LDS: 8 x 16.6 KB)
#endif
#ifndef PMN_CM_NW
#define PMN_CM_NW 1
#endif

// Every hypothesis count up to PMN_MAX_DEPTH at the cascade's three (C, G) pairs; anything else returns PMN_ERR_SHAPE and the
// caller takes the streaming kernel.
int pmn_launch_corr_mfma(GatherArgs& a, int C, int G, bool pixelwise, hipStream_t stream) {
    if (a.D < 1 || a.D > 64 || a.N > CM_MAX_VIEWS) return PMN_ERR_SHAPE;
    if ((size_t)a.hs * a.ws * C * 4 >= (1ull << 32) || a.hs * a.ws >= (1 << 24) || a.ws >= (1 << 16)) return PMN_ERR_SHAPE;
    if (!pixelwise) {
        if (C == 16 && G == 4) return launch_corr_mfma<16, 4, PMN_CM_QP, PMN_CM_NW, false>(a, stream);
        if (C == 32 && G == 8) return launch_corr_mfma<32, 8, PMN_CM_QP, PMN_CM_NW, false>(a, stream);
        if (C == 64 && G == 8) return launch_corr_mfma<64, 8, PMN_CM_QP64, PMN_CM_NW, false>(a, stream);
        return PMN_ERR_SHAPE;
    }
    if (C == 16 && G == 4) return launch_corr_mfma<16, 4, PMN_CM_QPX, 1, true>(a, stream);
    if (C == 32 && G == 8) return launch_corr_mfma<32, 8, PMN_CM_QPX, 1, true>(a, stream);
    if (C == 64 && G == 8) return launch_corr_mfma<64, 8, PMN_CM_QPX, 1, true>(a, stream);
    return PMN_ERR_SHAPE;
}
