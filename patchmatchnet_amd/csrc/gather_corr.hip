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
//   * a workgroup owns a tile of NPIX = 256/LPI consecutive pixels x all D hypotheses and alternates two roles:
//       lane role   LPI lanes <-> pixel.  Lane lc projects hypotheses lc, lc + LPI, ... of ITS OWN pixel and keeps their tap records
//                   {texel offset, 4 corner weights} in registers; the group then walks the pixel's hypotheses: DPP broadcast of
//                   the record, 4 x global_load_dwordx4, bilinear blend, product with the register-resident reference quad,
//                   in-lane + one DPP step group reduction.  No LDS and no barrier inside the loop (all three modes since round 6:
//                   the FeatureWeightNet launches parked their records in LDS behind two barriers until then).  With known view
//                   weights the per-(pixel,group,d) sums over views accumulate in REGISTERS of the owning lane (the d loop is
//                   fully unrolled over the compile-time bound DT);
//       item role   thread <-> (pixel, a few hypotheses): after ONE hand-over through LDS (similarity tile + barrier) runs the
//                   pointwise MLPs on its items (weights broadcast from LDS, each weight row reused for all the thread's items)
//                   and stores cost[pixel][d] (hypothesis-last, what pmn_aggregate_regress gathers) / the feature weights; with
//                   PixelwiseNet each view's tile goes through LDS to the item role, which evaluates the net, takes the max over D
//                   and keeps the weighted sums in its own registers (the cascade's stage-3 launch is the wave-private
//                   pixelwise_wave_kernel below instead).
//     The [C,D,h,w] warped volume and the per-view [G,D,h,w] similarity never touch HBM.
//   * no MFMA: ~10 flop per gathered float, no dense contraction worth a matrix core (the MLPs are 16x8 / 8x16).
#include <type_traits>

#include "gather_common.hpp"


// Broadcast `v` from lane SL of every aligned group of LPI lanes (SL compile-time): DPP quad_perm for 4-lane groups,
// DPP row_newbcast for 16-lane groups (one VALU op, no LDS), ds_bpermute for 8-lane groups.
template <int LPI, int SL>
__device__ __forceinline__ int group_bcast_i(int v) {
    // mov_dpp (bound_ctrl set): one instruction; update_dpp with an `old` operand costs a v_mov to initialise the destination
    if constexpr (LPI == 4) {
        return __builtin_amdgcn_mov_dpp(v, SL | (SL << 2) | (SL << 4) | (SL << 6), 0xF, 0xF, true);
    } else if constexpr (LPI == 16) {
        return __builtin_amdgcn_mov_dpp(v, 0x150 + SL, 0xF, 0xF, true);
    } else {
        // 8-lane groups = the two halves of a 16-lane DPP row: row_newbcast of lane SL into banks 0-1 (lanes 0..7 of the
        // row), of lane 8+SL into banks 2-3 (lanes 8..15) -- two VALU ops instead of a ds_bpermute LDS round trip
        const int lo = __builtin_amdgcn_update_dpp(v, v, 0x150 + SL, 0xF, 0x3, false);
        return __builtin_amdgcn_update_dpp(lo, v, 0x150 + 8 + SL, 0xF, 0xC, false);
    }
}
template <int LPI, int SL>
__device__ __forceinline__ float group_bcast_f(float v) {
    return __int_as_float(group_bcast_i<LPI, SL>(__float_as_int(v)));
}

// Source map of view v, batch element b: a slice of the stacked [N,B,hs,ws,C] tensor, or -- pmn_warp_correlate_views -- the v-th
// entry of a device table of per-view addresses (the maps then stay where their producer left them: eval.py's per-scan feature
// cache hands a sample its views without copying 0.5 GB of pyramids into a stacked buffer).  Wave-uniform.
template <int C>
__device__ __forceinline__ const char* pmn_view_base(const GatherArgs& a, int v, int b, int hs, int ws) {
    if (a.src_tab) return reinterpret_cast<const char*>(a.src_tab[v]) + ((size_t)b * hs * ws) * (C * 4);
    return reinterpret_cast<const char*>(a.src) + ((size_t)(v * a.B + b) * hs * ws) * (C * 4);
}

// One (pixel, hypothesis) item of the lane role in two halves, so a batch of items can have ALL its corner loads in flight before the first blend (left to
// itself hipcc schedules load -> wait -> blend item by item).  `sbase` is the wave-uniform base of the view's map (SGPR pair),
// `bo` the 32-bit byte offset of this lane's channel quad of the north-west texel: the loads use the SGPR-base + VGPR-offset
// form, east corner through the immediate offset field (no 64-bit VALU address arithmetic).
struct PmnCorners { float4 t00, t01, t10, t11; };

// (The view's base may come out of a device table of addresses -- an integer turned pointer, whose address space hipcc cannot know:
// it then emits flat_load_dwordx4 on a 64-bit VGPR address built with a v_lshl_add_u64 per corner pair, and a FLAT access also
// occupies the LDS queue and both wait counters.  The maps are global memory: the explicit address space gives
// global_load_dwordx4 v, v_offset, s[base] offset:imm.)
typedef float pmn_f4n __attribute__((ext_vector_type(4)));  // (a native vector: HIP's float4 class cannot be read through an address-space pointer)
typedef const pmn_f4n __attribute__((address_space(1))) * pmn_gptr4;
__device__ __forceinline__ float4 pmn_global_load4(const char __attribute__((address_space(1))) * p) {
    const pmn_f4n v = *(pmn_gptr4)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
template <int C>
__device__ __forceinline__ PmnCorners load_corners(const char* __restrict__ sbase, const unsigned bo, const unsigned row_bytes) {
    typedef const char __attribute__((address_space(1))) * gptr1;
    const gptr1 g = (gptr1)sbase;
    PmnCorners c;
    c.t00 = pmn_global_load4(g + bo);
    c.t01 = pmn_global_load4(g + bo + C * 4);
    c.t10 = pmn_global_load4(g + (bo + row_bytes));
    c.t11 = pmn_global_load4(g + (bo + row_bytes) + C * 4);
    return c;
}

template <int LPG, int CG>
__device__ __forceinline__ float blend_corners(const PmnCorners& c, float4 w4, const float4 refq) {
    w4 = pmn_settle4(w4);  // scalars of a float4, broadcast into packed math: lesson 46 (gather_common.hpp)
    const pmn_f2 wa = {w4.x, w4.x}, wb = {w4.y, w4.y}, wc = {w4.z, w4.z}, wd = {w4.w, w4.w};
    pmn_f2 lo = pmn_f2{c.t00.x, c.t00.y} * wa;
    pmn_f2 hi = pmn_f2{c.t00.z, c.t00.w} * wa;
    lo = __builtin_elementwise_fma(pmn_f2{c.t01.x, c.t01.y}, wb, lo);
    hi = __builtin_elementwise_fma(pmn_f2{c.t01.z, c.t01.w}, wb, hi);
    lo = __builtin_elementwise_fma(pmn_f2{c.t10.x, c.t10.y}, wc, lo);
    hi = __builtin_elementwise_fma(pmn_f2{c.t10.z, c.t10.w}, wc, hi);
    lo = __builtin_elementwise_fma(pmn_f2{c.t11.x, c.t11.y}, wd, lo);
    hi = __builtin_elementwise_fma(pmn_f2{c.t11.z, c.t11.w}, wd, hi);
    float s = fmaf(hi.y, refq.w, fmaf(hi.x, refq.z, fmaf(lo.y, refq.y, lo.x * refq.x)));
    if (LPG == 2) s += pmn_pair_swap(s);
    return s * (1.0f / CG);
}

// EXACT: the hypothesis count equals the compile-time bound DT, so every `d < D` test folds away and the unrolled phase-B
// loop is straight-line code (with run-time guards each item becomes its own basic block and hipcc serialises
// load -> wait -> compute per item: no memory-level parallelism).
// (occupancy: the FeatureWeightNet launches run 5 waves per SIMD -- 96 registers; at 98 the stage-1 launch lost 10 % -- the others 4 / 3;
// 5 waves measured no faster for the known-weights launches)
template <int C, int G, int MODE, int DT, bool EXACT>
__global__ __launch_bounds__(PMN_BLOCK, (MODE == MODE_PIXELWISE ? 3 : (MODE == MODE_NEIGHBOR && DT <= 16) ? 5 : 4)) void gather_corr_kernel(const GatherArgs a) {
    constexpr int LPI = C / 4;               // lanes per (pixel, hypothesis) item
    constexpr int NPIX = PMN_BLOCK / LPI;    // pixels per workgroup tile
    constexpr int CG = C / G;                // channels per correlation group (4 or 8)
    constexpr int LPG = CG / 4;              // lanes per group (1 or 2)
    constexpr int PAD = 32 / G;              // LDS row padding: rows of different groups land on different banks
    constexpr int DSTEP = PMN_BLOCK / NPIX;  // item role: a thread's hypotheses are dA0, dA0+DSTEP, ...
    constexpr int NIT = DT / DSTEP;          // item role: hypotheses per thread
    constexpr int NI_MAX = (MODE == MODE_PIXELWISE) ? 2 : 4;  // PIXELWISE also keeps NIT*G running sums live
    constexpr int NI = NIT < NI_MAX ? NIT : NI_MAX;            // MLP evaluated for NI items at a time
    static_assert(CG == 4 || CG == 8, "group size must be 4 or 8 channels");
    static_assert(NIT >= 1, "DT must cover at least one hypothesis per item-role thread");

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int tile = pmn_xcd_tile(blockIdx.x, a.ntiles);
    constexpr int DEXACT = (MODE == MODE_NEIGHBOR) ? (DT == 16 ? 9 : 17) : DT;  // FeatureWeightNet: EXACT = the reference's K = 9 / 17
    const int D = EXACT ? DEXACT : a.D;
    const int N = a.N, h = a.h, w = a.w, hs = a.hs, ws = a.ws;
    const int hw = h * w;
    const int items = NPIX * D;
    const int SS = items + PAD;
    const int p0 = tile * NPIX;

    extern __shared__ float4 smem4[];
    float* simt = reinterpret_cast<float*>(smem4);                // [G][SS] similarity tile
    float* wlds_a = simt + ((G * SS + 3) & ~3);                   // MLP a weights (16-byte aligned)
    float* wlds_b = wlds_a + MLP_LDS_FLOATS;                      // MLP b weights           (PIXELWISE)
    unsigned long long* vwkey = reinterpret_cast<unsigned long long*>(wlds_b + MLP_LDS_FLOATS);  // [NPIX]
    int* tab = reinterpret_cast<int*>(vwkey + NPIX);              // [2K]                    (NEIGHBOR)

    // ---- prologue -------------------------------------------------------------------------------------------
    for (int i = tid; i < PMN_MLP_FLOATS; i += PMN_BLOCK) {
        wlds_a[i] = a.mlp_a[i];
        if (MODE == MODE_PIXELWISE) wlds_b[i] = a.mlp_b[i];
    }
    if (MODE == MODE_NEIGHBOR) {
        // static indices only: a dynamic index into the by-value kernarg struct would spill it to scratch
#pragma unroll
        for (int i = 0; i < 2 * PMN_MAX_NEIGHBORS; ++i)
            if (tid == i) tab[i] = a.table[i];
    }

    // item role: a fixed pixel of the tile, hypotheses dA0 + j*DSTEP
    const int pixA = tid % NPIX, dA0 = tid / NPIX;
    const int pA = p0 + pixA;
    const bool okA = pA < hw;
    const int yA = okA ? pA / w : 0, xA = okA ? pA - yA * w : 0;

    // lane role: lane lc of the lane group that owns pixel `grp` of the tile
    const int grp = tid / LPI, lc = tid % LPI;
    const int pB = p0 + grp;
    const bool okB = pB < hw;
    float4 refq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (okB) refq = reinterpret_cast<const float4*>(a.ref)[((size_t)b * hw + pB) * LPI + lc];
    const int yB = okB ? pB / w : 0, xB = okB ? pB - yB * w : 0;
    const int wv = w >> a.vw_shift, hwv = (h >> a.vw_shift) * wv;
    const int vw_idx = (yB >> a.vw_shift) * wv + (xB >> a.vw_shift);
    const bool owner = (lc % LPG) == 0;
    const int gB = lc / LPG;


    if (MODE == MODE_VIEWS) {
        // ================= known view weights: barrier-free streaming over the views ================================
        // Each lane projects RPL = DT/LPI hypotheses of ITS OWN pixel (d = lc + j*LPI), keeps the tap records in
        // registers and the records are broadcast inside the lane group when hypothesis d is gathered -- no LDS
        // round trip and no workgroup barrier inside the view loop, so every wave streams on its own and the tap
        // loads of one wave overlap the projection math of the others.  Sums over views accumulate in registers.
        constexpr int RPL = DT / LPI;
        static_assert(DT % LPI == 0, "DT is a multiple of the lane-group size");
        float acc[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) acc[d] = 0.0f;
        const float xf = (float)xB, yf = (float)yB;
        float rdep[RPL];  // my RPL hypotheses (the same for every view)
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
            const int d = lc + j * LPI;
            rdep[j] = (okB && d < D) ? a.depth[((size_t)b * D + d) * hw + pB] : -1.0f;
        }
        for (int v = 0; v < N; ++v) {
            const PmnPose lane_pose = pmn_make_pose(a.proj + ((size_t)b * N + v) * 16, xf, yf, h, w);
            float rw00[RPL], rw01[RPL], rw10[RPL], rw11[RPL];
            int roff[RPL];
#pragma unroll
            for (int j = 0; j < RPL; ++j) {
                const int d = lc + j * LPI;
                PmnTaps t;
                t.off = 0;
                t.w00 = t.w01 = t.w10 = t.w11 = 0.0f;
                if (okB && d < D) {
                    // the reference's own IEEE chain (pmn_common.hpp); a behind-camera hypothesis lands outside every tap
                    float ix, iy;
                    pmn_pose_position(lane_pose, rdep[j], h, w, hs, ws, ix, iy);
                    t = pmn_make_taps(ix, iy, hs, ws);
                }
                rw00[j] = t.w00; rw01[j] = t.w01; rw10[j] = t.w10; rw11[j] = t.w11;
                roff[j] = t.off;
            }
            const char* sbase = pmn_view_base<C>(a, v, b, hs, ws);
            const unsigned lane_bytes = lc * 16u, row_bytes = (unsigned)ws * (C * 4);
            const float vw = okB ? a.vw_in[((size_t)b * N + v) * hwv + vw_idx] : 0.0f;
            // Batches of NB items: records broadcast + all 4*NB corner loads issued, THEN the NB blends; the fences keep hipcc
            // from re-serialising each item (load -> wait -> blend) or hoisting every load of the unrolled loop (spills).
            constexpr int NB = 2;  // measured: 4 (at 3 waves/SIMD) is no faster on the cascade's hypotheses
            static_assert(DT % NB == 0, "hypothesis bound is a multiple of the batch");
#pragma unroll
            for (int d0 = 0; d0 < DT; d0 += NB) {
                PmnCorners cn[NB];
                float4 wq[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int d = d0 + i;
                    if (EXACT || d < D) {
                        float4 w4;
                        int off;
                        // compile-time (j, source lane) of hypothesis d
#define PMN_BCAST_CASE(SL)                                                    \
    case SL:                                                                  \
        w4.x = group_bcast_f<LPI, SL>(rw00[d / LPI]);                         \
        w4.y = group_bcast_f<LPI, SL>(rw01[d / LPI]);                         \
        w4.z = group_bcast_f<LPI, SL>(rw10[d / LPI]);                         \
        w4.w = group_bcast_f<LPI, SL>(rw11[d / LPI]);                         \
        off = group_bcast_i<LPI, SL>(roff[d / LPI]);                          \
        break;
                        switch (d % LPI) {
                            PMN_BCAST_CASE(0) PMN_BCAST_CASE(1) PMN_BCAST_CASE(2) PMN_BCAST_CASE(3)
                            PMN_BCAST_CASE(4) PMN_BCAST_CASE(5) PMN_BCAST_CASE(6) PMN_BCAST_CASE(7)
                            PMN_BCAST_CASE(8) PMN_BCAST_CASE(9) PMN_BCAST_CASE(10) PMN_BCAST_CASE(11)
                            PMN_BCAST_CASE(12) PMN_BCAST_CASE(13) PMN_BCAST_CASE(14) PMN_BCAST_CASE(15)
                            default: w4 = make_float4(0.f, 0.f, 0.f, 0.f); off = 0; break;
                        }
#undef PMN_BCAST_CASE
                        wq[i] = w4;
                        cn[i] = load_corners<C>(sbase, (unsigned)off * (C * 4) + lane_bytes, row_bytes);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int d = d0 + i;
                    if (EXACT || d < D) acc[d] = mul_add_unfused(acc[d], blend_corners<LPG, CG>(cn[i], wq[i], refq), vw);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (owner) {
#pragma unroll
            for (int d = 0; d < DT; ++d)
                if (d < D) simt[gB * SS + d * NPIX + grp] = acc[d];
        }
        __syncthreads();
        if (!okA) return;
        float wtot = 1e-5f;
        const int vwi = (yA >> a.vw_shift) * wv + (xA >> a.vw_shift);
        for (int v = 0; v < N; ++v) wtot += a.vw_in[((size_t)b * N + v) * hwv + vwi];
        float x[NIT][G], o[NIT];
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int d = min(dA0 + j * DSTEP, D - 1);
#pragma unroll
            for (int g = 0; g < G; ++g) x[j][g] = simt[g * SS + d * NPIX + pixA] / wtot;
        }
        mlp_items<G, NIT, NI>(wlds_a, x, o);
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int d = dA0 + j * DSTEP;
            if (d < D) {
                if (a.sim_out) {
#pragma unroll
                    for (int g = 0; g < G; ++g) a.sim_out[(((size_t)b * G + g) * D + d) * hw + pA] = x[j][g];
                }
                a.out[((size_t)b * hw + pA) * D + d] = o[j];  // cost is hypothesis-last [B,h,w,D]
            }
        }
        return;
    }

    // ================= PIXELWISE (N views, view weights computed here) / NEIGHBOR (one pseudo view) ======================
    float ssum[NIT][G];
#pragma unroll
    for (int j = 0; j < NIT; ++j)
#pragma unroll
        for (int g = 0; g < G; ++g) ssum[j][g] = 0.0f;
    float wsum = 1e-5f;
    __syncthreads();  // tab / MLP weights visible

    for (int v = 0; v < N; ++v) {
        if (MODE == MODE_PIXELWISE && tid < NPIX) vwkey[tid] = 0ull;
        const char* sbase = MODE == MODE_NEIGHBOR ? reinterpret_cast<const char*>(a.ref) + ((size_t)b * hs * ws) * (C * 4)
                                                  : pmn_view_base<C>(a, v, b, hs, ws);
        const unsigned lane_bytes = lc * 16u, row_bytes = (unsigned)ws * (C * 4);
        if constexpr (MODE == MODE_PIXELWISE) {
            // Lane role as in MODE_VIEWS: every lane projects RPL hypotheses of ITS OWN pixel, the records stay in registers and
            // are broadcast inside the lane group by DPP when hypothesis d is gathered -- no LDS records, no barrier while the
            // view's similarity tile is filled (round 1 walked 32 hypotheses at a time behind 4 barriers per view).
            constexpr int RPL = DT / LPI;
            const float xf = (float)xB, yf = (float)yB;
            const PmnPose lane_pose = pmn_make_pose(a.proj + ((size_t)b * N + v) * 16, xf, yf, h, w);
            float rw00[RPL], rw01[RPL], rw10[RPL], rw11[RPL];
            int roff[RPL];
#pragma unroll
            for (int j = 0; j < RPL; ++j) {
                const int d = lc + j * LPI;
                PmnTaps t;
                t.off = 0;
                t.w00 = t.w01 = t.w10 = t.w11 = 0.0f;
                if (okB && d < D) {
                    float ix, iy;
                    pmn_pose_position(lane_pose, a.depth[((size_t)b * D + d) * hw + pB], h, w, hs, ws, ix, iy);
                    t = pmn_make_taps(ix, iy, hs, ws);
                }
                rw00[j] = t.w00; rw01[j] = t.w01; rw10[j] = t.w10; rw11[j] = t.w11;
                roff[j] = t.off;
            }
            constexpr int NB = 2;
#pragma unroll
            for (int d0 = 0; d0 < DT; d0 += NB) {
                PmnCorners cn[NB];
                float4 wq[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int d = d0 + i;
                    if (EXACT || d < D) {
                        float4 w4;
                        int off;
#define PMN_BCAST_CASE(SL)                                                    \
    case SL:                                                                  \
        w4.x = group_bcast_f<LPI, SL>(rw00[d / LPI]);                         \
        w4.y = group_bcast_f<LPI, SL>(rw01[d / LPI]);                         \
        w4.z = group_bcast_f<LPI, SL>(rw10[d / LPI]);                         \
        w4.w = group_bcast_f<LPI, SL>(rw11[d / LPI]);                         \
        off = group_bcast_i<LPI, SL>(roff[d / LPI]);                          \
        break;
                        switch (d % LPI) {
                            PMN_BCAST_CASE(0) PMN_BCAST_CASE(1) PMN_BCAST_CASE(2) PMN_BCAST_CASE(3)
                            PMN_BCAST_CASE(4) PMN_BCAST_CASE(5) PMN_BCAST_CASE(6) PMN_BCAST_CASE(7)
                            PMN_BCAST_CASE(8) PMN_BCAST_CASE(9) PMN_BCAST_CASE(10) PMN_BCAST_CASE(11)
                            PMN_BCAST_CASE(12) PMN_BCAST_CASE(13) PMN_BCAST_CASE(14) PMN_BCAST_CASE(15)
                            default: w4 = make_float4(0.f, 0.f, 0.f, 0.f); off = 0; break;
                        }
#undef PMN_BCAST_CASE
                        wq[i] = w4;
                        cn[i] = load_corners<C>(sbase, (unsigned)off * (C * 4) + lane_bytes, row_bytes);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int d = d0 + i;
                    if (EXACT || d < D) {
                        const float s = blend_corners<LPG, CG>(cn[i], wq[i], refq);
                        if (owner) simt[gB * SS + d * NPIX + grp] = s;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();  // this view's similarity tile is complete
        } else if constexpr (MODE == MODE_NEIGHBOR) {
            // FeatureWeightNet, round 6: the tap records stay in REGISTERS as in MODE_VIEWS -- lane lc of a pixel's lane group projects
            // neighbours k = lc, lc + LPI, ... of ITS OWN pixel (get_grid's position, pmn_neighbor_position) and the record is broadcast
            // inside the group by DPP when neighbour k is gathered.  Rounds 1-5 had the item role park the records in LDS behind two
            // workgroup barriers (one before, one after the gather); what is left is the one barrier of the hand-over to the MLP.
            constexpr int KD = EXACT ? (DT == 16 ? 9 : 17) : DT;  // EXACT here = "K is the reference's 9 (DT 16) or 17 (DT 32)"
            constexpr int RPL = (KD + LPI - 1) / LPI;
            float rw00[RPL], rw01[RPL], rw10[RPL], rw11[RPL];
            int roff[RPL];
#pragma unroll
            for (int j = 0; j < RPL; ++j) {
                const int k = lc + j * LPI;
                PmnTaps t;
                t.off = 0;
                t.w00 = t.w01 = t.w10 = t.w11 = 0.0f;
                if (okB && k < (EXACT ? KD : D)) {
                    float ix, iy;
                    const float ox = a.offsets[((size_t)b * 2 * D + 2 * k) * hw + pB];
                    const float oy = a.offsets[((size_t)b * 2 * D + 2 * k + 1) * hw + pB];
                    pmn_neighbor_position((float)xB, (float)yB, tab[2 * k], tab[2 * k + 1], ox, oy, h, w, ix, iy);
                    t = pmn_make_taps(ix, iy, hs, ws);
                }
                rw00[j] = t.w00; rw01[j] = t.w01; rw10[j] = t.w10; rw11[j] = t.w11;
                roff[j] = t.off;
            }
            constexpr int NB = 2;
#pragma unroll
            for (int d0 = 0; d0 < KD; d0 += NB) {
                PmnCorners cn[NB];
                float4 wq[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int d = d0 + i;
                    if (d < KD && (EXACT || d < D)) {
                        float4 w4;
                        int off;
#define PMN_BCAST_CASE(SL)                                                    \
    case SL:                                                                  \
        w4.x = group_bcast_f<LPI, SL>(rw00[d / LPI]);                         \
        w4.y = group_bcast_f<LPI, SL>(rw01[d / LPI]);                         \
        w4.z = group_bcast_f<LPI, SL>(rw10[d / LPI]);                         \
        w4.w = group_bcast_f<LPI, SL>(rw11[d / LPI]);                         \
        off = group_bcast_i<LPI, SL>(roff[d / LPI]);                          \
        break;
                        switch (d % LPI) {
                            PMN_BCAST_CASE(0) PMN_BCAST_CASE(1) PMN_BCAST_CASE(2) PMN_BCAST_CASE(3)
                            PMN_BCAST_CASE(4) PMN_BCAST_CASE(5) PMN_BCAST_CASE(6) PMN_BCAST_CASE(7)
                            PMN_BCAST_CASE(8) PMN_BCAST_CASE(9) PMN_BCAST_CASE(10) PMN_BCAST_CASE(11)
                            PMN_BCAST_CASE(12) PMN_BCAST_CASE(13) PMN_BCAST_CASE(14) PMN_BCAST_CASE(15)
                            default: w4 = make_float4(0.f, 0.f, 0.f, 0.f); off = 0; break;
                        }
#undef PMN_BCAST_CASE
                        wq[i] = w4;
                        cn[i] = load_corners<C>(sbase, (unsigned)off * (C * 4) + lane_bytes, row_bytes);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int d = d0 + i;
                    if (d < KD && (EXACT || d < D)) {
                        const float s = blend_corners<LPG, CG>(cn[i], wq[i], refq);
                        if (owner) simt[gB * SS + d * NPIX + grp] = s;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();  // the similarity tile is complete
        }
        // item role: this view's similarities of my items (re-read from the LDS tile in chunks of NI to keep few live)
        if (MODE == MODE_NEIGHBOR) {
            if (!okA) return;
            // scalar MLP, NI items per call: on this launch the packed-pair form (mlp_items) measured 7 % SLOWER at stage 1
            // (137.7 vs 128.5 us, same box) -- and re-pairing a float [NIT][G] array sent it through scratch.
            // Only the items some lane of this WAVE owns are evaluated: the hypothesis bound DT = 16 / 32 covers K = 9 / 17
            // neighbours, and a thread's items are d = dA0 + j * DSTEP -- at K = 9 that is 2.25 of the 4 (C = 16), 1.25 of the 2
            // (C = 32), 0.56 of the 1 (C = 64) rounds the clamped form evaluated for every thread, and the MLP (200-264 FMAs per
            // item) was 57 % of this kernel's instructions.  The count is wave-uniform (dA0 = tid / NPIX), so the switch is a
            // scalar branch; every evaluated item is computed exactly as before.
            const int wave_d0 = __builtin_amdgcn_readfirstlane((tid & ~63) / NPIX);  // smallest dA0 of this wave
            const int nvw = wave_d0 < D ? (D - wave_d0 + DSTEP - 1) / DSTEP : 0;     // items its lanes own: j < nvw
            float o[NIT];
#pragma unroll
            for (int j = 0; j < NIT; ++j) o[j] = 0.0f;
#pragma unroll
            for (int c = 0; c < NIT / NI; ++c) {
                auto run = [&](auto nv_tag) {
                    constexpr int NV = decltype(nv_tag)::value;
                    float xc[NV][G], oc[NV];
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        const int d = min(dA0 + (c * NI + i) * DSTEP, D - 1);
#pragma unroll
                        for (int g = 0; g < G; ++g) xc[i][g] = pmn_settle(simt[g * SS + d * NPIX + pixA]);  // (hipcc may pack the MLP's fmas: lesson 46)
                    }
                    mlp_from_lds<G, NV>(wlds_a, xc, oc);
#pragma unroll
                    for (int i = 0; i < NV; ++i) o[c * NI + i] = oc[i];
                };
                const int nvc = min(max(nvw - c * NI, 0), NI);
                static_assert(NI <= 4, "switch below");
                switch (nvc) {
                    case 1: run(std::integral_constant<int, 1>{}); break;
                    case 2: if constexpr (NI >= 2) run(std::integral_constant<int, 2>{}); break;
                    case 3: if constexpr (NI >= 3) run(std::integral_constant<int, 3>{}); break;
                    case 4: if constexpr (NI >= 4) run(std::integral_constant<int, 4>{}); break;
                    default: break;
                }
            }
#pragma unroll
            for (int j = 0; j < NIT; ++j) {
                const int d = dA0 + j * DSTEP;
                if (d < D) a.out[((size_t)b * D + d) * hw + pA] = pmn_sigmoid(o[j]);
            }
            return;
        }
        // PixelwiseNet + max over D (first arg-max on ties through the ~d low word)
        if constexpr (MODE == MODE_PIXELWISE) {
            unsigned long long best = 0ull;
#pragma unroll
            for (int c = 0; c < NIT / NI; ++c) {
                static_assert(NI == 2, "PixelwiseNet is evaluated for one pair of items at a time");
                float r[NI];
                pmn_f2 xq[1][G], rq[1];
                {
                    const int da = min(dA0 + (c * NI) * DSTEP, D - 1), db = min(dA0 + (c * NI + 1) * DSTEP, D - 1);
#pragma unroll
                    for (int g = 0; g < G; ++g) xq[0][g] = pmn_f2{pmn_settle(simt[g * SS + da * NPIX + pixA]), pmn_settle(simt[g * SS + db * NPIX + pixA])};
                }
                mlp_pairs_from_lds<G, 1>(wlds_b, xq, rq);
                r[0] = rq[0].x;
                r[1] = rq[0].y;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int d = dA0 + (c * NI + i) * DSTEP;
                    if (d < D) {
                        const unsigned long long key = ((unsigned long long)__float_as_uint(pmn_sigmoid(r[i])) << 32) |
                                                       (unsigned long long)(0xFFFFFFFFu - (unsigned)d);
                        best = key > best ? key : best;
                    }
                }
            }
            atomicMax(&vwkey[pixA], best);
        }
        __syncthreads();
        const unsigned long long key = vwkey[pixA];
        const float vwp = __uint_as_float((unsigned)(key >> 32));
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int d = min(dA0 + j * DSTEP, D - 1);
#pragma unroll
            for (int g = 0; g < G; ++g) ssum[j][g] = mul_add_unfused(ssum[j][g], pmn_settle(simt[g * SS + d * NPIX + pixA]), vwp);
        }
        wsum += vwp;
        if (tid < NPIX && okA) {
            const size_t o = ((size_t)b * N + v) * hw + pA;
            a.vw_out[o] = vwp;
            if (a.vw_argmax) a.vw_argmax[o] = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
        }
        __syncthreads();  // vwkey / simt are rewritten by the next view
    }

    if constexpr (MODE == MODE_NEIGHBOR) return;  // (its single pseudo view returned above; without this the dead tail below
                                                  //  still costs the instantiation a scratch frame)
    if (!okA) return;
    float o[NIT];
#pragma unroll
    for (int j = 0; j < NIT; ++j)
#pragma unroll
        for (int g = 0; g < G; ++g) ssum[j][g] = ssum[j][g] / wsum;
    mlp_items<G, NIT, NI>(wlds_a, ssum, o);
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
        const int d = dA0 + j * DSTEP;
        if (d < D) {
            if (a.sim_out) {
#pragma unroll
                for (int g = 0; g < G; ++g) a.sim_out[(((size_t)b * G + g) * D + d) * hw + pA] = ssum[j][g];
            }
            a.out[((size_t)b * hw + pA) * D + d] = o[j];  // cost is hypothesis-last [B,h,w,D]
        }
    }
}

// ---- PixelwiseNet launch, wave-private form (round 5) -----------------------------------------------------------------
// MODE_PIXELWISE of the kernel above hands every view's similarity tile from the lane role to the item role ACROSS the workgroup:
// three workgroup barriers, 64 ds_write_b32 + 32 ds_read_b32 + one 64-bit LDS atomic max per wave and view (round 4's counters:
// LDS 32 % busy, 20 % of the wave cycles waiting, 3 waves per SIMD).  Here a WAVE owns its pixels from the gather to the view
// weight: the lane groups that gather pixel p and the lanes that run PixelwiseNet on pixel p's hypotheses sit in the same wave, so
//   * the hand-over is wave-private LDS (a wave's LDS instructions execute in order: no barrier anywhere in the view loop, every
//     wave streams on its own as in MODE_VIEWS);
//   * an owner lane parks four consecutive hypotheses with ONE ds_write_b128 (row = (pixel, group), hypothesis fastest) and an
//     item-role lane -- (pixel, NIT consecutive hypotheses) -- fetches a group's NIT values with one ds_read_b128 / b64;
//   * the max over D is a DPP max over the 16 (PW = 4) or 32 (PW = 2) lanes of the pixel on the 64-bit (response bits | ~d) key
//     -- no atomic, no key array --, and the view's similarities stay in the item lane's registers for the weighted sum.
// PW = pixels per wave: 4 = one 16-lane group per pixel walks all 64 hypotheses; 2 = two groups per pixel walk 32 each, half the
// work per wave in twice as many waves (the launch is 7500 waves of PW = 4 over 3072-4096 wave slots: a short tail).
// Every (pixel, hypothesis, view) goes through exactly the operations of the kernel above in the same order; the max is order-free:
// same bits (scripts/ab_forward_bits.py).  C = 64, G = 8, D <= 64 (the cascade's stage 3); other shapes use the kernel above.
template <int LPI>
__device__ __forceinline__ void bcast_record(const int sl, const float r00, const float r01, const float r10, const float r11,
                                             const int ro, float4& w4, int& off) {
#define PMN_BCAST_CASE(SL)                      \
    case SL:                                    \
        w4.x = group_bcast_f<LPI, SL>(r00);     \
        w4.y = group_bcast_f<LPI, SL>(r01);     \
        w4.z = group_bcast_f<LPI, SL>(r10);     \
        w4.w = group_bcast_f<LPI, SL>(r11);     \
        off = group_bcast_i<LPI, SL>(ro);       \
        break;
    switch (sl) {
        PMN_BCAST_CASE(0) PMN_BCAST_CASE(1) PMN_BCAST_CASE(2) PMN_BCAST_CASE(3)
        PMN_BCAST_CASE(4) PMN_BCAST_CASE(5) PMN_BCAST_CASE(6) PMN_BCAST_CASE(7)
        PMN_BCAST_CASE(8) PMN_BCAST_CASE(9) PMN_BCAST_CASE(10) PMN_BCAST_CASE(11)
        PMN_BCAST_CASE(12) PMN_BCAST_CASE(13) PMN_BCAST_CASE(14) PMN_BCAST_CASE(15)
        default: w4 = make_float4(0.f, 0.f, 0.f, 0.f); off = 0; break;
    }
#undef PMN_BCAST_CASE
}

template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_max_u64(const unsigned long long v) {
    const int lo = (int)(unsigned)(v & 0xFFFFFFFFull), hi = (int)(unsigned)(v >> 32);
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
    return o > v ? o : v;
}

#ifndef PMN_PW
#define PMN_PW 2  // pixels per wave of the PixelwiseNet launch (4 or 2; 0 = the workgroup-tile kernel above)
#endif
#ifndef PMN_PW_WAVES
#define PMN_PW_WAVES 4  // waves per SIMD the launch is compiled for
#endif
#ifndef PMN_PW_INTERLEAVE
#define PMN_PW_INTERLEAVE 1  // PW = 2: lane group k of a pixel walks hypotheses k, k + 2, ... (1: 2 % faster at configs[1]) or the block [k*DL, (k+1)*DL) (0)
#endif

template <int PW, bool EXACT>
__global__ __launch_bounds__(PMN_BLOCK, PMN_PW_WAVES) void pixelwise_wave_kernel(const GatherArgs a) {
    constexpr int C = 64, G = 8, DT = 64, LPI = 16, CG = 8, LPG = 2;
    constexpr int LGP = 4 / PW;            // lane groups per pixel
    constexpr int DL = DT / LGP;           // hypotheses a lane group walks: its block [k*DL, (k+1)*DL)
    constexpr int RPL = DL / LPI;          // hypotheses a lane projects per view
    constexpr int LPP = 64 / PW;           // item role: lanes per pixel
    constexpr int NIT = DT / LPP;          // item role: consecutive hypotheses per lane (4 or 2)
    constexpr int GS = DT + 4;             // LDS row pitch (floats): rows of the 8 groups start 17 x 16 B apart
    constexpr int PS = G * GS;             // LDS floats per pixel
    constexpr int WPB = PMN_BLOCK / 64;    // waves per workgroup
    constexpr int NPIX = WPB * PW;         // pixels per workgroup
    // LDS position of a hypothesis inside its (pixel, group) row: lane group k's il-th item sits at k*DL + il.  Which hypothesis that
    // is: the block form (d = position) or, interleaved, d = k + LGP*il -- the lane groups of a pixel then gather NEIGHBOURING
    // hypotheses in the same instruction (their taps share cache lines) instead of two segments half an epipolar line apart.
    constexpr bool IL = PMN_PW_INTERLEAVE != 0 && LGP > 1;
    auto hyp_of = [](int kk, int il) { return IL ? kk + LGP * il : kk * DL + il; };
    static_assert(PW == 4 || PW == 2, "pixels per wave");
    static_assert(NIT % 2 == 0, "PixelwiseNet runs on pairs of items");

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.y;
    const int tile = pmn_xcd_tile(blockIdx.x, a.ntiles);
    const int D = EXACT ? DT : a.D;
    const int N = a.N, h = a.h, w = a.w, hs = a.hs, ws = a.ws;
    const int hw = h * w;
    const int pw0 = tile * NPIX + wave * PW;  // first pixel of this wave

    extern __shared__ float4 smem4[];
    float* wlds_a = reinterpret_cast<float*>(smem4);   // SimilarityNet weights
    float* wlds_b = wlds_a + MLP_LDS_FLOATS;           // PixelwiseNet weights
    float* simw = wlds_b + MLP_LDS_FLOATS + wave * (PW * PS);  // this wave's [PW][G][GS] similarity rows
    for (int i = tid; i < PMN_MLP_FLOATS; i += PMN_BLOCK) {
        wlds_a[i] = a.mlp_a[i];
        wlds_b[i] = a.mlp_b[i];
    }
    __syncthreads();  // the only workgroup barrier

    // lane role: lane lc of lane group lg gathers block k of pixel pl
    const int lg = lane >> 4, lc = lane & 15;
    const int pl = lg / LGP, k = lg % LGP;
    const int pB = pw0 + pl;
    const bool okB = pB < hw;
    float4 refq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (okB) refq = reinterpret_cast<const float4*>(a.ref)[((size_t)b * hw + pB) * LPI + lc];
    const int yB = okB ? pB / w : 0, xB = okB ? pB - yB * w : 0;
    const bool owner = (lc % LPG) == 0;
    const int gB = lc / LPG;
    float rdep[RPL];
    bool rok[RPL];
#pragma unroll
    for (int j = 0; j < RPL; ++j) {
        const int d = hyp_of(k, lc + j * LPI);
        rok[j] = okB && (EXACT || d < D);
        rdep[j] = rok[j] ? a.depth[((size_t)b * D + d) * hw + pB] : 1.0f;
    }
    float* srow = simw + pl * PS + gB * GS + k * DL;  // owner lanes: this (pixel, group) row, my block

    // item role: lane q of the LPP lanes of pixel pi runs the pointwise nets on hypotheses [d0, d0 + NIT)
    const int pi = lane / LPP, q = lane % LPP;
    const int d0 = q * NIT;
    const int pA = pw0 + pi;
    const bool okA = pA < hw;
    const float* xrow = simw + pi * PS + d0;

    float ssum[NIT][G];
#pragma unroll
    for (int j = 0; j < NIT; ++j)
#pragma unroll
        for (int g = 0; g < G; ++g) ssum[j][g] = 0.0f;
    float wsum = 1e-5f;
    const float xf = (float)xB, yf = (float)yB;
    const unsigned lane_bytes = lc * 16u, row_bytes = (unsigned)ws * (C * 4);

    for (int v = 0; v < N; ++v) {
        // ---- lane role: project my RPL hypotheses (records stay in registers), then walk the block ----------------------------
        const PmnPose lane_pose = pmn_make_pose(a.proj + ((size_t)b * N + v) * 16, xf, yf, h, w);
        float rw00[RPL], rw01[RPL], rw10[RPL], rw11[RPL];
        int roff[RPL];
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
            PmnTaps t;
            t.off = 0;
            t.w00 = t.w01 = t.w10 = t.w11 = 0.0f;
            if (rok[j]) {
                float ix, iy;
                pmn_pose_position(lane_pose, rdep[j], h, w, hs, ws, ix, iy);
                t = pmn_make_taps(ix, iy, hs, ws);
            }
            rw00[j] = t.w00; rw01[j] = t.w01; rw10[j] = t.w10; rw11[j] = t.w11;
            roff[j] = t.off;
        }
        const char* sbase = pmn_view_base<C>(a, v, b, hs, ws);
        constexpr int NB = 2;
#pragma unroll
        for (int i0 = 0; i0 < DL; i0 += 2 * NB) {
            float sv[2 * NB];
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                PmnCorners cn[NB];
                float4 wq[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int il = i0 + hb * NB + i;  // position in the block; hypothesis d = k*DL + il
                    float4 w4;
                    int off;
                    bcast_record<LPI>(il % LPI, rw00[il / LPI], rw01[il / LPI], rw10[il / LPI], rw11[il / LPI], roff[il / LPI], w4, off);
                    wq[i] = w4;
                    cn[i] = load_corners<C>(sbase, (unsigned)off * (C * 4) + lane_bytes, row_bytes);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NB; ++i) sv[hb * NB + i] = blend_corners<LPG, CG>(cn[i], wq[i], refq);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (owner) *reinterpret_cast<float4*>(srow + i0) = make_float4(sv[0], sv[1], sv[2], sv[3]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- item role: PixelwiseNet on my NIT hypotheses of this view, max over the pixel's lanes --------------------------
        // (NIT = 4: the rows are fetched pair by pair for the net and again for the weighted sum -- 16 more LDS reads per view instead
        //  of 32 registers live across the net, which is what decides between 4 waves per SIMD with and without scratch)
        float x[NIT][G];
        if constexpr (NIT == 2) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float2 t2 = *reinterpret_cast<const float2*>(xrow + g * GS);
                x[0][g] = t2.x; x[1][g] = t2.y;
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {  // LDS -> packed MLP: lesson 46 (a second loop: all eight reads in flight, then the moves)
                x[0][g] = pmn_settle(x[0][g]); x[1][g] = pmn_settle(x[1][g]);
            }
        }
        unsigned long long best = 0ull;
#pragma unroll
        for (int c = 0; c < NIT / 2; ++c) {
            pmn_f2 xq[1][G], rq[1];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if constexpr (NIT == 2) {
                    xq[0][g] = pmn_f2{x[0][g], x[1][g]};
                } else {
                    const float2 t2 = *reinterpret_cast<const float2*>(xrow + g * GS + 2 * c);
                    xq[0][g] = pmn_f2{pmn_settle(t2.x), pmn_settle(t2.y)};
                }
            }
            mlp_pairs_from_lds<G, 1>(wlds_b, xq, rq);
            const float r[2] = {rq[0].x, rq[0].y};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int d = hyp_of((d0 + 2 * c + i) / DL, (d0 + 2 * c + i) % DL);
                if (EXACT || d < D) {
                    const unsigned long long key = ((unsigned long long)__float_as_uint(pmn_sigmoid(r[i])) << 32) |
                                                   (unsigned long long)(0xFFFFFFFFu - (unsigned)d);
                    best = key > best ? key : best;
                }
            }
        }
        best = dpp_max_u64<0xB1>(best);   // quad_perm [1,0,3,2]
        best = dpp_max_u64<0x4E>(best);   // quad_perm [2,3,0,1]
        best = dpp_max_u64<0x124>(best);  // row_ror:4
        best = dpp_max_u64<0x128>(best);  // row_ror:8  -> every lane of a 16-lane row holds the row's max
        if constexpr (LPP == 32) {
            const unsigned long long o = __shfl_xor(best, 16, 64);
            best = o > best ? o : best;
        }
        const float vwp = __uint_as_float((unsigned)(best >> 32));
        if constexpr (NIT == 4) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float4 t4 = pmn_settle4(*reinterpret_cast<const float4*>(xrow + g * GS));
                x[0][g] = t4.x; x[1][g] = t4.y; x[2][g] = t4.z; x[3][g] = t4.w;
            }
        }
#pragma unroll
        for (int j = 0; j < NIT; ++j)
#pragma unroll
            for (int g = 0; g < G; ++g) ssum[j][g] = mul_add_unfused(ssum[j][g], x[j][g], vwp);
        wsum += vwp;
        if (q == 0 && okA) {
            const size_t o = ((size_t)b * N + v) * hw + pA;
            a.vw_out[o] = vwp;
            if (a.vw_argmax) a.vw_argmax[o] = (int)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // (the next view's rows are written after these reads: same wave, LDS executes in order)
    }

    if (!okA) return;
    float o[NIT];
#pragma unroll
    for (int j = 0; j < NIT; ++j)
#pragma unroll
        for (int g = 0; g < G; ++g) ssum[j][g] = ssum[j][g] / wsum;
    mlp_items<G, NIT, 2>(wlds_a, ssum, o);
    float* orow = a.out + ((size_t)b * hw + pA) * D + d0;  // cost is hypothesis-last [B,h,w,D]
    if constexpr (IL) {
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int d = hyp_of((d0 + j) / DL, (d0 + j) % DL);
            if (EXACT || d < D) a.out[((size_t)b * hw + pA) * D + d] = o[j];
        }
    } else if constexpr (EXACT) {
        if constexpr (NIT == 4) *reinterpret_cast<float4*>(orow) = make_float4(o[0], o[1], o[2], o[3]);
        else *reinterpret_cast<float2*>(orow) = make_float2(o[0], o[1]);
    } else {
#pragma unroll
        for (int j = 0; j < NIT; ++j)
            if (d0 + j < D) orow[j] = o[j];
    }
    if (a.sim_out) {
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int d = hyp_of((d0 + j) / DL, (d0 + j) % DL);
            if (d < D) {
#pragma unroll
                for (int g = 0; g < G; ++g) a.sim_out[(((size_t)b * G + g) * D + d) * hw + pA] = ssum[j][g];
            }
        }
    }
}

template <int PW>
static int launch_pixelwise_wave(GatherArgs& a, hipStream_t stream) {
    constexpr int NPIX = (PMN_BLOCK / 64) * PW, PS = 8 * (64 + 4);
    a.ntiles = (a.h * a.w + NPIX - 1) / NPIX;
    const size_t lds = (size_t)2 * MLP_LDS_FLOATS * 4 + (size_t)NPIX * PS * 4;
    if (a.D == 64) {
        PMN_LAUNCH((pixelwise_wave_kernel<PW, true>), dim3(a.ntiles, a.B), dim3(PMN_BLOCK), lds, stream, a);
    } else {
        PMN_LAUNCH((pixelwise_wave_kernel<PW, false>), dim3(a.ntiles, a.B), dim3(PMN_BLOCK), lds, stream, a);
    }
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

// ---- host side ------------------------------------------------------------------------------------------------------

template <int C, int G, int MODE, int DT, bool EXACT>
static int launch_gather_impl(GatherArgs& a, hipStream_t stream) {
    constexpr int LPI = C / 4, NPIX = PMN_BLOCK / LPI, PAD = 32 / G;
    const int hw = a.h * a.w;
    a.ntiles = (hw + NPIX - 1) / NPIX;
    const int items = NPIX * a.D, SS = items + PAD;
    size_t lds = (size_t)((G * SS + 3) & ~3) * 4 + 2 * MLP_LDS_FLOATS * 4 + NPIX * 8 +
                 2 * PMN_MAX_NEIGHBORS * 4;
    lds = (lds + 15) & ~(size_t)15;
    auto kern = gather_corr_kernel<C, G, MODE, DT, EXACT>;
    if (lds > 160 * 1024) return PMN_ERR_SHAPE;
    if (lds > 48 * 1024) {
        const int rc = pmn_raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
        if (rc != PMN_OK) return rc;
    }
    PMN_LAUNCH(kern, dim3(a.ntiles, a.B), dim3(PMN_BLOCK), lds, stream, a);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

template <int C, int G, int MODE, int DT>
static int launch_gather(GatherArgs& a, hipStream_t stream) {
    if constexpr (MODE == MODE_VIEWS || MODE == MODE_PIXELWISE) {
        if (a.D == DT) return launch_gather_impl<C, G, MODE, DT, true>(a, stream);
    } else {
        if (a.D == (DT == 16 ? 9 : 17)) return launch_gather_impl<C, G, MODE, DT, true>(a, stream);  // the reference's neighbour counts
    }
    return launch_gather_impl<C, G, MODE, DT, false>(a, stream);
}

// DT = compile-time bound of the hypothesis loop: next power of two >= D, and >= the item-role stride C/4
template <int C, int G, int MODE>
static int dispatch_depth(GatherArgs& a, hipStream_t stream) {
    constexpr int DSTEP = C / 4;  // = PMN_BLOCK / NPIX
    if constexpr (MODE == MODE_PIXELWISE) {
        return launch_gather<C, G, MODE, 64>(a, stream);
    } else if constexpr (MODE == MODE_NEIGHBOR) {  // K = 9 or 17 neighbours
        if (a.D <= 16) return launch_gather<C, G, MODE, 16>(a, stream);
        return launch_gather<C, G, MODE, 32>(a, stream);
    } else {
        if constexpr (DSTEP <= 8) {
            if (a.D <= 8) return launch_gather<C, G, MODE, 8>(a, stream);
        }
        if (a.D <= 16) return launch_gather<C, G, MODE, 16>(a, stream);
        if (a.D <= 32) return launch_gather<C, G, MODE, 32>(a, stream);
        return launch_gather<C, G, MODE, 64>(a, stream);
    }
}

template <int MODE>
static int dispatch_gather(GatherArgs& a, int C, int G, hipStream_t stream) {
    if (C == 64 && G == 8) return dispatch_depth<64, 8, MODE>(a, stream);
    if (C == 32 && G == 8) return dispatch_depth<32, 8, MODE>(a, stream);
    if (C == 16 && G == 4) return dispatch_depth<16, 4, MODE>(a, stream);
    return PMN_ERR_SHAPE;
}

static int warp_correlate_impl(const float* ref_nhwc, const float* src_nhwc, const unsigned long long* src_table,
                               const float* rel_proj, const float* depth_sample, const float* view_weights_in, int vw_shift,
                               const float* similarity_mlp, const float* pixelwise_mlp, int B, int N, int C, int G,
                               int D, int h, int w, int hs, int ws, float* cost_out, float* view_weights_out,
                               int* vw_argmax_out, float* similarity_out, void* stream) {
    if (!ref_nhwc || (!src_nhwc && !src_table) || !rel_proj || !depth_sample || !similarity_mlp || !cost_out) return PMN_ERR_ARG;
    if (!view_weights_in && (!pixelwise_mlp || !view_weights_out)) return PMN_ERR_ARG;
    if (B < 1 || N < 1 || D < 1 || h < 2 || w < 2 || hs < 2 || ws < 2 || vw_shift < 0 || vw_shift > 2) return PMN_ERR_ARG;
    if (D > PMN_MAX_DEPTH) return PMN_ERR_SHAPE;
    if ((h | w) & ((1 << vw_shift) - 1)) return PMN_ERR_ARG;
    GatherArgs a;
    memset(&a, 0, sizeof(a));
    a.ref = ref_nhwc;
    a.src = src_nhwc;
    a.src_tab = src_table;
    a.proj = rel_proj;
    a.depth = depth_sample;
    a.vw_in = view_weights_in;
    a.mlp_a = similarity_mlp;
    a.mlp_b = pixelwise_mlp;
    a.vw_out = view_weights_out;
    a.vw_argmax = vw_argmax_out;
    a.sim_out = similarity_out;
    a.out = cost_out;
    a.B = B; a.N = N; a.D = D; a.h = h; a.w = w; a.hs = hs; a.ws = ws;
    a.vw_shift = vw_shift;
#ifdef PMN_EXPERIMENTAL
    // libpmn_hip_experimental.so only (`make EXPERIMENTAL=1`, include/pmn_hip_experimental.h): kernel family by pmn_set_tuning
    // key 1 -- bit 0 = windowed kernels where they cover the shape (the lane = item engine, or with bit 4 the first windowed
    // form), bits 2 / 3 keep the streaming kernel for the PixelwiseNet / the known-weights launches, bit 5 = the tile-window kernel.
    // The product library has none of this: every launch is the streaming kernel below.
    const int flags = src_table ? 0 : pmn_gather_flags();  // (the research families read the stacked layout only)
    const bool pixelwise = view_weights_in == nullptr;
    if (flags & 64) {  // round 4: correlate-then-interpolate on the fp32 matrix cores (experimental/corr_mfma.hip)
        const int rc = pmn_launch_corr_mfma(a, C, G, pixelwise, (hipStream_t)stream);
        if (rc != PMN_ERR_SHAPE) return rc;
    }
    if ((flags & 32) && !pixelwise) {
        const int rc = pmn_launch_gather_tile(a, C, G, (hipStream_t)stream);
        if (rc != PMN_ERR_SHAPE) return rc;
    }
    if ((flags & 1) && !(flags & (pixelwise ? 4 : 8))) {
        const int rc = (flags & 16) ? pmn_launch_gather_win(a, C, G, pixelwise, (hipStream_t)stream)
                                    : pmn_launch_gather_lane(a, C, G, pixelwise, (hipStream_t)stream);
        if (rc != PMN_ERR_SHAPE) return rc;
    }
#endif
    if (view_weights_in) return dispatch_gather<MODE_VIEWS>(a, C, G, (hipStream_t)stream);
#if PMN_PW != 0
    if (C == 64 && G == 8 && D <= 64) return launch_pixelwise_wave<PMN_PW>(a, (hipStream_t)stream);  // the cascade's stage 3
#endif
    return dispatch_gather<MODE_PIXELWISE>(a, C, G, (hipStream_t)stream);
}

extern "C" int pmn_warp_correlate(const float* ref_nhwc, const float* src_nhwc, const float* rel_proj,
                                  const float* depth_sample, const float* view_weights_in, int vw_shift,
                                  const float* similarity_mlp, const float* pixelwise_mlp, int B, int N, int C, int G,
                                  int D, int h, int w, int hs, int ws, float* cost_out, float* view_weights_out,
                                  int* vw_argmax_out, float* similarity_out, void* stream) {
    return warp_correlate_impl(ref_nhwc, src_nhwc, nullptr, rel_proj, depth_sample, view_weights_in, vw_shift, similarity_mlp,
                               pixelwise_mlp, B, N, C, G, D, h, w, hs, ws, cost_out, view_weights_out, vw_argmax_out, similarity_out,
                               stream);
}

extern "C" int pmn_warp_correlate_views(const float* ref_nhwc, const void* src_view_table, const float* rel_proj,
                                        const float* depth_sample, const float* view_weights_in, int vw_shift,
                                        const float* similarity_mlp, const float* pixelwise_mlp, int B, int N, int C, int G,
                                        int D, int h, int w, int hs, int ws, float* cost_out, float* view_weights_out,
                                        int* vw_argmax_out, float* similarity_out, void* stream) {
    if (!src_view_table) return PMN_ERR_ARG;
    return warp_correlate_impl(ref_nhwc, nullptr, static_cast<const unsigned long long*>(src_view_table), rel_proj, depth_sample,
                               view_weights_in, vw_shift, similarity_mlp, pixelwise_mlp, B, N, C, G, D, h, w, hs, ws, cost_out,
                               view_weights_out, vw_argmax_out, similarity_out, stream);
}

extern "C" int pmn_feature_weight(const float* ref_nhwc, const float* eval_offsets, const int* eval_table_host,
                                  const float* mlp, int B, int C, int G, int K, int h, int w, float* out_feature_weight,
                                  void* stream) {
    if (!ref_nhwc || !eval_offsets || !eval_table_host || !mlp || !out_feature_weight) return PMN_ERR_ARG;
    if (B < 1 || h < 2 || w < 2) return PMN_ERR_ARG;
    if (K < 1 || K > PMN_MAX_NEIGHBORS) return PMN_ERR_SHAPE;
    GatherArgs a;
    memset(&a, 0, sizeof(a));
    a.ref = ref_nhwc;
    a.src = ref_nhwc;
    a.offsets = eval_offsets;
    a.mlp_a = mlp;
    a.out = out_feature_weight;
    a.B = B; a.N = 1; a.D = K; a.h = h; a.w = w; a.hs = h; a.ws = w;
    for (int i = 0; i < 2 * K; ++i) a.table[i] = eval_table_host[i];
    return dispatch_gather<MODE_NEIGHBOR>(a, C, G, (hipStream_t)stream);
}
