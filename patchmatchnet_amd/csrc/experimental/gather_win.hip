// gather_win.hip -- the fused warp + bilinear gather + group-wise correlation kernel, windowed form.
//
// Same arithmetic, same results (bit for bit) as the streaming kernel of gather_corr.hip; what changes is where the bilinear
// taps come from.  The streaming kernel issues 4 x global_load_dwordx4 per (pixel, hypothesis, channel quad): 29.5 GB of taps per
// depth map through the 64 B/clk/CU vector-memory path (the measured limiter: 63-77 % of that path's peak, 9 % of the HBM
// roofline).  Neighbouring pixels and consecutive hypotheses read almost the same source texels, so here every WAVE stages the
// bounding box of its taps -- a "window" of the source map -- into its own slice of LDS once per (view, 8 hypotheses) and takes
// the taps from there (ds_read_b128: 256 B/clk/CU, no tags, no address coalescer): 7-20 taps per staged texel on the cascade's
// hypotheses (tests/studies/footprint_study.py).
//
// Reference: models/module.py:130-181 (differentiable_warping), models/patchmatch.py:192-217 (group correlation, view
// aggregation), :570 (SimilarityNet MLP), :695-702 (PixelwiseNet).
//
// Mapping (wave64, 256-thread workgroups):
//   * a wave owns a 16-channel SLICE of the feature maps (C/16 slices: groups never straddle a slice, so slices are independent
//     until the pointwise MLP) and a 64-item set: MODE_VIEWS 16x4 pixels x one hypothesis per step (8 steps = the workgroup's
//     hypothesis chunk, blockIdx.y), MODE_PIXELWISE 16 pixels x 4 hypotheses per step.  A LANE is one (pixel, hypothesis)
//     item: it projects its own tap record (no cross-lane broadcast), reads the 4 corners x 4 channel quads of its slice
//     (16 x ds_read_b128, addresses = one base per window row + immediates), blends, and reduces its groups in-lane (no DPP).
//   * the window is wave-private: bounding box by a DPP/readlane reduction, staging by the wave's own global_load_dwordx4 ->
//     ds_write_b128, consumption by the same wave -- LDS is in order per wave, so there is NO workgroup barrier in the view
//     loop and the waves of a CU drift apart (one stages while another blends).
//   * a tap outside the window (bounding box larger than the wave's LDS share: depth discontinuities, wide baselines) is
//     fetched from HBM/L1 by that lane alone (exec-masked), so any input is handled; only the speed depends on the data.
//   * bank conflicts: ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27} {4-11,16-19,28-31} (+32)
//     (MI355X_MICROARCH.md, LDS); each group is one 16-pixel row of items and lane i of a row walks its four channel quads in
//     the order (j + i/4) % 4, so the 16 lanes of a group touch 16 different 16-byte slots whenever pixels i..i+3 of a row hit
//     4 consecutive texels -- independent of the window's width and alignment.
#include "../gather_common.hpp"

typedef float pmn_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char pmn_lds_char;        // LDS (ds_read / ds_write, never flat)
typedef const __attribute__((address_space(1))) char pmn_glb_char;  // global (global_load, never flat)
#define PMN_LDS_F4(p) (*reinterpret_cast<__attribute__((address_space(3))) pmn_f4*>(p))
#define PMN_GLB_F4(p) (*reinterpret_cast<const __attribute__((address_space(1))) pmn_f4*>(p))

struct WinGeom {
    int bx0, by0, bw, bh;  // wave-uniform
};

// ---- wave-level min / max (all 64 lanes active) --------------------------------------------------------------------------
template <bool MAX>
__device__ __forceinline__ int wave_minmax(int v) {
#define PMN_STEP(ctrl)                                                         \
    {                                                                          \
        const int o = __builtin_amdgcn_update_dpp(v, v, ctrl, 0xF, 0xF, false); \
        v = MAX ? max(v, o) : min(v, o);                                       \
    }
    PMN_STEP(0xB1)   // quad_perm [1,0,3,2]
    PMN_STEP(0x4E)   // quad_perm [2,3,0,1]
    PMN_STEP(0x141)  // row_half_mirror
    PMN_STEP(0x140)  // row_mirror
#undef PMN_STEP
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return MAX ? max(max(a, b), max(c, d)) : min(min(a, b), min(c, d));
}

// lane -> (row 0..3, column 0..15) of the wave's item set so that every ds_read_b128 lane group is one row
__device__ __forceinline__ void lane_to_item(int lane, int& row, int& col) {
    const int l = lane & 31;
    int g, i;
    if (l < 4) { g = 0; i = l; }
    else if (l < 12) { g = 1; i = l - 4; }
    else if (l < 16) { g = 0; i = l - 8; }
    else if (l < 20) { g = 1; i = l - 8; }
    else if (l < 28) { g = 0; i = l - 12; }
    else { g = 1; i = l - 16; }
    row = (lane >> 5) * 2 + g;
    col = i;
}

// One channel quad of one item: bilinear blend of the four corners, then the dot product with the reference quad -- the
// operation order of gather_corr.hip's blend_corners (per channel ((t00*w00 + t01*w01) + t10*w10) + t11*w11).
__device__ __forceinline__ float blend_dot(const pmn_f4 t00, const pmn_f4 t01, const pmn_f4 t10, const pmn_f4 t11,
                                           const pmn_f4 w4, const pmn_f4 refq) {
    const pmn_f2 wa = {w4.x, w4.x}, wb = {w4.y, w4.y}, wc = {w4.z, w4.z}, wd = {w4.w, w4.w};
    pmn_f2 lo = pmn_f2{t00.x, t00.y} * wa;
    pmn_f2 hi = pmn_f2{t00.z, t00.w} * wa;
    lo = __builtin_elementwise_fma(pmn_f2{t01.x, t01.y}, wb, lo);
    hi = __builtin_elementwise_fma(pmn_f2{t01.z, t01.w}, wb, hi);
    lo = __builtin_elementwise_fma(pmn_f2{t10.x, t10.y}, wc, lo);
    hi = __builtin_elementwise_fma(pmn_f2{t10.z, t10.w}, wc, hi);
    lo = __builtin_elementwise_fma(pmn_f2{t11.x, t11.y}, wd, lo);
    hi = __builtin_elementwise_fma(pmn_f2{t11.z, t11.w}, wd, hi);
    return fmaf(hi.y, refq.w, fmaf(hi.x, refq.z, fmaf(lo.y, refq.y, lo.x * refq.x)));
}

using PosePix = PmnPose;  // the reference's own warp chain (pmn_common.hpp): round 4 replaced the v_rcp projection everywhere

__device__ __forceinline__ PosePix make_pose_pix(const float* __restrict__ P, float xf, float yf, int h, int w) {
    return pmn_make_pose(P, xf, yf, h, w);
}

// Tap record of one item; returns false (zero weights, corner (0,0)) for an inactive lane or a hypothesis behind the source
// camera (reference sentinel, module.py:166-169).
__device__ __forceinline__ bool project_item(const PosePix& q, float dep, bool active, int h, int w, int hs, int ws, PmnTapsXY& t) {
    t.x0 = 0;
    t.y0 = 0;
    t.w00 = t.w01 = t.w10 = t.w11 = 0.0f;
    bool tv = false;
    if (active) {
        float ix, iy;
        if (pmn_pose_position(q, dep, h, w, hs, ws, ix, iy)) {
            t = pmn_make_taps_xy(ix, iy, hs, ws);
            tv = true;
        }
    }
    return tv;
}

// Window of one round: bounding box of the north-west corners of the wave's first / last hypotheses (+1 for the south-east
// corners).  Positions move monotonically with depth along the epipolar line and the hypotheses of a pixel are sorted, so the
// end points bound the round -- and where they do not (arbitrary caller data), the stragglers take the global path.  A box
// larger than the wave's LDS share is cut down around the position of the item in the middle of the set.
__device__ __forceinline__ WinGeom make_window(const PmnTapsXY& ta, bool va, const PmnTapsXY& tb, bool vb, int cap_texels,
                                               int hs, int ws) {
    const int BIG = 1 << 20;
    int lo_x = BIG, hi_x = -BIG, lo_y = BIG, hi_y = -BIG;
    if (va) { lo_x = ta.x0; hi_x = ta.x0; lo_y = ta.y0; hi_y = ta.y0; }
    if (vb) { lo_x = min(lo_x, tb.x0); hi_x = max(hi_x, tb.x0); lo_y = min(lo_y, tb.y0); hi_y = max(hi_y, tb.y0); }
    // centre candidate: the middle of lane 52's segment (row 2, column 8 of the item set) when that lane is live
    const int cxl = (va && vb) ? ((ta.x0 + tb.x0) >> 1) : (va ? ta.x0 : (vb ? tb.x0 : -1));
    const int cyl = (va && vb) ? ((ta.y0 + tb.y0) >> 1) : (va ? ta.y0 : (vb ? tb.y0 : -1));
    int sx0 = wave_minmax<false>(lo_x), sx1 = wave_minmax<true>(hi_x);
    int sy0 = wave_minmax<false>(lo_y), sy1 = wave_minmax<true>(hi_y);
    const int cx = __builtin_amdgcn_readlane(cxl, 52), cy = __builtin_amdgcn_readlane(cyl, 52);
    WinGeom g;
    if (sx0 > sx1) {  // no live item in this wave
        g.bx0 = 0; g.by0 = 0; g.bw = 2; g.bh = 2;
        return g;
    }
    g.bx0 = sx0;
    g.by0 = sy0;
    g.bw = sx1 - sx0 + 2;
    g.bh = sy1 - sy0 + 2;
    if (g.bw * g.bh > cap_texels) {
        const int bh = min(g.bh, 8);
        const int bw = max(min(g.bw, cap_texels / bh), 2);
        const int ccx = cx >= 0 ? cx : ((sx0 + sx1) >> 1), ccy = cy >= 0 ? cy : ((sy0 + sy1) >> 1);
        g.bx0 = min(max(ccx - (bw >> 1) + 1, sx0), sx1 + 2 - bw);
        g.by0 = min(max(ccy - (bh >> 1) + 1, sy0), sy1 + 2 - bh);
        g.bw = bw;
        g.bh = bh;
    }
    (void)hs; (void)ws;
    return g;
}

// Stage the window [by0, by0+bh) x [bx0, bx0+bw) of one source map's 16-channel slice into the wave's LDS region: texel-major,
// 64 B per texel.  A wave-instruction moves 16 texels of one window row (lane = texel * 4 + channel quad: 1 KB contiguous in
// LDS, 64-byte runs in HBM); segments are issued in batches of NB loads before the first ds_write.
template <int C, int NB>
__device__ __forceinline__ void stage_window(pmn_lds_char* win, pmn_glb_char* src_slice, const WinGeom& g, int ws, int lane) {
    const int nseg = (g.bw + 15) >> 4;
    const int total = g.bh * nseg;
    const int tcol = lane >> 2, quad = lane & 3;
    int r = 0, c0 = 0;
    for (int s0 = 0; s0 < total; s0 += NB) {
        pmn_f4 buf[NB];
        int rr = r, cc = c0;
        // loads are unconditional at clamped (always legal) positions so the batch stays in registers and carries no
        // branches; the stores are masked
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int col = min(cc + tcol, g.bw - 1), row = min(rr, g.bh - 1);
            const unsigned go = ((unsigned)((g.by0 + row) * ws + g.bx0 + col) * (unsigned)(C * 4)) + quad * 16u;
            buf[k] = PMN_GLB_F4(src_slice + go);
            cc += 16;
            if (cc >= g.bw) { cc = 0; ++rr; }
        }
        // all NB loads are issued before the first store (left alone hipcc sinks each load into its store's branch:
        // load -> vmcnt(0) -> ds_write, one memory round trip per 16 bytes)
#pragma unroll
        for (int k = 0; k < NB; ++k) asm volatile("" : "+v"(buf[k]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int col = c0 + tcol;
            if (s0 + k < total && col < g.bw) {
                PMN_LDS_F4(win + ((((unsigned)(r * g.bw + col)) << 6) + quad * 16u)) = buf[k];
            }
            c0 += 16;
            if (c0 >= g.bw) { c0 = 0; ++r; }
        }
    }
}

// The 16 corner quads of one item (4 channel quads in this lane's rotated order x 4 corners) and the four blended dot
// products.  `win` = the wave's LDS window, `src_slice` = the view's map + slice offset (global), `ofs[j]` = byte offset of
// the j-th quad this lane visits, `refq[j]` the matching reference quad.
template <int C>
__device__ __forceinline__ void gather_item16(pmn_lds_char* win, pmn_glb_char* src_slice, const WinGeom& g, const PmnTapsXY& t,
                                              bool tv, int ws, const unsigned (&ofs)[4], const pmn_f4 (&refq)[4],
                                              float (&dot)[4]) {
    const int lx = tv ? t.x0 - g.bx0 : 0, ly = tv ? t.y0 - g.by0 : 0;
    const bool inside = (unsigned)lx < (unsigned)(g.bw - 1) && (unsigned)ly < (unsigned)(g.bh - 1);
    pmn_f4 c[4][4];
    if (inside) {
        const unsigned aN = ((unsigned)(ly * g.bw + lx)) << 6, aS = aN + ((unsigned)g.bw << 6);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c[j][0] = PMN_LDS_F4(win + (aN + ofs[j]));
            c[j][1] = PMN_LDS_F4(win + (aN + ofs[j] + 64));
            c[j][2] = PMN_LDS_F4(win + (aS + ofs[j]));
            c[j][3] = PMN_LDS_F4(win + (aS + ofs[j] + 64));
        }
    } else {
        const unsigned go = (unsigned)(t.y0 * ws + t.x0) * (unsigned)(C * 4), rb = (unsigned)ws * (unsigned)(C * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c[j][0] = PMN_GLB_F4(src_slice + (go + ofs[j]));
            c[j][1] = PMN_GLB_F4(src_slice + (go + ofs[j]) + C * 4);
            c[j][2] = PMN_GLB_F4(src_slice + ((go + rb) + ofs[j]));
            c[j][3] = PMN_GLB_F4(src_slice + ((go + rb) + ofs[j]) + C * 4);
        }
    }
    const pmn_f4 w4 = {t.w00, t.w01, t.w10, t.w11};
#pragma unroll
    for (int j = 0; j < 4; ++j) dot[j] = blend_dot(c[j][0], c[j][1], c[j][2], c[j][3], w4, refq[j]);
}

// Group similarities of the slice from the four quad dot products.  CG = 4: quad (j + rot) % 4 IS group (j + rot) % 4 of the
// slice, sim[j] keeps the lane's rotated order.  CG = 8: quads {0,1} / {2,3} pair up; sim[0] is the group holding quad `rot`
// (slice group (rot >> 1) & 1), sim[1] the other one.  The two-term sums are commutative, so the value does not depend on
// the rotation and equals gather_corr.hip's pair-swap sum.
template <int CG>
__device__ __forceinline__ void group_sims(const float (&dot)[4], int rot, float (&sim)[16 / CG]) {
    if constexpr (CG == 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sim[j] = dot[j] * 0.25f;
    } else {
        const bool odd = rot & 1;
        const float a = dot[0] + (odd ? dot[3] : dot[1]);
        const float b = dot[2] + (odd ? dot[1] : dot[3]);
        sim[0] = a * 0.125f;
        sim[1] = b * 0.125f;
    }
}

// slice-local group index of sim[k] for a lane with rotation `rot`
template <int CG>
__device__ __forceinline__ int sim_group(int k, int rot) {
    if constexpr (CG == 4) return (k + rot) & 3;
    return ((rot >> 1) & 1) ^ k;
}

// ============================================================================================================================
// MODE_VIEWS: view weights known.  grid = (pixel tiles, hypothesis chunks of 8, batch).
// ============================================================================================================================
template <int C, int G, bool ROT>
__global__ __launch_bounds__(PMN_BLOCK, 3) void gather_win_views_kernel(const GatherArgs a, const int cap_bytes, const int dbg) {
    constexpr int NS = C / 16;         // channel slices = waves sharing a pixel sub-tile
    constexpr int NSUB = 4 / NS;       // 16x4-pixel sub-tiles per workgroup (stacked in y)
    constexpr int NPIXWG = 64 * NSUB;  // pixels per workgroup
    constexpr int CG = C / G;          // channels per correlation group
    constexpr int GPS = 16 / CG;       // groups per slice
    constexpr int DCH = 8;             // hypotheses per workgroup
    constexpr int NIT = DCH * NPIXWG / PMN_BLOCK;  // epilogue: (pixel, hypothesis) items per thread: 8 / 4 / 2
    constexpr int NI = NIT < 4 ? NIT : 4;
    static_assert(CG == 4 || CG == 8, "group size must be 4 or 8 channels");
    static_assert(NS == 1 || NS == 2 || NS == 4, "16, 32 or 64 channels");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int slice = wave % NS, sub = wave / NS;
    int irow, icol;
    lane_to_item(lane, irow, icol);
    const int rot = ROT ? (icol >> 2) & 3 : 0;

    const int N = a.N, h = a.h, w = a.w, hs = a.hs, ws = a.ws, D = a.D;
    const int hw = h * w;
    const int b = blockIdx.z;
    const int d0 = blockIdx.y * DCH;
    const int nd = min(DCH, D - d0);  // run-time on purpose: with a compile-time 8 hipcc merges the steps and spills
    const int ntx = (w + 15) >> 4;
    const int tile = pmn_xcd_tile(blockIdx.x, a.ntiles);
    const int ty = tile / ntx, tx = tile - ty * ntx;
    const int x = tx * 16 + icol, y = (ty * NSUB + sub) * 4 + irow;
    const bool ok = x < w && y < h;
    const int p = ok ? y * w + x : 0;

    pmn_lds_char* win = (pmn_lds_char*)(smem + wave * cap_bytes);
    const int cap_texels = cap_bytes >> 6;
    float* wlds = reinterpret_cast<float*>(smem + 4 * cap_bytes);
    for (int i = tid; i < PMN_MLP_FLOATS; i += PMN_BLOCK) wlds[i] = a.mlp_a[i];

    unsigned ofs[4];
    pmn_f4 refq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ofs[j] = (unsigned)((j + rot) & 3) * 16u;
        refq[j] = pmn_f4{0.f, 0.f, 0.f, 0.f};
        if (ok) refq[j] = PMN_GLB_F4((pmn_glb_char*)a.ref + (((size_t)b * hw + p) * (C * 4) + slice * 64 + ofs[j]));
    }
    float rdep[DCH];
#pragma unroll
    for (int s = 0; s < DCH; ++s) rdep[s] = (ok && s < nd) ? a.depth[((size_t)b * D + d0 + s) * hw + p] : 0.0f;

    float acc[DCH][GPS];
#pragma unroll
    for (int s = 0; s < DCH; ++s)
#pragma unroll
        for (int k = 0; k < GPS; ++k) acc[s][k] = 0.0f;

    const float xf = (float)x, yf = (float)y;
    const int wv = w >> a.vw_shift, hwv = (h >> a.vw_shift) * wv;
    const int vw_idx = (y >> a.vw_shift) * wv + (x >> a.vw_shift);

    for (int v = 0; v < N; ++v) {
        const PosePix q = make_pose_pix(a.proj + ((size_t)b * N + v) * 16, xf, yf, h, w);
        pmn_glb_char* src_slice = (pmn_glb_char*)a.src + (((size_t)(v * a.B + b) * hs * ws) * (C * 4) + slice * 64);
        const float vw = ok ? a.vw_in[((size_t)b * N + v) * hwv + vw_idx] : 0.0f;
        PmnTapsXY ta, tb;
        const bool va = project_item(q, rdep[0], ok, h, w, hs, ws, ta);
        float dlast = rdep[0];
#pragma unroll
        for (int s = 1; s < DCH; ++s)
            if (s < nd) dlast = rdep[s];
        const bool vb = project_item(q, dlast, ok, h, w, hs, ws, tb);
        WinGeom g;
        if (dbg & 16) {  // ablation: no bounding-box reduction
            g.bx0 = max(min(ta.x0 - 8, ws - 34), 0); g.by0 = max(min(ta.y0 - 2, hs - 6), 0); g.bw = min(32, ws); g.bh = min(6, hs);
            g.bx0 = __builtin_amdgcn_readfirstlane(g.bx0); g.by0 = __builtin_amdgcn_readfirstlane(g.by0);
        } else {
            g = make_window(ta, va, tb, vb, cap_texels, hs, ws);
        }
        if (!(dbg & 1)) stage_window<C, 8>(win, src_slice, g, ws, lane);  // ablation bit 0: no staging
#pragma unroll
        for (int s = 0; s < DCH; ++s) {
            if (s < nd) {
                PmnTapsXY t = ta;
                bool tv = va;
                if (!(dbg & 4)) tv = project_item(q, rdep[s], ok, h, w, hs, ws, t);  // ablation bit 2: no per-item projection
                float dot[4], sim[GPS];
                if (dbg & 2) {  // ablation bit 1: no taps
                    dot[0] = t.w00; dot[1] = t.w01; dot[2] = t.w10; dot[3] = t.w11;
                } else
                gather_item16<C>(win, src_slice, g, t, tv, ws, ofs, refq, dot);
                group_sims<CG>(dot, rot, sim);
#pragma unroll
                for (int k = 0; k < GPS; ++k) acc[s][k] = mul_add_unfused(acc[s][k], sim[k], vw);
            }
            __builtin_amdgcn_sched_barrier(0);  // one item at a time: left alone hipcc hoists all 8 projections and 128 loads
        }
    }

    // ---- hand-over to the pointwise MLP: the G groups of an item live in NS different waves -----------------------------------
    __syncthreads();  // every wave is done with its window: the region is re-used for the similarity tile
    float* simt = reinterpret_cast<float*>(smem);  // [G][DCH][NPIXWG]
    {
        const int pid = sub * 64 + irow * 16 + icol;
#pragma unroll
        for (int s = 0; s < DCH; ++s)
#pragma unroll
            for (int k = 0; k < GPS; ++k) {
                const int gg = slice * GPS + sim_group<CG>(k, rot);
                simt[(gg * DCH + s) * NPIXWG + pid] = acc[s][k];
            }
    }
    __syncthreads();
    {
        const int pid = tid % NPIXWG, dpart = tid / NPIXWG;
        const int ex = tx * 16 + (pid & 15), ey = ty * NSUB * 4 + (pid >> 4);
        if (ex >= w || ey >= h) return;
        const int ep = ey * w + ex;
        float wtot = 1e-5f;
        const int vwi = (ey >> a.vw_shift) * wv + (ex >> a.vw_shift);
        for (int v = 0; v < N; ++v) wtot += a.vw_in[((size_t)b * N + v) * hwv + vwi];
        float xin[NIT][G], o[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int dl = dpart * NIT + k;
#pragma unroll
            for (int gI = 0; gI < G; ++gI) xin[k][gI] = simt[(gI * DCH + dl) * NPIXWG + pid] / wtot;
        }
        if (dbg & 8) {  // ablation bit 3: no pointwise MLP
#pragma unroll
            for (int k = 0; k < NIT; ++k) o[k] = xin[k][0];
        } else
        mlp_items<G, NIT, NI>(wlds, xin, o);
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int d = d0 + dpart * NIT + k;
            if (d < D) {
                if (a.sim_out) {
#pragma unroll
                    for (int gI = 0; gI < G; ++gI) a.sim_out[(((size_t)b * G + gI) * D + d) * hw + ep] = xin[k][gI];
                }
                a.out[((size_t)b * hw + ep) * D + d] = o[k];  // cost is hypothesis-last [B,h,w,D]
            }
        }
    }
}

// ============================================================================================================================
// MODE_PIXELWISE: the view weights are computed here (PixelwiseNet + max over ALL hypotheses of a view, stage-3 iteration 1).
// grid = (pixel tiles, 1, batch).  A workgroup owns NPIX = 1024/C pixels (rows of 16) x all D hypotheses; per view the four
// waves (slice x pixel row) fill the similarity tile [G][D][NPIX] in LDS from their windows -- 8 hypotheses per window, lanes =
// 16 pixels x 4 hypotheses, 2 items per lane and round -- without any barrier; then the item role (thread <-> pixel, a few
// hypotheses) runs PixelwiseNet, the max over D (64-bit LDS atomic max, value | ~d: first arg-max) and the weighted sums.  This
// is gather_corr.hip's MODE_PIXELWISE with phases A/B replaced: 3 barriers per view instead of 6 + 2 per 32 hypotheses.
// ============================================================================================================================
template <int C, int G, bool ROT>
__global__ __launch_bounds__(PMN_BLOCK) void gather_win_pixelwise_kernel(const GatherArgs a, const int cap_bytes, const int dbg) {
    constexpr int NS = C / 16;
    constexpr int NSUB = 4 / NS;     // 16-pixel rows per workgroup
    constexpr int NPIX = 16 * NSUB;  // pixels per workgroup
    constexpr int CG = C / G;
    constexpr int GPS = 16 / CG;
    constexpr int PAD = 32 / G;
    constexpr int DT = PMN_MAX_DEPTH;
    constexpr int DSTEP = PMN_BLOCK / NPIX;  // item role: a thread's hypotheses are dA0, dA0 + DSTEP, ...
    constexpr int NIT = DT / DSTEP;
    constexpr int NI = NIT < 2 ? NIT : 2;
    static_assert(CG == 4 || CG == 8, "group size must be 4 or 8 channels");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int slice = wave % NS, prow = wave / NS;
    int dsub, icol;
    lane_to_item(lane, dsub, icol);
    const int rot = ROT ? (icol >> 2) & 3 : 0;

    const int N = a.N, h = a.h, w = a.w, hs = a.hs, ws = a.ws, D = a.D;
    const int hw = h * w;
    const int b = blockIdx.z;
    const int ntx = (w + 15) >> 4;
    const int tile = pmn_xcd_tile(blockIdx.x, a.ntiles);
    const int ty = tile / ntx, tx = tile - ty * ntx;
    const int SS = NPIX * D + PAD;

    pmn_lds_char* win = (pmn_lds_char*)(smem + wave * cap_bytes);
    const int cap_texels = cap_bytes >> 6;
    float* simt = reinterpret_cast<float*>(smem + 4 * cap_bytes);                                // [G][SS]
    float* wlds_a = simt + ((G * SS + 3) & ~3);                                                  // similarity_net
    float* wlds_b = wlds_a + PMN_MLP_FLOATS;                                                     // pixel_wise_net
    unsigned long long* vwkey = reinterpret_cast<unsigned long long*>(wlds_b + PMN_MLP_FLOATS);  // [NPIX]
    for (int i = tid; i < PMN_MLP_FLOATS; i += PMN_BLOCK) {
        wlds_a[i] = a.mlp_a[i];
        wlds_b[i] = a.mlp_b[i];
    }

    // lane role: pixel (x, y), hypotheses dsub + 4 * k
    const int x = tx * 16 + icol, y = ty * NSUB + prow;
    const bool ok = x < w && y < h;
    const int p = ok ? y * w + x : 0;
    const int pid = prow * 16 + icol;
    unsigned ofs[4];
    pmn_f4 refq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ofs[j] = (unsigned)((j + rot) & 3) * 16u;
        refq[j] = pmn_f4{0.f, 0.f, 0.f, 0.f};
        if (ok) refq[j] = PMN_GLB_F4((pmn_glb_char*)a.ref + (((size_t)b * hw + p) * (C * 4) + slice * 64 + ofs[j]));
    }
    const float xf = (float)x, yf = (float)y;

    // item role: a fixed pixel of the tile, hypotheses dA0 + j * DSTEP
    const int pixA = tid % NPIX, dA0 = tid / NPIX;
    const int xA = tx * 16 + (pixA & 15), yA = ty * NSUB + (pixA >> 4);
    const bool okA = xA < w && yA < h;
    const int pA = okA ? yA * w + xA : 0;
    float ssum[NIT][G];
#pragma unroll
    for (int j = 0; j < NIT; ++j)
#pragma unroll
        for (int gI = 0; gI < G; ++gI) ssum[j][gI] = 0.0f;
    float wsum = 1e-5f;
    __syncthreads();  // MLP weights visible

    for (int v = 0; v < N; ++v) {
        if (tid < NPIX) vwkey[tid] = 0ull;
        const PosePix q = make_pose_pix(a.proj + ((size_t)b * N + v) * 16, xf, yf, h, w);
        pmn_glb_char* src_slice = (pmn_glb_char*)a.src + (((size_t)(v * a.B + b) * hs * ws) * (C * 4) + slice * 64);
        for (int dc0 = 0; dc0 < D; dc0 += 8) {
            const int da = dc0 + dsub, db = da + 4;
            const bool la = ok && da < D, lb = ok && db < D;
            const float depa = la ? a.depth[((size_t)b * D + da) * hw + p] : 0.0f;
            const float depb = lb ? a.depth[((size_t)b * D + db) * hw + p] : 0.0f;
            PmnTapsXY ta, tb;
            const bool va = project_item(q, depa, la, h, w, hs, ws, ta);
            const bool vb = project_item(q, depb, lb, h, w, hs, ws, tb);
            const WinGeom g = make_window(ta, va, tb, vb, cap_texels, hs, ws);
            if (!(dbg & 1)) stage_window<C, 8>(win, src_slice, g, ws, lane);
            {
                float dot[4], sim[GPS];
                if (dbg & 2) {
                    dot[0] = ta.w00; dot[1] = ta.w01; dot[2] = ta.w10; dot[3] = ta.w11;
                } else
                gather_item16<C>(win, src_slice, g, ta, va, ws, ofs, refq, dot);
                group_sims<CG>(dot, rot, sim);
                if (da < D) {
#pragma unroll
                    for (int k = 0; k < GPS; ++k) simt[(slice * GPS + sim_group<CG>(k, rot)) * SS + da * NPIX + pid] = sim[k];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (dc0 + 4 < D) {
                float dot[4], sim[GPS];
                if (dbg & 2) {
                    dot[0] = tb.w00; dot[1] = tb.w01; dot[2] = tb.w10; dot[3] = tb.w11;
                } else
                gather_item16<C>(win, src_slice, g, tb, vb, ws, ofs, refq, dot);
                group_sims<CG>(dot, rot, sim);
                if (db < D) {
#pragma unroll
                    for (int k = 0; k < GPS; ++k) simt[(slice * GPS + sim_group<CG>(k, rot)) * SS + db * NPIX + pid] = sim[k];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();  // this view's similarity tile is complete
        // PixelwiseNet + max over D (first arg-max on ties through the ~d low word)
        {
            unsigned long long best = 0ull;
#pragma unroll
            for (int c = 0; c < NIT / NI; ++c) {
                static_assert(NI == 2, "PixelwiseNet is evaluated for one pair of items at a time");
                float r[NI];
                pmn_f2 xq[1][G], rq[1];
                {
                    const int da = min(dA0 + (c * NI) * DSTEP, D - 1), db = min(dA0 + (c * NI + 1) * DSTEP, D - 1);
#pragma unroll
                    for (int g = 0; g < G; ++g) xq[0][g] = pmn_f2{simt[g * SS + da * NPIX + pixA], simt[g * SS + db * NPIX + pixA]};
                }
                if (dbg & 8) {
                    rq[0] = xq[0][0];
                } else {
                    mlp_pairs_from_lds<G, 1>(wlds_b, xq, rq);
                }
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
            for (int gI = 0; gI < G; ++gI) ssum[j][gI] = mul_add_unfused(ssum[j][gI], simt[gI * SS + d * NPIX + pixA], vwp);
        }
        wsum += vwp;
        if (tid < NPIX && okA) {
            const size_t o = ((size_t)b * N + v) * hw + pA;
            a.vw_out[o] = vwp;
            if (a.vw_argmax) a.vw_argmax[o] = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
        }
        __syncthreads();  // vwkey / simt are rewritten by the next view
    }

    if (!okA) return;
    float o[NIT];
#pragma unroll
    for (int j = 0; j < NIT; ++j)
#pragma unroll
        for (int gI = 0; gI < G; ++gI) ssum[j][gI] = ssum[j][gI] / wsum;
    mlp_items<G, NIT, NI>(wlds_a, ssum, o);
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
        const int d = dA0 + j * DSTEP;
        if (d < D) {
            if (a.sim_out) {
#pragma unroll
                for (int gI = 0; gI < G; ++gI) a.sim_out[(((size_t)b * G + gI) * D + d) * hw + pA] = ssum[j][gI];
            }
            a.out[((size_t)b * hw + pA) * D + d] = o[j];  // cost is hypothesis-last [B,h,w,D]
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------

static int g_win_cap_bytes = 12 * 1024;  // LDS window per wave, MODE_VIEWS
static int g_win_cap_pix_bytes = 8 * 1024;  // LDS window per wave, MODE_PIXELWISE (16 x 1 pixel tiles: smaller boxes)
static int g_win_dbg = 0;                 // timing ablations (results are then meaningless), key 3
static int g_win_flags = 0;              // pmn_set_tuning key 1; default = the fastest measured family per launch (profiles/)

int pmn_gather_flags() { return g_win_flags; }

extern "C" int pmn_set_tuning(int key, int value) {
    if (key >= 4 && key <= 6) return pmn_lane_set_tuning(key, value);
    if (key == 10) return pmn_tile_set_tuning(key, value);
    if (key == 0) {
        if (value < 1024 || value > 40 * 1024 || (value & 1023)) return PMN_ERR_ARG;
        g_win_cap_bytes = value;
        return PMN_OK;
    }
    if (key == 1) {
        g_win_flags = value;
        return PMN_OK;
    }
    if (key == 3) {
        g_win_dbg = value;
        return PMN_OK;
    }
    if (key == 2) {
        if (value < 1024 || value > 40 * 1024 || (value & 1023)) return PMN_ERR_ARG;
        g_win_cap_pix_bytes = value;
        return PMN_OK;
    }
    return PMN_ERR_ARG;
}

template <int C, int G, bool ROT>
static int launch_views(GatherArgs& a, hipStream_t stream) {
    constexpr int NS = C / 16, NSUB = 4 / NS, NPIXWG = 64 * NSUB;
    const int ntx = (a.w + 15) / 16, nty = (a.h + 4 * NSUB - 1) / (4 * NSUB);
    a.ntiles = ntx * nty;
    const int cap = g_win_cap_bytes;
    size_t simt = (size_t)G * 8 * NPIXWG * 4;
    size_t lds = (size_t)4 * cap;
    if (lds < simt) return PMN_ERR_SHAPE;  // the similarity tile re-uses the window region
    lds += PMN_MLP_FLOATS * 4;
    lds = (lds + 15) & ~(size_t)15;
    auto kern = gather_win_views_kernel<C, G, ROT>;
    static size_t lds_set = 0;  // per instantiation: largest dynamic-LDS size the attribute has been raised to
    if (lds > 48 * 1024 && lds > lds_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return PMN_ERR_LAUNCH;
        lds_set = lds;
    }
    PMN_LAUNCH(kern, dim3(a.ntiles, (a.D + 7) / 8, a.B), dim3(PMN_BLOCK), lds, stream, a, cap, g_win_dbg);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

template <int C, int G>
static int dispatch_views(GatherArgs& a, hipStream_t stream) {
    return (g_win_flags & 2) ? launch_views<C, G, false>(a, stream) : launch_views<C, G, true>(a, stream);
}

template <int C, int G, bool ROT>
static int launch_pixelwise(GatherArgs& a, hipStream_t stream) {
    constexpr int NS = C / 16, NSUB = 4 / NS, NPIX = 16 * NSUB, PAD = 32 / G;
    const int ntx = (a.w + 15) / 16, nty = (a.h + NSUB - 1) / NSUB;
    a.ntiles = ntx * nty;
    const int cap = g_win_cap_pix_bytes;
    const int SS = NPIX * a.D + PAD;
    size_t lds = (size_t)4 * cap + (size_t)((G * SS + 3) & ~3) * 4 + 2 * PMN_MLP_FLOATS * 4 + NPIX * 8;
    lds = (lds + 15) & ~(size_t)15;
    if (lds > 160 * 1024) return PMN_ERR_SHAPE;
    auto kern = gather_win_pixelwise_kernel<C, G, ROT>;
    static size_t lds_set = 0;
    if (lds > 48 * 1024 && lds > lds_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return PMN_ERR_LAUNCH;
        lds_set = lds;
    }
    PMN_LAUNCH(kern, dim3(a.ntiles, 1, a.B), dim3(PMN_BLOCK), lds, stream, a, cap, g_win_dbg);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

template <int C, int G>
static int dispatch_pixelwise(GatherArgs& a, hipStream_t stream) {
    return (g_win_flags & 2) ? launch_pixelwise<C, G, false>(a, stream) : launch_pixelwise<C, G, true>(a, stream);
}

int pmn_launch_gather_win(GatherArgs& a, int C, int G, bool pixelwise, hipStream_t stream) {
    if (pixelwise) {
        if (C == 64 && G == 8) return dispatch_pixelwise<64, 8>(a, stream);
        if (C == 32 && G == 8) return dispatch_pixelwise<32, 8>(a, stream);
        if (C == 16 && G == 4) return dispatch_pixelwise<16, 4>(a, stream);
        return PMN_ERR_SHAPE;
    }
    if (C == 64 && G == 8) return dispatch_views<64, 8>(a, stream);
    if (C == 32 && G == 8) return dispatch_views<32, 8>(a, stream);
    if (C == 16 && G == 4) return dispatch_views<16, 4>(a, stream);
    return PMN_ERR_SHAPE;
}
