// gather_lane.hip -- the fused warp + bilinear gather + group-wise correlation kernel: lane = item, wave-autonomous form.
//
// What the counters said about the two earlier forms (profiles/r02_*): the kernel is bound by VALU ISSUE -- every VALU
// instruction occupies its SIMD for 4 cycles -- not by HBM, L1 or LDS: gather_corr.hip spends 217 VALU instructions per 1024
// gathered channel samples (5 DPP broadcasts + address + reduction glue per 12 useful ones, replicated over the C/4 lanes of an
// item), the first windowed kernel (gather_win.hip) 258 (a projection per item and 16-channel slice, a barrier-fenced
// exchange + scalar MLP).  The useful work is ~50 (32 packed blends + 16 dot FMAs per 16 channels).  This form cuts the glue:
//
//   * a LANE is one (pixel, hypothesis) item for ALL C channels: it projects its tap record once and walks the C/16 channel
//     slices one after the other, so nothing is broadcast, reduced across lanes or recomputed per slice;
//   * a WAVE is autonomous: 16x4 pixels x DCH hypotheses (8 at C = 16, else 4), its own LDS window of the source map per
//     (view, slice) -- bounding box of its taps, staged by its own loads, consumed by ds_read_b128 (gather_win.hip explains the
//     window, the per-lane global fallback for taps outside it and the bank-conflict-free quad rotation) -- and, because a lane
//     ends up holding all G group similarities of its items, the pointwise MLP runs IN THE LANE: no LDS exchange, no
//     __syncthreads anywhere in the kernel.  The 4 waves of a workgroup are 4 independent units (neighbouring hypothesis
//     chunks / tiles: they share reference pixels and window texels in L1/L2);
//   * the MLPs evaluate two items per instruction (v_pk_fma_f32 with the weight broadcast through op_sel), the tap weights
//     of interior items skip the border selects (wave-uniform fast path), packed accumulation over views.
//
// Two epilogues on the same engine:
//   MODE_AGG  known view weights: sum over views of similarity * weight in registers, / weight sum, SimilarityNet MLP -> cost.
//   MODE_VW   PixelwiseNet (models/patchmatch.py:695-702): per view the lane's similarities go through PixelwiseNet + sigmoid,
//             the max over the lane's hypotheses is merged into a per-(pixel, view) 64-bit key (value | ~d: first arg-max) with
//             one global atomic max.  The first iteration of stage 3 is then pass 1 (MODE_VW) -> unpack keys -> pass 2
//             (MODE_AGG with the weights just computed): the similarities are gathered twice, which costs less than holding
//             [G][64][pixels] tiles in LDS behind 3 barriers per view (gather_corr.hip, gather_win.hip).
//
// Arithmetic and operation order are those of gather_corr.hip (pinned against the oracle / the reference's golden tensors), so
// all three forms agree bit for bit (tests/test_gather_win.py).
// Reference: models/module.py:130-181, models/patchmatch.py:192-217, :570, :695-702.
#include "../gather_common.hpp"

typedef float pmn_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char pmn_lds_char;        // LDS (ds_read / ds_write, never flat)
typedef const __attribute__((address_space(1))) char pmn_glb_char;  // global (global_load, never flat)
#define PMN_LDS_F4(p) (*reinterpret_cast<__attribute__((address_space(3))) pmn_f4*>(p))
#define PMN_LDS_F(p) (*reinterpret_cast<__attribute__((address_space(3))) float*>(p))
#define PMN_GLB_F4(p) (*reinterpret_cast<const __attribute__((address_space(1))) pmn_f4*>(p))

enum { MODE_AGG = 0, MODE_VW = 1 };

struct LaneWin {
    int bx0, by0, bw, bh;  // wave-uniform window geometry (texels)
};

template <bool MAX>
__device__ __forceinline__ int lane_wave_minmax(int v) {
#define PMN_STEP(ctrl)                                                          \
    {                                                                           \
        const int o = __builtin_amdgcn_update_dpp(v, v, ctrl, 0xF, 0xF, false); \
        v = MAX ? max(v, o) : min(v, o);                                        \
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

// lane -> (row 0..3, column 0..15) of the wave's 16x4 pixel tile so that every ds_read_b128 lane group
// ({0-3,12-15,20-27} {4-11,16-19,28-31} (+32), MI355X_MICROARCH.md) is one row of 16 pixels
__device__ __forceinline__ void lane_tile_pos(int lane, int& row, int& col) {
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

// blend of the four corners of one channel quad + dot product with the reference quad (gather_corr.hip's order)
__device__ __forceinline__ float lane_blend_dot(const pmn_f4 t00, const pmn_f4 t01, const pmn_f4 t10, const pmn_f4 t11,
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

using LanePose = PmnPose;  // the reference's own warp chain (pmn_common.hpp): round 4 replaced the v_rcp projection everywhere

__device__ __forceinline__ LanePose lane_make_pose(const float* __restrict__ P, float xf, float yf, int h, int w) {
    return pmn_make_pose(P, xf, yf, h, w);
}

// Tap record of one item.  The general path is gather_corr.hip's (behind-camera sentinel, module.py:166-169; border handling
// of pmn_axis).  When EVERY lane of the wave is an interior item (both low corners in [0, size-2]: the 4 taps exist at their
// natural slots) the selects of pmn_axis are skipped -- same floor / weight expressions, same values.
__device__ __forceinline__ bool lane_project(const LanePose& q, float dep, bool active, int h, int w, int hs, int ws, PmnTapsXY& t) {
    float ix, iy;
    const bool front = pmn_pose_position(q, dep, h, w, hs, ws, ix, iy) && active;
    const bool interior = front && ix >= 0.0f && ix < (float)(ws - 1) && iy >= 0.0f && iy < (float)(hs - 1);
    if (__builtin_amdgcn_ballot_w64(!interior) == 0ull) {
#pragma clang fp contract(off)
        const float fx = floorf(ix), fy = floorf(iy);
        const float ax = (fx + 1.0f) - ix, bx = ix - fx, ay = (fy + 1.0f) - iy, by = iy - fy;
        t.x0 = (int)fx;
        t.y0 = (int)fy;
        t.w00 = ax * ay;
        t.w01 = bx * ay;
        t.w10 = ax * by;
        t.w11 = bx * by;
        return true;
    }
    t.x0 = 0;
    t.y0 = 0;
    t.w00 = t.w01 = t.w10 = t.w11 = 0.0f;
    if (front) t = pmn_make_taps_xy(ix, iy, hs, ws);
    return front;
}

// window = bounding box of [lo, hi] corner ranges (+1 for the south-east corners); cut down around (cx, cy) when it does not fit
__device__ __forceinline__ LaneWin lane_make_window(int lo_x, int hi_x, int lo_y, int hi_y, int cxl, int cyl, int cap_texels) {
    const int sx0 = lane_wave_minmax<false>(lo_x), sx1 = lane_wave_minmax<true>(hi_x);
    const int sy0 = lane_wave_minmax<false>(lo_y), sy1 = lane_wave_minmax<true>(hi_y);
    LaneWin g;
    if (sx0 > sx1) {  // no live item in this wave
        g.bx0 = 0; g.by0 = 0; g.bw = 2; g.bh = 2;
        return g;
    }
    g.bx0 = sx0;
    g.by0 = sy0;
    g.bw = sx1 - sx0 + 2;
    g.bh = sy1 - sy0 + 2;
    if (g.bw * g.bh > cap_texels) {
        const int cx = __builtin_amdgcn_readlane(cxl, 52), cy = __builtin_amdgcn_readlane(cyl, 52);  // lane 52 = row 2, column 8
        const int bh = min(g.bh, 8);
        const int bw = max(min(g.bw, (int)((float)cap_texels / (float)bh)), 2);
        const int ccx = cx >= 0 ? cx : ((sx0 + sx1) >> 1), ccy = cy >= 0 ? cy : ((sy0 + sy1) >> 1);
        g.bx0 = min(max(ccx - (bw >> 1) + 1, sx0), sx1 + 2 - bw);
        g.by0 = min(max(ccy - (bh >> 1) + 1, sy0), sy1 + 2 - bh);
        g.bw = bw;
        g.bh = bh;
    }
    return g;
}

// Stage the window of one source map's 16-channel slice into the wave's LDS region (texel-major, 64 B per texel) by LDS-DMA
// (global_load_lds_dwordx4): a wave-instruction moves 16 texels of one window row (lane = texel * 4 + channel quad) straight
// from the L1 return path into LDS at base + lane * 16 -- no staging registers, no ds_write, no VALU beyond the source address
// (the register-staged form of gather_win.hip measured 211 / 250 / 231 us on the s3-it2 / s2 / s1 launches, this one 186 / 224 /
// 221; streaming: 170 / 172 / 183).  Lanes past the end of a row are masked off (their destination would be the next row's first texels).
// Nothing waits here: the caller waits for vmcnt(0) before the first ds_read of the window.
template <int C>
__device__ __forceinline__ void lane_stage_window_dma(pmn_lds_char* win, pmn_glb_char* src_slice, const LaneWin& g, int ws, int lane) {
    const int tcol = lane >> 2, quad = lane & 3;
    for (int r = 0; r < g.bh; ++r) {
        const unsigned grow = (unsigned)((g.by0 + r) * ws + g.bx0);
        for (int c0 = 0; c0 < g.bw; c0 += 16) {
            if (c0 + tcol < g.bw) {
                const unsigned go = (grow + (unsigned)(c0 + tcol)) * (unsigned)(C * 4) + quad * 16u;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_slice + go),
                                                 (__attribute__((address_space(3))) void*)(win + (((unsigned)(r * g.bw + c0)) << 6)), 16, 0, 0);
            }
        }
    }
}

// The 16 corner quads of one item's slice (4 channel quads in this lane's rotated order x 4 corners) and the four blended
// dots.  HALF: two batches of 8 loads (32 instead of 64 registers in flight) -- the 3-waves-per-SIMD build.
template <int C, bool HALF>
__device__ __forceinline__ void lane_gather16(pmn_lds_char* win, pmn_glb_char* src_slice, const LaneWin& g, int x0, int y0,
                                              const pmn_f4 w4, bool tv, int ws, const unsigned (&ofs)[4], const pmn_f4 (&refq)[4],
                                              float (&dot)[4]) {
    const int lx = tv ? x0 - g.bx0 : 0, ly = tv ? y0 - g.by0 : 0;
    const bool inside = (unsigned)lx < (unsigned)(g.bw - 1) && (unsigned)ly < (unsigned)(g.bh - 1);
    const unsigned aN = ((unsigned)(ly * g.bw + lx)) << 6, aS = aN + ((unsigned)g.bw << 6);
    const unsigned go = (unsigned)(y0 * ws + x0) * (unsigned)(C * 4), rb = (unsigned)ws * (unsigned)(C * 4);
    constexpr int NJ = HALF ? 2 : 4;
#pragma unroll
    for (int j0 = 0; j0 < 4; j0 += NJ) {
        pmn_f4 c[NJ][4];
        if (inside) {
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int j = j0 + jj;
                c[jj][0] = PMN_LDS_F4(win + (aN + ofs[j]));
                c[jj][1] = PMN_LDS_F4(win + (aN + ofs[j] + 64));
                c[jj][2] = PMN_LDS_F4(win + (aS + ofs[j]));
                c[jj][3] = PMN_LDS_F4(win + (aS + ofs[j] + 64));
            }
        } else {
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int j = j0 + jj;
                c[jj][0] = PMN_GLB_F4(src_slice + (go + ofs[j]));
                c[jj][1] = PMN_GLB_F4(src_slice + (go + ofs[j]) + C * 4);
                c[jj][2] = PMN_GLB_F4(src_slice + ((go + rb) + ofs[j]));
                c[jj][3] = PMN_GLB_F4(src_slice + ((go + rb) + ofs[j]) + C * 4);
            }
        }
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) dot[j0 + jj] = lane_blend_dot(c[jj][0], c[jj][1], c[jj][2], c[jj][3], w4, refq[j0 + jj]);
        if (HALF) __builtin_amdgcn_sched_barrier(0);
    }
}

// Pointwise MLP G -> 16 -> 8 -> 1 for NP PAIRS of items, weights broadcast from LDS (packed block of params.pack_mlp, see
// gather_common.hpp mlp_from_lds for the record layout).  Each component goes through exactly the scalar version's operations
// in the same order (v_pk_fma_f32 = two IEEE fmas), so the results are bit-identical to it.
template <int G, int NP>
__device__ __forceinline__ void lane_mlp_pairs(pmn_lds_char* W, const pmn_f2 (&x)[NP][G], pmn_f2 (&out)[NP]) {
    pmn_f2 a1[NP][8];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int k = 0; k < 8; ++k) a1[p][k] = pmn_f2{0.0f, 0.0f};
#pragma unroll 1
    for (int j = 0; j < 16; ++j) {
        pmn_lds_char* rp = W + 80 * j;
        const pmn_f4 r0 = PMN_LDS_F4(rp), c0 = PMN_LDS_F4(rp + 32), c1 = PMN_LDS_F4(rp + 48);
        float w0[8] = {r0.x, r0.y, r0.z, r0.w, 0.f, 0.f, 0.f, 0.f};
        if (G == 8) {
            const pmn_f4 r1 = PMN_LDS_F4(rp + 16);
            w0[4] = r1.x; w0[5] = r1.y; w0[6] = r1.z; w0[7] = r1.w;
        }
        const float w1c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const float t0 = PMN_LDS_F(rp + 64);
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
    const pmn_f4 ta = PMN_LDS_F4(W + 1280), tb = PMN_LDS_F4(W + 1296), wa = PMN_LDS_F4(W + 1312), wb = PMN_LDS_F4(W + 1328);
    const float t1[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
    const float w2[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
    const float b2 = PMN_LDS_F(W + 1344);
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

// ============================================================================================================================
// One wave = one unit: a 16x4 pixel tile x DCH consecutive hypotheses, all views, all channels.
// ============================================================================================================================
template <int C, int G, int DCH, int MODE, int WPS>
__global__ __launch_bounds__(PMN_BLOCK, WPS) void gather_lane_kernel(const GatherArgs a, const int cap_bytes, const int nunits,
                                                                   unsigned long long* __restrict__ keys, const int dbg) {
    constexpr int NS = C / 16;    // channel slices walked by every lane
    constexpr int CG = C / G;     // channels per correlation group (4 or 8)
    constexpr int GPS = 16 / CG;  // groups per slice (4 or 2)
    static_assert(CG == 4 || CG == 8, "group size must be 4 or 8 channels");
    static_assert(DCH % 2 == 0, "hypotheses are processed in pairs by the MLP");
    constexpr int MLP_BYTES = ((PMN_MLP_FLOATS * 4 + 15) / 16) * 16;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wg = pmn_xcd_tile(blockIdx.x, gridDim.x);
    const int unit = wg * 4 + wave;
    if (unit >= nunits) return;  // no barrier in this kernel: a wave may leave alone
    int irow, icol;
    lane_tile_pos(lane, irow, icol);
    const int rot = (icol >> 2) & 3;

    const int N = a.N, h = a.h, w = a.w, hs = a.hs, ws = a.ws, D = a.D;
    const int hw = h * w;
    const int b = blockIdx.z;
    const int nchunk = (D + DCH - 1) / DCH;
    const int tile = unit / nchunk, chunk = unit - tile * nchunk;
    const int d0 = chunk * DCH;
    const int nd = min(DCH, D - d0);  // run-time on purpose (a compile-time count lets hipcc merge the steps and spill)
    const int ntx = (w + 15) >> 4;
    const int ty = tile / ntx, tx = tile - ty * ntx;
    const int x = tx * 16 + icol, y = ty * 4 + irow;
    const bool ok = x < w && y < h;
    const int p = ok ? y * w + x : 0;

    pmn_lds_char* win = (pmn_lds_char*)(smem + wave * (cap_bytes + MLP_BYTES));
    pmn_lds_char* wlds = win + cap_bytes;
    const int cap_texels = cap_bytes >> 6;
    {  // the wave's own copy of the MLP block (no workgroup barrier needed before its use)
        const float* mlp = MODE == MODE_VW ? a.mlp_b : a.mlp_a;
        for (int i = lane; i < PMN_MLP_FLOATS; i += 64) PMN_LDS_F(wlds + 4 * i) = mlp[i];
    }

    unsigned ofs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ofs[j] = (unsigned)((j + rot) & 3) * 16u;
    pmn_glb_char* refp = (pmn_glb_char*)a.ref + ((size_t)b * hw + p) * (C * 4);

    float rdep[DCH];
#pragma unroll
    for (int s = 0; s < DCH; ++s) rdep[s] = (ok && s < nd) ? a.depth[((size_t)b * D + d0 + s) * hw + p] : 0.0f;

    // sims[s][sl * GPS + k]: slice sl, k-th group in this lane's rotated order.  MODE_AGG: running sum over views of
    // similarity * view weight (the hypothesis pair packed for the accumulate); MODE_VW: this view's similarity.
    pmn_f2 sims[DCH / 2][G];
#pragma unroll
    for (int s = 0; s < DCH / 2; ++s)
#pragma unroll
        for (int k = 0; k < G; ++k) sims[s][k] = pmn_f2{0.0f, 0.0f};
    float wtot = 1e-5f;

    const float xf = (float)x, yf = (float)y;
    const int wv = w >> a.vw_shift, hwv = (h >> a.vw_shift) * wv;
    const int vw_idx = (y >> a.vw_shift) * wv + (x >> a.vw_shift);

    float vw_next = 0.0f;
    if (MODE == MODE_AGG && ok) vw_next = a.vw_in[((size_t)b * N) * hwv + vw_idx];

    for (int v = 0; v < N; ++v) {
        const LanePose q = lane_make_pose(a.proj + ((size_t)b * N + v) * 16, xf, yf, h, w);
        const float vw = vw_next;
        if (MODE == MODE_AGG && ok && v + 1 < N) vw_next = a.vw_in[((size_t)b * N + v + 1) * hwv + vw_idx];
        if (MODE == MODE_AGG) wtot += vw;

        // Tap records of the lane's DCH items.  NS > 1: all of them once per view (kept in registers, shared by the channel
        // slices, exact bounding box).  NS == 1: first and last only (they bound the sorted chunk: positions move monotonically
        // along the epipolar line), the others are projected right before their use -- 8 live records would not fit.
        constexpr bool STORE = NS > 1;
        int rx0[DCH], ry0[DCH];
        pmn_f4 rw[DCH];
        unsigned tvmask = 0;
        int lo_x = 1 << 20, hi_x = -(1 << 20), lo_y = 1 << 20, hi_y = -(1 << 20);
#pragma unroll
        for (int s = 0; s < DCH; ++s) {
            rx0[s] = 0;
            ry0[s] = 0;
            rw[s] = pmn_f4{0.f, 0.f, 0.f, 0.f};
            if (STORE || s == 0) {
                PmnTapsXY t;
                t.x0 = 0; t.y0 = 0; t.w00 = t.w01 = t.w10 = t.w11 = 0.0f;
                bool tv = false;
                if (s < nd) tv = lane_project(q, rdep[s], ok, h, w, hs, ws, t);
                rx0[s] = t.x0;
                ry0[s] = t.y0;
                rw[s] = pmn_f4{t.w00, t.w01, t.w10, t.w11};
                if (tv) {
                    tvmask |= 1u << s;
                    lo_x = min(lo_x, t.x0); hi_x = max(hi_x, t.x0);
                    lo_y = min(lo_y, t.y0); hi_y = max(hi_y, t.y0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!STORE) {
            float dlast = rdep[0];
#pragma unroll
            for (int s = 1; s < DCH; ++s)
                if (s < nd) dlast = rdep[s];
            PmnTapsXY t;
            if (lane_project(q, dlast, ok, h, w, hs, ws, t)) {
                lo_x = min(lo_x, t.x0); hi_x = max(hi_x, t.x0);
                lo_y = min(lo_y, t.y0); hi_y = max(hi_y, t.y0);
            }
        }
        const bool anyv = lo_x <= hi_x;
        const LaneWin g = lane_make_window(lo_x, hi_x, lo_y, hi_y, anyv ? (lo_x + hi_x) >> 1 : -1, anyv ? (lo_y + hi_y) >> 1 : -1,
                                           cap_texels);
        const pmn_f2 vw2 = {vw, vw};

#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            pmn_glb_char* src_slice = (pmn_glb_char*)a.src + (((size_t)(v * a.B + b) * hs * ws) * (C * 4) + sl * 64);
            pmn_f4 refq[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                refq[j] = pmn_f4{0.f, 0.f, 0.f, 0.f};
                if (ok) refq[j] = PMN_GLB_F4(refp + (sl * 64 + ofs[j]));
            }
            if (!(dbg & 1)) {
                lane_stage_window_dma<C>(win, src_slice, g, ws, lane);
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the window has landed
            }
#pragma unroll
            for (int s2 = 0; s2 < DCH / 2; ++s2) {
                float simp[2][GPS];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int s = 2 * s2 + e;
#pragma unroll
                    for (int k = 0; k < GPS; ++k) simp[e][k] = 0.0f;
                    if (s < nd) {
                        if (!STORE && s > 0) {
                            PmnTapsXY t;
                            if (lane_project(q, rdep[s], ok, h, w, hs, ws, t)) tvmask |= 1u << s;
                            rx0[s] = t.x0;
                            ry0[s] = t.y0;
                            rw[s] = pmn_f4{t.w00, t.w01, t.w10, t.w11};
                        }
                        float dot[4];
                        if (dbg & 2) {
                            dot[0] = rw[s].x; dot[1] = rw[s].y; dot[2] = rw[s].z; dot[3] = rw[s].w;
                        } else {
                            lane_gather16<C, WPS >= 3>(win, src_slice, g, rx0[s], ry0[s], rw[s], (tvmask >> s) & 1, ws, ofs, refq, dot);
                        }
                        if constexpr (CG == 4) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) simp[e][j] = dot[j] * 0.25f;
                        } else {  // quads {0,1} / {2,3} pair up; the two-term sums are commutative (= the streaming kernel's pair swap)
                            const bool odd = rot & 1;
                            simp[e][0] = (dot[0] + (odd ? dot[3] : dot[1])) * 0.125f;
                            simp[e][1] = (dot[2] + (odd ? dot[1] : dot[3])) * 0.125f;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int k = 0; k < GPS; ++k) {
                    const pmn_f2 sp = {simp[0][k], simp[1][k]};
                    if (MODE == MODE_AGG) {
#pragma clang fp contract(off)
                        sims[s2][sl * GPS + k] = sims[s2][sl * GPS + k] + sp * vw2;  // two roundings, as the reference's mul and add
                    } else {
                        sims[s2][sl * GPS + k] = sp;
                    }
                }
            }
        }

        if (MODE == MODE_VW) {
            // PixelwiseNet + sigmoid on this view's DCH items, max over them (first arg-max on ties through ~d), one atomic
            pmn_f2 xin[DCH / 2][G], r[DCH / 2];
#pragma unroll
            for (int s2 = 0; s2 < DCH / 2; ++s2)
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {
                    if constexpr (CG == 4) {
#pragma unroll
                        for (int qd = 0; qd < 4; ++qd) {  // group qd of the slice sits at rotated position (qd - rot) & 3
                            const pmn_f2 c0 = sims[s2][sl * 4 + 0], c1 = sims[s2][sl * 4 + 1], c2 = sims[s2][sl * 4 + 2],
                                         c3 = sims[s2][sl * 4 + 3];
                            const int pos = (qd - rot) & 3;
                            xin[s2][sl * 4 + qd] = pos == 0 ? c0 : (pos == 1 ? c1 : (pos == 2 ? c2 : c3));
                        }
                    } else {  // (component-wise selects on VALUES: `sw ? a[1] : a[0]` would become a scratch array indexed by sw)
                        const bool sw = (rot >> 1) & 1;
                        const pmn_f2 e0 = sims[s2][sl * 2 + 0], e1 = sims[s2][sl * 2 + 1];
                        xin[s2][sl * 2 + 0] = pmn_f2{sw ? e1.x : e0.x, sw ? e1.y : e0.y};
                        xin[s2][sl * 2 + 1] = pmn_f2{sw ? e0.x : e1.x, sw ? e0.y : e1.y};
                    }
                }
            lane_mlp_pairs<G, DCH / 2>(wlds, xin, r);
            unsigned long long best = 0ull;
#pragma unroll
            for (int s = 0; s < DCH; ++s) {
                if (s < nd) {
                    const float rv = (s & 1) ? r[s / 2].y : r[s / 2].x;
                    const unsigned long long key = ((unsigned long long)__float_as_uint(pmn_sigmoid(rv)) << 32) |
                                                   (unsigned long long)(0xFFFFFFFFu - (unsigned)(d0 + s));
                    best = key > best ? key : best;
                }
            }
            if (ok) atomicMax(&keys[((size_t)b * N + v) * hw + p], best);
        }
    }

    if (MODE == MODE_VW || !ok) return;
    // ---- MODE_AGG epilogue: / weight sum, SimilarityNet MLP, cost (hypothesis-last) --------------------------------------------
    pmn_f2 xin[DCH / 2][G], o[DCH / 2];
    const pmn_f2 wt2 = {wtot, wtot};
#pragma unroll
    for (int s2 = 0; s2 < DCH / 2; ++s2)
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            if constexpr (CG == 4) {
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const pmn_f2 c0 = sims[s2][sl * 4 + 0], c1 = sims[s2][sl * 4 + 1], c2 = sims[s2][sl * 4 + 2],
                                 c3 = sims[s2][sl * 4 + 3];
                    const int pos = (qd - rot) & 3;
                    xin[s2][sl * 4 + qd] = (pos == 0 ? c0 : (pos == 1 ? c1 : (pos == 2 ? c2 : c3))) / wt2;
                }
            } else {
                const bool sw = (rot >> 1) & 1;
                const pmn_f2 e0 = sims[s2][sl * 2 + 0], e1 = sims[s2][sl * 2 + 1];
                xin[s2][sl * 2 + 0] = pmn_f2{sw ? e1.x : e0.x, sw ? e1.y : e0.y} / wt2;
                xin[s2][sl * 2 + 1] = pmn_f2{sw ? e0.x : e1.x, sw ? e0.y : e1.y} / wt2;
            }
        }
    if (dbg & 8) {
#pragma unroll
        for (int s2 = 0; s2 < DCH / 2; ++s2) o[s2] = xin[s2][0];
    } else {
        constexpr int NPB = DCH / 2 > 2 ? 2 : DCH / 2;  // pairs per MLP pass (bounds the live layer-2 accumulators)
#pragma unroll
        for (int c0 = 0; c0 < DCH / 2; c0 += NPB) {
            pmn_f2 xi[NPB][G], oo[NPB];
#pragma unroll
            for (int i = 0; i < NPB; ++i)
#pragma unroll
                for (int gI = 0; gI < G; ++gI) xi[i][gI] = xin[c0 + i][gI];
            lane_mlp_pairs<G, NPB>(wlds, xi, oo);
#pragma unroll
            for (int i = 0; i < NPB; ++i) o[c0 + i] = oo[i];
        }
    }
    float* outp = a.out + ((size_t)b * hw + p) * D + d0;  // cost is hypothesis-last [B,h,w,D]
#pragma unroll
    for (int s = 0; s < DCH; ++s) {
        if (s < nd) {
            outp[s] = (s & 1) ? o[s / 2].y : o[s / 2].x;
            if (a.sim_out) {
#pragma unroll
                for (int gI = 0; gI < G; ++gI)
                    a.sim_out[(((size_t)b * G + gI) * D + d0 + s) * hw + p] = (s & 1) ? xin[s / 2][gI].y : xin[s / 2][gI].x;
            }
        }
    }
}

// keys -> view weights (+ arg-max index): one thread per (batch, view, pixel)
__global__ void unpack_keys_kernel(const unsigned long long* __restrict__ keys, float* __restrict__ vw, int* __restrict__ argmax,
                                   size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = keys[i];
    vw[i] = __uint_as_float((unsigned)(key >> 32));
    if (argmax) argmax[i] = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
}

// ---- host side ---------------------------------------------------------------------------------------------------------------

static int g_lane_cap_bytes = 12 * 1024;  // LDS window per wave
static int g_lane_dbg = 0;
static int g_lane_wps = 3;  // waves per SIMD the kernel is built for (3: 168 registers, half batches; 2: 256 registers)

int pmn_lane_set_tuning(int key, int value) {
    if (key == 4) {
        if (value < 1024 || value > 36 * 1024 || (value & 1023)) return PMN_ERR_ARG;
        g_lane_cap_bytes = value;
        return PMN_OK;
    }
    if (key == 5) {
        g_lane_dbg = value;
        return PMN_OK;
    }
    if (key == 6) {
        if (value != 2 && value != 3) return PMN_ERR_ARG;
        g_lane_wps = value;
        return PMN_OK;
    }
    return PMN_ERR_ARG;
}

template <int C, int G, int DCH, int MODE, int WPS>
static int launch_lane_wps(GatherArgs& a, unsigned long long* keys, hipStream_t stream) {
    const int ntx = (a.w + 15) / 16, nty = (a.h + 3) / 4;
    const int nchunk = (a.D + DCH - 1) / DCH;
    const int nunits = ntx * nty * nchunk;
    const int nwg = (nunits + 3) / 4;
    const int cap = g_lane_cap_bytes;
    size_t lds = (size_t)4 * (cap + ((PMN_MLP_FLOATS * 4 + 15) / 16) * 16);
    auto kern = gather_lane_kernel<C, G, DCH, MODE, WPS>;
    static size_t lds_set = 0;  // per instantiation: largest dynamic-LDS size the attribute has been raised to
    if (lds > 48 * 1024 && lds > lds_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return PMN_ERR_LAUNCH;
        lds_set = lds;
    }
    PMN_LAUNCH(kern, dim3(nwg, 1, a.B), dim3(PMN_BLOCK), lds, stream, a, cap, nunits, keys, g_lane_dbg);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

template <int C, int G, int DCH, int MODE>
static int launch_lane(GatherArgs& a, unsigned long long* keys, hipStream_t stream) {
    return g_lane_wps == 2 ? launch_lane_wps<C, G, DCH, MODE, 2>(a, keys, stream) : launch_lane_wps<C, G, DCH, MODE, 3>(a, keys, stream);
}

template <int C, int G, int DCH>
static int run_lane(GatherArgs& a, bool pixelwise, hipStream_t stream) {
    if (!pixelwise) return launch_lane<C, G, DCH, MODE_AGG>(a, nullptr, stream);
    // PixelwiseNet launch: pass 1 (keys) -> unpack -> pass 2 (aggregation with the weights just computed).  The keys
    // ([B,N,h,w] 64-bit) borrow the cost buffer ([B,h,w,D] floats), which pass 2 overwrites afterwards.
    const size_t nkeys = (size_t)a.B * a.N * a.h * a.w;
    if (nkeys * 8 > (size_t)a.B * a.h * a.w * a.D * 4) return PMN_ERR_SHAPE;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(a.out);
    if (hipMemsetAsync(keys, 0, nkeys * 8, stream) != hipSuccess) return PMN_ERR_LAUNCH;
    int rc = launch_lane<C, G, DCH, MODE_VW>(a, keys, stream);
    if (rc != PMN_OK) return rc;
    PMN_LAUNCH(unpack_keys_kernel, dim3((unsigned)((nkeys + 255) / 256)), dim3(256), 0, stream, keys, a.vw_out,
                       a.vw_argmax, nkeys);
    PMN_CHECK_LAUNCH();
    GatherArgs a2 = a;
    a2.vw_in = a.vw_out;
    a2.vw_shift = 0;
    return launch_lane<C, G, DCH, MODE_AGG>(a2, nullptr, stream);
}

int pmn_launch_gather_lane(GatherArgs& a, int C, int G, bool pixelwise, hipStream_t stream) {
    if (C == 64 && G == 8) return run_lane<64, 8, 4>(a, pixelwise, stream);
    if (C == 32 && G == 8) return run_lane<32, 8, 4>(a, pixelwise, stream);
    if (C == 16 && G == 4) return run_lane<16, 4, 8>(a, pixelwise, stream);
    return PMN_ERR_SHAPE;
}
