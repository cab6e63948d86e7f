// gather_tile.hip -- pmn_warp_correlate with known view weights, tile-window form (opt-in: pmn_set_tuning key 1, bit 5).
//
// What the two earlier LDS-window forms taught (DESIGN.md lessons 16-19): the streaming kernel runs the vector-memory pipe at the
// rate a pure load loop reaches, so only taps served from LDS (128 B/clk/CU instead of 64) can be cheaper -- but a window per
// (wave, 8 hypotheses) re-uses a staged texel only ~3 times, pays a projection per channel slice, a window set-up per 8 items
// of a lane and holds 64 corner registers per lane.  This form keeps the streaming kernel's lean walk and changes the source of
// the taps:
//   * a workgroup owns a 16x4 pixel tile x DT hypotheses (all of them at stage 1 / 2, half at stage 3) and, per source view,
//     ONE window: the bounding box of every tap of the tile.  512-1024 items x 4 taps land in 150-400 staged texels (re-use
//     10-25x, tests/studies/footprint_study.py), so staging is ~5 % of the tap bytes and goes through LDS-DMA
//     (global_load_lds_dwordx4: no registers, no ds_write), one 16-channel slice ahead of the walk (two window buffers);
//   * thread (pixel, d) projects its DT/4 hypotheses ONCE per view and parks {4 corner weights, window byte offset} in an LDS
//     record tile, shared by all channel slices;
//   * the walk is the streaming kernel's: 4 lanes = the channel quads of one 16-channel slice of an item, a wave = one tile row of
//     16 pixels, hypothesis by hypothesis: record (ds_read_b128 + b64), four ds_read_b128 corner quads from the window, packed
//     blend with op_sel weights, dot with the register-resident reference quad, group sum, accumulation over views in registers.
// Taps outside the window (bounding box larger than the window capacity) are loaded from global memory by the wave that meets
// them.  Arithmetic and operation order are gather_corr.hip's, so the results agree bit for bit (tests/test_gather_win.py).
// Reference: models/module.py:130-181, models/patchmatch.py:192-217, :570.
#include "../gather_common.hpp"

typedef float pmn_t4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char pmn_tlds;        // LDS (ds_read / ds_write, never flat)
typedef const __attribute__((address_space(1))) char pmn_tglb;  // global (global_load, never flat)
#define PMN_TLDS_F4(p) (*reinterpret_cast<__attribute__((address_space(3))) pmn_t4*>(p))
#define PMN_TLDS_F(p) (*reinterpret_cast<__attribute__((address_space(3))) float*>(p))
#define PMN_TLDS_I(p) (*reinterpret_cast<__attribute__((address_space(3))) int*>(p))
#define PMN_TGLB_F4(p) (*reinterpret_cast<const __attribute__((address_space(1))) pmn_t4*>(p))

template <bool MAX>
__device__ __forceinline__ int tile_wave_minmax(int v) {
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

// packed blend with the weight taken from one half of a register pair (op_sel), see scripts/experiments/README.md
__device__ __forceinline__ pmn_f2 tile_pk_mul_lo(const pmn_f2 t, const pmn_f2 w) {
    pmn_f2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(r) : "v"(t), "v"(w));
    return r;
}
__device__ __forceinline__ pmn_f2 tile_pk_fma_lo(const pmn_f2 t, const pmn_f2 w, pmn_f2 c) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(c) : "v"(t), "v"(w));
    return c;
}
__device__ __forceinline__ pmn_f2 tile_pk_fma_hi(const pmn_f2 t, const pmn_f2 w, pmn_f2 c) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(c) : "v"(t), "v"(w));
    return c;
}

// per channel ((t00*w00 + t01*w01) + t10*w10) + t11*w11, then the dot product with the reference quad: gather_corr.hip's order
__device__ __forceinline__ float tile_blend_dot(const pmn_t4 t00, const pmn_t4 t01, const pmn_t4 t10, const pmn_t4 t11,
                                                const pmn_t4 w4, const pmn_t4 refq) {
    const pmn_f2 wab = {w4.x, w4.y}, wcd = {w4.z, w4.w};
    pmn_f2 lo = tile_pk_mul_lo(pmn_f2{t00.x, t00.y}, wab);
    pmn_f2 hi = tile_pk_mul_lo(pmn_f2{t00.z, t00.w}, wab);
    lo = tile_pk_fma_hi(pmn_f2{t01.x, t01.y}, wab, lo);
    hi = tile_pk_fma_hi(pmn_f2{t01.z, t01.w}, wab, hi);
    lo = tile_pk_fma_lo(pmn_f2{t10.x, t10.y}, wcd, lo);
    hi = tile_pk_fma_lo(pmn_f2{t10.z, t10.w}, wcd, hi);
    lo = tile_pk_fma_hi(pmn_f2{t11.x, t11.y}, wcd, lo);
    hi = tile_pk_fma_hi(pmn_f2{t11.z, t11.w}, wcd, hi);
    return fmaf(hi.y, refq.w, fmaf(hi.x, refq.z, fmaf(lo.y, refq.y, lo.x * refq.x)));
}

// Window taps are read with explicit ds_read_b128 statements: hipcc cannot prove that the LDS-DMA of the NEXT slice's window
// (in flight during the walk) does not alias the window being read, and would put s_waitcnt vmcnt(0) in front of every
// compiler-generated read of it -- the walk would wait for the staging it is supposed to hide.  tile_lds_wait() is the matching
// s_waitcnt lgkmcnt(0); it takes the loaded registers as in/out operands so that no consumer can be scheduled above it.
template <int OFS>
__device__ __forceinline__ pmn_t4 tile_lds_read(const unsigned addr) {
    pmn_t4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFS));
    return r;
}
// one item's LDS reads of the walk: corner weights + the four corner quads (5 x ds_read_b128, issued together)
struct TileItem {
    pmn_t4 w, c00, c01, c10, c11;
};
__device__ __forceinline__ void tile_issue(TileItem& it, const unsigned rec_addr, const unsigned aN, const unsigned aS) {
    it.w = tile_lds_read<0>(rec_addr);
    it.c00 = tile_lds_read<0>(aN);
    it.c01 = tile_lds_read<64>(aN);
    it.c10 = tile_lds_read<0>(aS);
    it.c11 = tile_lds_read<64>(aS);
}
// LDS returns in order: waiting until at most PENDING reads are outstanding completes everything issued before them.  The two
// items ride as in/out operands so that nothing that consumes them can be scheduled above the wait.
template <int PENDING>
__device__ __forceinline__ void tile_wait(TileItem& a, TileItem& b) {
    asm volatile("s_waitcnt lgkmcnt(%10)"
                 : "+v"(a.w), "+v"(a.c00), "+v"(a.c01), "+v"(a.c10), "+v"(a.c11), "+v"(b.w), "+v"(b.c00), "+v"(b.c01), "+v"(b.c10),
                   "+v"(b.c11)
                 : "n"(PENDING));
}

// grid = (pixel tiles of 16x4, D / DT hypothesis chunks, batch); 256 threads = 4 waves = the 4 rows of the tile.
// (the hypothesis count of a chunk, nd, stays a run-time value on purpose: with a compile-time count hipcc merges the walk's steps
// and spills)
template <int C, int G, int DT>
__global__ __launch_bounds__(PMN_BLOCK, 2) void gather_tile_kernel(const GatherArgs a, const int cap_bytes) {
    constexpr int NS = C / 16;     // channel slices
    constexpr int CG = C / G;      // channels per correlation group (4 or 8)
    constexpr int LPG = CG / 4;    // lanes (quads) per group
    constexpr int GPS = 16 / CG;   // groups per slice
    constexpr int TP = 64;         // pixels per tile
    constexpr int NR = DT / 4;     // records projected per thread and view
    constexpr int NIT = DT / 4;    // epilogue items per thread
    constexpr int NI = NIT < 4 ? NIT : 4;
    constexpr int REC_BYTES = DT * TP * (2 * 16 + 4);  // two buffers of corner weights (float4) + one of window offsets (int)
    static_assert(CG == 4 || CG == 8, "group size must be 4 or 8 channels");
    static_assert(DT % 4 == 0, "hypotheses split over the four pixel-copies of the projection role");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    pmn_tlds* smem = (pmn_tlds*)smem_raw;
    pmn_tlds* recw = smem;                          // [2][DT][TP] float4: views alternate between the two buffers
    pmn_tlds* reco = smem + 2 * DT * TP * 16;           // [TP][DT] int: window byte offset, or -(texel + 1) for a tap block outside it
    pmn_tlds* red = smem + REC_BYTES;               // [4 waves][4] ints
    pmn_tlds* wlds = red + 64;                      // MLP block
    pmn_tlds* win0 = wlds + ((PMN_MLP_FLOATS * 4 + 15) / 16) * 16;  // two window buffers of cap_bytes; later the similarity tile

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int N = a.N, h = a.h, w = a.w, hs = a.hs, ws = a.ws, D = a.D;
    const int hw = h * w;
    const int b = blockIdx.z;
    const int d_base = blockIdx.y * DT;
    const int nd = min(DT, D - d_base);
    const int ntx = (w + 15) >> 4;
    const int tile = pmn_xcd_tile(blockIdx.x, a.ntiles);
    const int ty = tile / ntx, tx = tile - ty * ntx;

    for (int i = tid; i < PMN_MLP_FLOATS; i += PMN_BLOCK) PMN_TLDS_F(wlds + 4 * i) = a.mlp_a[i];

    // ---- projection role: pixel pr_pix of the tile, hypotheses pr_d0 + 4 j ----------------------------------------------------
    const int pr_pix = tid & 63, pr_d0 = tid >> 6;
    const int pr_x = tx * 16 + (pr_pix & 15), pr_y = ty * 4 + (pr_pix >> 4);
    const bool pr_ok = pr_x < w && pr_y < h;
    const int pr_p = pr_ok ? pr_y * w + pr_x : 0;
    float rdep[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int d = pr_d0 + 4 * j;
        rdep[j] = (pr_ok && d < nd) ? a.depth[((size_t)b * D + d_base + d) * hw + pr_p] : -1.0f;
    }

    // ---- walk role: wave = tile row, 4 lanes per pixel = the channel quads of a 16-channel slice --------------------------------
    const int wk_col = lane >> 2, quad = lane & 3;
    const int wk_x = tx * 16 + wk_col, wk_y = ty * 4 + wave;
    const bool wk_ok = wk_x < w && wk_y < h;
    const int wk_p = wk_ok ? wk_y * w + wk_x : 0;
    const int wk_pix = wave * 16 + wk_col;
    pmn_t4 refq[NS];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
        refq[sl] = pmn_t4{0.f, 0.f, 0.f, 0.f};
        if (wk_ok) refq[sl] = PMN_TGLB_F4((pmn_tglb*)a.ref + (((size_t)b * hw + wk_p) * (C * 4) + sl * 64 + quad * 16));
    }
    float acc[NS][DT];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl)
#pragma unroll
        for (int d = 0; d < DT; ++d) acc[sl][d] = 0.0f;

    const float xf = (float)pr_x, yf = (float)pr_y;
    const int wv = w >> a.vw_shift, hwv = (h >> a.vw_shift) * wv;
    const int wk_vw_idx = (wk_y >> a.vw_shift) * wv + (wk_x >> a.vw_shift);
    const int cap_texels = cap_bytes >> 6;
    const int BIG = 1 << 20;

    struct Geom { int bx0, by0, bw, bh; };

    // Tap records of the thread's hypotheses for view v, the tile's window (bounding box of every tap, cut down when it does not
    // fit a window buffer) and the records parked in record buffer `rb`.  Contains ONE workgroup barrier.
    auto prepare_view = [&](const int v, const int rb) -> Geom {
        const PmnPose pose = pmn_make_pose(a.proj + ((size_t)b * N + v) * 16, xf, yf, h, w);  // the reference's own warp chain (round 4)
        PmnTapsXY rec[NR];
        bool rv[NR];
        int lo_x = BIG, hi_x = -BIG, lo_y = BIG, hi_y = -BIG;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            PmnTapsXY t;
            t.x0 = 0; t.y0 = 0;
            t.w00 = t.w01 = t.w10 = t.w11 = 0.0f;
            bool valid = false;
            if (pr_ok && pr_d0 + 4 * j < nd) {
                float ix, iy;
                if (pmn_pose_position(pose, rdep[j], h, w, hs, ws, ix, iy)) {  // false: behind the camera, samples nothing
                    t = pmn_make_taps_xy(ix, iy, hs, ws);
                    valid = true;
                }
            }
            rec[j] = t;
            rv[j] = valid;
            if (valid) {
                lo_x = min(lo_x, t.x0); hi_x = max(hi_x, t.x0);
                lo_y = min(lo_y, t.y0); hi_y = max(hi_y, t.y0);
            }
        }
        {
            const int m0 = tile_wave_minmax<false>(lo_x), m1 = tile_wave_minmax<true>(hi_x);
            const int m2 = tile_wave_minmax<false>(lo_y), m3 = tile_wave_minmax<true>(hi_y);
            if (lane == 0) {
                PMN_TLDS_I(red + wave * 16 + 0) = m0;
                PMN_TLDS_I(red + wave * 16 + 4) = m1;
                PMN_TLDS_I(red + wave * 16 + 8) = m2;
                PMN_TLDS_I(red + wave * 16 + 12) = m3;
            }
        }
        __syncthreads();  // every wave has also read its window offsets of the CURRENT view by now: `reco` may be rewritten
        Geom g;
        {
            int sx0 = BIG, sx1 = -BIG, sy0 = BIG, sy1 = -BIG;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sx0 = min(sx0, PMN_TLDS_I(red + q * 16 + 0));
                sx1 = max(sx1, PMN_TLDS_I(red + q * 16 + 4));
                sy0 = min(sy0, PMN_TLDS_I(red + q * 16 + 8));
                sy1 = max(sy1, PMN_TLDS_I(red + q * 16 + 12));
            }
            if (sx0 > sx1) { sx0 = sx1 = 0; sy0 = sy1 = 0; }  // no live item in this tile and view
            int bx0 = sx0, by0 = sy0, bw = sx1 - sx0 + 2, bh = sy1 - sy0 + 2;
            if (bw * bh > cap_texels) {  // cut down around the middle of the box; the stragglers take the global path
                const int nbh = max(min(bh, cap_texels / min(bw, 96)), 2);
                const int nbw = max(min(bw, cap_texels / nbh), 2);
                bx0 = sx0 + ((bw - nbw) >> 1);
                by0 = sy0 + ((bh - nbh) >> 1);
                bw = nbw; bh = nbh;
            }
            g.bx0 = __builtin_amdgcn_readfirstlane(bx0); g.by0 = __builtin_amdgcn_readfirstlane(by0);
            g.bw = __builtin_amdgcn_readfirstlane(bw); g.bh = __builtin_amdgcn_readfirstlane(bh);
        }
        pmn_tlds* rw = recw + rb * (DT * TP * 16);
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int d = pr_d0 + 4 * j;
            const int lx = rec[j].x0 - g.bx0, ly = rec[j].y0 - g.by0;
            int off = 0;  // dead items (zero weights) read the window's first texel: finite numbers times zero
            if (rv[j]) {
                const bool inside = (unsigned)lx < (unsigned)(g.bw - 1) && (unsigned)ly < (unsigned)(g.bh - 1);
                off = inside ? (ly * g.bw + lx) << 6 : -(rec[j].y0 * ws + rec[j].x0) - 1;
            }
            PMN_TLDS_F4(rw + ((d * TP + pr_pix) << 4)) = pmn_t4{rec[j].w00, rec[j].w01, rec[j].w10, rec[j].w11};
            PMN_TLDS_I(reco + ((pr_pix * DT + d) << 2)) = off;
        }
        return g;
    };

    // LDS-DMA staging of one 16-channel slice of a view's window: rows in 16-texel segments, dealt round-robin to the four waves
    auto stage = [&](pmn_tlds* win, const int v, const int sl, const Geom& g) {
        pmn_tglb* vb = (pmn_tglb*)a.src + ((size_t)(v * a.B + b) * hs * ws) * (C * 4);
        const int nseg = (g.bw + 15) >> 4;
        int s = 0;
        for (int r = 0; r < g.bh; ++r) {
            const unsigned grow = (unsigned)((g.by0 + r) * ws + g.bx0);
            for (int k = 0; k < nseg; ++k, ++s) {
                if ((s & 3) != wave) continue;
                const int c0 = k << 4;
                if (c0 + (lane >> 2) < g.bw) {
                    const unsigned go = (grow + (unsigned)(c0 + (lane >> 2))) * (unsigned)(C * 4) + sl * 64 + (lane & 3) * 16u;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vb + go),
                                                     (__attribute__((address_space(3))) void*)(win + (((unsigned)(r * g.bw + c0)) << 6)),
                                                     16, 0, 0);
                }
            }
        }
    };

    // ---- software pipeline over (view, slice) steps: while step s is walked, the window of step s+1 is in flight and -- at the
    // first slice of a view -- the NEXT view has been projected, boxed and its records parked in the other record buffer ------------
    Geom cur = prepare_view(0, 0);
    stage(win0, 0, 0, cur);
    for (int v = 0; v < N; ++v) {
        const float vw = wk_ok ? a.vw_in[((size_t)b * N + v) * hwv + wk_vw_idx] : 0.0f;
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's share of the (v, slice 0) window has landed
        __syncthreads();                     // records of view v and that window are visible; the previous view is finished
        const unsigned rowb = (unsigned)cur.bw << 6;
        int off[DT];  // this lane's pixel: window byte offsets of all its hypotheses (shared by the channel slices)
        bool any_stray = false;
#pragma unroll
        for (int d4 = 0; d4 < DT; d4 += 4) {
            const pmn_t4 o4 = PMN_TLDS_F4(reco + ((wk_pix * DT + d4) << 2));
            off[d4] = __float_as_int(o4.x); off[d4 + 1] = __float_as_int(o4.y);
            off[d4 + 2] = __float_as_int(o4.z); off[d4 + 3] = __float_as_int(o4.w);
        }
#pragma unroll
        for (int d = 0; d < DT; ++d) any_stray |= off[d] < 0 && d < nd;
        Geom nxt = cur;
        if (v + 1 < N) nxt = prepare_view(v + 1, (v + 1) & 1);
        pmn_tglb* vbase = (pmn_tglb*)a.src + ((size_t)(v * a.B + b) * hs * ws) * (C * 4);
        pmn_tlds* rcur = recw + (v & 1) * (DT * TP * 16);
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            const int step = v * NS + sl;
            pmn_tlds* win = win0 + (step & 1) * cap_bytes;
            pmn_tlds* wnext = win0 + ((step + 1) & 1) * cap_bytes;
            if (sl + 1 < NS) stage(wnext, v, sl + 1, cur);
            else if (v + 1 < N) stage(wnext, v + 1, 0, nxt);
            pmn_tglb* gsl = vbase + sl * 64 + quad * 16;
            const unsigned win_addr = (unsigned)(size_t)win + quad * 16u;  // this lane's LDS byte address of the window's first texel
            // Software pipeline, two items per step: the reads of step k+1 are issued before step k is blended (a wave has one
            // partner per SIMD at this kernel's LDS footprint, so it must cover its own LDS latency).
            const unsigned rec_addr = (unsigned)(size_t)rcur + (unsigned)(wk_pix << 4);
            TileItem buf[2][2];
            auto issue_step = [&](int d0, TileItem (&it)[2]) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned aN = win_addr + (unsigned)max(off[d0 + i], 0);  // strays: see the pass below
                    tile_issue(it[i], rec_addr + (unsigned)((d0 + i) * (TP * 16)), aN, aN + rowb);
                }
            };
            issue_step(0, buf[0]);
#pragma unroll
            for (int d0 = 0; d0 < DT; d0 += 2) {
                TileItem(&now)[2] = buf[(d0 >> 1) & 1];
                if (d0 + 2 < DT) {
                    issue_step(d0 + 2, buf[((d0 >> 1) + 1) & 1]);
                    tile_wait<10>(now[0], now[1]);
                } else {
                    tile_wait<0>(now[0], now[1]);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float s = tile_blend_dot(now[i].c00, now[i].c01, now[i].c10, now[i].c11, now[i].w, refq[sl]);
                    if (LPG == 2) s += pmn_pair_swap(s);
                    s = s * (1.0f / CG);
                    const float upd = mul_add_unfused(acc[sl][d0 + i], s, vw);
                    const bool take = off[d0 + i] >= 0 && d0 + i < nd;
                    acc[sl][d0 + i] = take ? upd : acc[sl][d0 + i];
                }
            }
            // Items whose taps lie outside the window (the bounding box did not fit): the wave that has some goes through its
            // hypotheses again and gathers those items from global memory.  Kept apart from the walk above on purpose: with the
            // global loads inside it hipcc waits for vmcnt(0) -- i.e. for the NEXT window's DMA -- before every step.
            if (__builtin_amdgcn_ballot_w64(any_stray) != 0ull) {
                const unsigned rb = (unsigned)ws * (unsigned)(C * 4);
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    const bool mine = off[d] < 0 && d < nd;
                    if (__builtin_amdgcn_ballot_w64(mine) != 0ull) {
                        const pmn_t4 wq = PMN_TLDS_F4(rcur + ((d * TP + wk_pix) << 4));
                        const unsigned go = (unsigned)(mine ? -off[d] - 1 : 0) * (unsigned)(C * 4);
                        const pmn_t4 g00 = PMN_TGLB_F4(gsl + go), g01 = PMN_TGLB_F4(gsl + go + C * 4);
                        const pmn_t4 g10 = PMN_TGLB_F4(gsl + (go + rb)), g11 = PMN_TGLB_F4(gsl + (go + rb) + C * 4);
                        float s = tile_blend_dot(g00, g01, g10, g11, wq, refq[sl]);
                        if (LPG == 2) s += pmn_pair_swap(s);
                        s = s * (1.0f / CG);
                        const float upd = mul_add_unfused(acc[sl][d], s, vw);
                        acc[sl][d] = mine ? upd : acc[sl][d];
                    }
                }
            }
            if (sl + 1 < NS) {
                __builtin_amdgcn_s_waitcnt(0x0F70);  // the next slice's window has landed
                __syncthreads();                     // ... for every wave, and nobody still reads the buffer it will replace
            }
        }
        cur = nxt;
    }

    // ---- hand-over to the pointwise MLP through the similarity tile [G][DT][TP] (it takes the window buffers' place) ------------
    __syncthreads();
    pmn_tlds* simt = win0;
    if ((quad % LPG) == 0) {
#pragma unroll
        for (int sl = 0; sl < NS; ++sl)
#pragma unroll
            for (int d = 0; d < DT; ++d)
                PMN_TLDS_F(simt + ((((sl * GPS + quad / LPG) * DT + d) * TP + wk_pix) << 2)) = acc[sl][d];
    }
    __syncthreads();
    if (!pr_ok) return;
    float wtot = 1e-5f;
    const int vwi = (pr_y >> a.vw_shift) * wv + (pr_x >> a.vw_shift);
    for (int v = 0; v < N; ++v) wtot += a.vw_in[((size_t)b * N + v) * hwv + vwi];
    // item pairs straight from the LDS tile (re-pairing a float [NIT][G] array afterwards goes through scratch)
    static_assert(NIT % 2 == 0, "the pointwise MLP runs on pairs of items");
    constexpr int NPC = NI / 2;  // pairs per MLP call
    pmn_f2 xq[NIT / 2][G];
    float o[NIT];
    const pmn_f2 wt2 = {wtot, wtot};
#pragma unroll
    for (int q = 0; q < NIT / 2; ++q) {
        const int da = min(pr_d0 + 4 * (2 * q), nd - 1), db = min(pr_d0 + 4 * (2 * q + 1), nd - 1);
#pragma unroll
        for (int g = 0; g < G; ++g)
            xq[q][g] = pmn_f2{PMN_TLDS_F(simt + (((g * DT + da) * TP + pr_pix) << 2)), PMN_TLDS_F(simt + (((g * DT + db) * TP + pr_pix) << 2))} / wt2;
    }
    const float* wmlp = reinterpret_cast<const float*>(smem_raw + (wlds - smem));
#pragma unroll
    for (int c = 0; c < NIT / NI; ++c) {
        pmn_f2 xi[NPC][G], oq[NPC];
#pragma unroll
        for (int i = 0; i < NPC; ++i)
#pragma unroll
            for (int g = 0; g < G; ++g) xi[i][g] = xq[c * NPC + i][g];
        mlp_pairs_from_lds<G, NPC>(wmlp, xi, oq);
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            o[c * NI + 2 * i] = oq[i].x;
            o[c * NI + 2 * i + 1] = oq[i].y;
        }
    }
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
        const int d = pr_d0 + 4 * j;
        if (d < nd) {
            if (a.sim_out) {
#pragma unroll
                for (int g = 0; g < G; ++g)
                    a.sim_out[(((size_t)b * G + g) * D + d_base + d) * hw + pr_p] = (j & 1) ? xq[j / 2][g].y : xq[j / 2][g].x;
            }
            a.out[((size_t)b * hw + pr_p) * D + d_base + d] = o[j];  // cost is hypothesis-last [B,h,w,D]
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------

static int g_tile_cap_bytes = 20 * 1024;  // one window buffer (pmn_set_tuning key 10)

int pmn_tile_set_tuning(int key, int value) {
    if (key != 10 || value < 16 * 1024 || value > 64 * 1024 || (value & 1023)) return PMN_ERR_ARG;
    g_tile_cap_bytes = value;
    return PMN_OK;
}

template <int C, int G, int DT>
static int launch_tile(GatherArgs& a, hipStream_t stream) {
    const int ntx = (a.w + 15) / 16, nty = (a.h + 3) / 4;
    a.ntiles = ntx * nty;
    const int chunks = (a.D + DT - 1) / DT;
    const int cap = g_tile_cap_bytes;
    const size_t lds = (size_t)DT * 64 * 36 + 64 + ((PMN_MLP_FLOATS * 4 + 15) / 16) * 16 + 2 * (size_t)cap;
    if ((size_t)G * DT * 64 * 4 > 2 * (size_t)cap || lds > 160 * 1024) return PMN_ERR_SHAPE;
    auto run = [&](auto kern) {
        static size_t lds_set = 0;
        if (lds > 48 * 1024 && lds > lds_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
                hipSuccess)
                return (int)PMN_ERR_LAUNCH;
            lds_set = lds;
        }
        PMN_LAUNCH(kern, dim3(a.ntiles, chunks, a.B), dim3(PMN_BLOCK), lds, stream, a, cap);
        PMN_CHECK_LAUNCH();
        return (int)PMN_OK;
    };
    return run(gather_tile_kernel<C, G, DT>);
}

int pmn_launch_gather_tile(GatherArgs& a, int C, int G, hipStream_t stream) {
    if (a.D < 4) return PMN_ERR_SHAPE;
    if (C == 64 && G == 8) return launch_tile<64, 8, 8>(a, stream);  // (16 hypotheses x 4 slices of accumulators spill)
    if (C == 32 && G == 8) return launch_tile<32, 8, 16>(a, stream);
    if (C == 16 && G == 4) return a.D <= 8 ? launch_tile<16, 4, 8>(a, stream) : launch_tile<16, 4, 16>(a, stream);
    return PMN_ERR_SHAPE;
}
