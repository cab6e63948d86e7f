// corr_mfma.hip -- pmn_warp_correlate as CORRELATE-THEN-INTERPOLATE on the fp32 matrix cores (round 4).
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
// Mapping (wave64; every wave owns its tile and runs on its own -- no workgroup barrier after the prologue):
//   * tile = 16 consecutive pixels (linear index over h*w).  lane = (n = lane & 15, k = lane >> 4); lane (n, k) owns the
//     hypotheses d = 4 j + k of pixel n (the `items`).  Hypotheses are walked in chunks of DCH consecutive d (they are sorted along d:
//     a chunk's taps sit on a short piece of the epipolar line).
//   * per (view, chunk): every lane projects its items (same arithmetic as the streaming kernel: v_rcp + one Newton step); a
//     wave reduction gives the bounding box of the live taps = the window (Wd x Hd texels, flattened row-major: texel q).
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

#include "gather_common.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// C channels, G groups, D hypotheses (exact), DCH hypotheses per chunk, QP = texel capacity of a wave's LDS window, NW waves per
// workgroup, PIXELWISE = view weights computed here by PixelwiseNet
template <int C, int G, int D, int DCH, int QP, int NW, bool PIXELWISE>
__global__ __launch_bounds__(64 * NW) void corr_mfma_kernel(const GatherArgs a) {
    constexpr int CG = C / G;            // channels per group: 4 or 8
    constexpr int KS = CG / 4;           // k-steps per group
    constexpr int NPASS = G / 4;         // four groups per pass
    constexpr int CPP = 4 * CG;          // channels per pass: 16 or 32
    constexpr int NJB = CPP / 16;        // dwordx4 loads per lane, N-tile and pass
    constexpr int IPL = D / 4;           // items per lane
    constexpr int IPC = DCH / 4;         // items per lane and chunk
    constexpr int NCH = D / DCH;
    constexpr int PP = 4 * QP + 4;       // floats per pixel block of R (the +4 spreads the pixels' ds_write_b128 over the banks)
    constexpr int TMAX = QP / 16;
    constexpr int NP = (IPL % 4 == 0) ? 2 : 1;  // pairs of items per SimilarityNet evaluation
    static_assert(CG == 4 || CG == 8, "group size must be 4 or 8 channels");
    static_assert(G % 4 == 0 && D % DCH == 0 && DCH % 8 == 0 && QP % 16 == 0 && QP <= 256, "shape");

    extern __shared__ float4 smem4[];
    float* wlds_a = reinterpret_cast<float*>(smem4);                 // SimilarityNet
    float* wlds_b = wlds_a + MLP_LDS_FLOATS;                         // PixelwiseNet
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float* R = wlds_a + (PIXELWISE ? 2 : 1) * MLP_LDS_FLOATS + wave * (16 * PP);
    const int n = lane & 15, k = lane >> 4;

    for (int i = tid; i < PMN_MLP_FLOATS; i += 64 * NW) {
        wlds_a[i] = a.mlp_a[i];
        if (PIXELWISE) wlds_b[i] = a.mlp_b[i];
    }
    __syncthreads();

    const int b = blockIdx.y;
    const int N = a.N, h = a.h, w = a.w, hs = a.hs, ws = a.ws;
    const int hw = h * w;
    const int tile = pmn_xcd_tile(blockIdx.x, a.ntiles) * NW + wave;
    const int p = tile * 16 + n;
    const bool ok = p < hw;
    if (tile * 16 >= hw) return;  // (whole wave; no barrier follows)
    const int y = ok ? p / w : 0, x = ok ? p - y * w : 0;
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
    float dep[IPL];
#pragma unroll
    for (int j = 0; j < IPL; ++j) dep[j] = ok ? a.depth[((size_t)b * D + 4 * j + k) * hw + p] : 0.0f;

    const float sxs = (float)(ws - 1) / (float)(w - 1), sys = (float)(hs - 1) / (float)(h - 1);
    const int wv = w >> a.vw_shift, hwv = (h >> a.vw_shift) * wv;
    const int vw_idx = (y >> a.vw_shift) * wv + (x >> a.vw_shift);

    // items (2 m, 2 m + 1) of a lane ride in the halves of packed registers: the view sum and the MLPs run on v_pk_* as they are
    pmn_f2 acc[IPL / 2][G];
#pragma unroll
    for (int m = 0; m < IPL / 2; ++m)
#pragma unroll
        for (int g = 0; g < G; ++g) acc[m][g] = pmn_f2{0.0f, 0.0f};
    float wsum = 1e-5f;
    const unsigned pixbase = (unsigned)n * PP;

    for (int v = 0; v < N; ++v) {
        const float* P = a.proj + ((size_t)b * N + v) * 16;
        const float rx = (fmaf(P[0], xf, P[1] * yf) + P[2]) * sxs, tx = P[3] * sxs;
        const float ry = (fmaf(P[4], xf, P[5] * yf) + P[6]) * sys, ty = P[7] * sys;
        const float rz = fmaf(P[8], xf, P[9] * yf) + P[10], tz = P[11];
        const char* sbase = reinterpret_cast<const char*>(a.src) + ((size_t)(v * a.B + b) * hs * ws) * (C * 4);
        pmn_f2 sv[(PIXELWISE ? IPL : IPC) / 2][G];  // this view's group similarities (PIXELWISE keeps all D until the view's weight is known)
        float vw = 0.0f;
        if (!PIXELWISE) vw = ok ? a.vw_in[((size_t)b * N + v) * hwv + vw_idx] : 0.0f;

#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            // ---- projection of this lane's IPC items of the chunk ------------------------------------------------------
            int x0[IPC], y0[IPC];
            float w00[IPC], w01[IPC], w10[IPC], w11[IPC];
            bool live[IPC];
            int lox = 0x7fffffff, loy = 0x7fffffff, hix = -0x7fffffff, hiy = -0x7fffffff;
#pragma unroll
            for (int i = 0; i < IPC; ++i) {
                const float dp = dep[c * IPC + i];
                const float pz = fmaf(rz, dp, tz);
                PmnTapsXY t;
                t.x0 = t.y0 = 0;
                t.w00 = t.w01 = t.w10 = t.w11 = 0.0f;
                if (ok && pz > 1e-3f) {  // behind-camera hypotheses sample nothing (reference sentinel, module.py:166-169)
                    float inv = __builtin_amdgcn_rcpf(pz);
                    inv = inv * fmaf(-pz, inv, 2.0f);
                    t = pmn_make_taps_xy(fmaf(rx, dp, tx) * inv, fmaf(ry, dp, ty) * inv, hs, ws);
                }
                x0[i] = t.x0; y0[i] = t.y0;
                w00[i] = t.w00; w01[i] = t.w01; w10[i] = t.w10; w11[i] = t.w11;
                live[i] = (t.w00 + t.w01) + (t.w10 + t.w11) > 0.0f;  // items without an in-range corner stay out of the window
                if (live[i]) {
                    lox = min(lox, t.x0); hix = max(hix, t.x0);
                    loy = min(loy, t.y0); hiy = max(hiy, t.y0);
                }
            }
            const int xmin = pmn_wave_minmax<false>(lox), xmax = pmn_wave_minmax<true>(hix);
            const int ymin = pmn_wave_minmax<false>(loy), ymax = pmn_wave_minmax<true>(hiy);
            const int so = PIXELWISE ? c * IPC : 0;  // where the chunk's items live in sv
#pragma unroll
            for (int m = 0; m < IPC / 2; ++m)
#pragma unroll
                for (int g = 0; g < G; ++g) sv[so / 2 + m][g] = pmn_f2{0.0f, 0.0f};

            if (xmin <= xmax) {  // (wave-uniform) some item of the chunk has a tap inside the source map
                const int Wd = xmax - xmin + 2, Hd = ymax - ymin + 2;
                // R of the window piece [ox, ox+cw) x [oy, oy+ch) (window coordinates) for the four groups of pass `ps`
                auto fill = [&](auto psc, const int ox, const int oy, const int cw, const int ch) __attribute__((always_inline)) {
                    constexpr int ps = decltype(psc)::value;
                    const int Qc = cw * ch;
                    // q -> (qy, qx) = divmod(q, cw) by a 20-bit reciprocal: exact for q < 1024, cw <= 512 (QP <= 256 here)
                    const unsigned magic = (1u << 20) / (unsigned)cw + 1u;
                    const unsigned tex0 = (unsigned)((ymin + oy) * ws + (xmin + ox));  // first texel of the piece (wave-uniform)
                    float4 raw[TMAX][NJB];
#pragma unroll
                    for (int t = 0; t < TMAX; ++t) {
                        const unsigned q = (unsigned)min(16 * t + n, Qc - 1);
                        const unsigned qy = __umul24(q, magic) >> 20;
                        const unsigned qx = q - __umul24(qy, (unsigned)cw);
                        const unsigned tex = tex0 + __umul24(qy, (unsigned)ws) + qx;
                        const unsigned bo = tex * (C * 4u) + (ps * CPP + 4 * k) * 4u;
#pragma unroll
                        for (int jj = 0; jj < NJB; ++jj) raw[t][jj] = *reinterpret_cast<const float4*>(sbase + bo + jj * 64);
                    }
#pragma unroll
                    for (int t = 0; t < TMAX; ++t) {
                        if (16 * t < Qc) {
                            float sT[NJB][4];
#pragma unroll
                            for (int jj = 0; jj < NJB; ++jj) {
                                sT[jj][0] = raw[t][jj].x; sT[jj][1] = raw[t][jj].y; sT[jj][2] = raw[t][jj].z; sT[jj][3] = raw[t][jj].w;
                                pmn_row_transpose4(sT[jj]);
                            }
#pragma unroll
                            for (int gl = 0; gl < 4; ++gl) {
                                f32x4 dd = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                                for (int s = 0; s < KS; ++s) {
                                    const int cof = gl * CG + 4 * s;  // channel offset inside the pass
                                    dd = __builtin_amdgcn_mfma_f32_16x16x4f32(sT[cof / 16][(cof % 16) / 4],
                                                                             refT[ps * NJB + cof / 16][(cof % 16) / 4], dd, 0, 0, 0);
                                }
                                *reinterpret_cast<f32x4*>(R + pixbase + gl * QP + 16 * t + 4 * k) = dd;
                            }
                        }
                    }
                };

                if (Wd * Hd <= QP) {
                    // ---- the whole window fits: every tap of every live item is inside, fixed nw, ne, sw, se blend order --------
                    unsigned rb[IPC];
#pragma unroll
                    for (int i = 0; i < IPC; ++i) rb[i] = pixbase + (unsigned)((y0[i] - ymin) * Wd + (x0[i] - xmin));
                    auto pass = [&](auto psc) __attribute__((always_inline)) {
                        constexpr int ps = decltype(psc)::value;
                        fill(psc, 0, 0, Wd, Hd);
                        pmn_wave_lds_fence();
#pragma unroll
                        for (int i = 0; i < IPC; ++i) {
                            if (live[i]) {
#pragma unroll
                                for (int gl = 0; gl < 4; ++gl) {
                                    const float* rp = R + rb[i] + gl * QP;
                                    const float r00 = rp[0], r01 = rp[1], r10 = rp[Wd], r11 = rp[Wd + 1];
                                    sv[(so + i) / 2][ps * 4 + gl][i & 1] = fmaf(r11, w11[i], fmaf(r10, w10[i], fmaf(r01, w01[i], r00 * w00[i])));
                                }
                            }
                        }
                        pmn_wave_lds_fence();
                    };
                    pass(std::integral_constant<int, 0>{});
                    if constexpr (NPASS > 1) pass(std::integral_constant<int, 1>{});
                } else {
                    // ---- window larger than the buffer: rectangular pieces, taps taken where they fall ------------------------
                    const int cwm = min(Wd, QP), chm = QP / cwm;
                    for (int oy = 0; oy < Hd; oy += chm) {
                        const int ch = min(chm, Hd - oy);
                        for (int ox = 0; ox < Wd; ox += cwm) {
                            const int cw = min(cwm, Wd - ox);
                            auto pass = [&](auto psc) __attribute__((always_inline)) {
                                constexpr int ps = decltype(psc)::value;
                                fill(psc, ox, oy, cw, ch);
                                pmn_wave_lds_fence();
#pragma unroll
                                for (int i = 0; i < IPC; ++i) {
                                    const int ux = x0[i] - xmin - ox, uy = y0[i] - ymin - oy;
#pragma unroll
                                    for (int tp = 0; tp < 4; ++tp) {
                                        const int qx = ux + (tp & 1), qy = uy + (tp >> 1);
                                        const float wt = tp == 0 ? w00[i] : tp == 1 ? w01[i] : tp == 2 ? w10[i] : w11[i];
                                        if (live[i] && (unsigned)qx < (unsigned)cw && (unsigned)qy < (unsigned)ch) {
                                            const float* rp = R + pixbase + (unsigned)(qy * cw + qx);
#pragma unroll
                                            for (int gl = 0; gl < 4; ++gl)
                                                sv[(so + i) / 2][ps * 4 + gl][i & 1] = fmaf(rp[gl * QP], wt, sv[(so + i) / 2][ps * 4 + gl][i & 1]);
                                        }
                                    }
                                }
                                pmn_wave_lds_fence();
                            };
                            pass(std::integral_constant<int, 0>{});
                            if constexpr (NPASS > 1) pass(std::integral_constant<int, 1>{});
                        }
                    }
                }
            }
            if (!PIXELWISE) {
#pragma unroll
                for (int m = 0; m < IPC / 2; ++m)
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        acc[c * (IPC / 2) + m][g] = mul_add_unfused2(acc[c * (IPC / 2) + m][g], sv[m][g] * (1.0f / CG), vw);
            }
        }

        if constexpr (PIXELWISE) {
            // PixelwiseNet on every item of the view, max over D (first arg-max on ties through the ~d low word), then the view sum
            unsigned long long best = 0ull;
#pragma unroll
            for (int m = 0; m < IPL / 2; ++m)
#pragma unroll
                for (int g = 0; g < G; ++g) sv[m][g] = sv[m][g] * (1.0f / CG);
#pragma unroll
            for (int m = 0; m < IPL / 2; ++m) {
                pmn_f2 xq[1][G], rq[1];
#pragma unroll
                for (int g = 0; g < G; ++g) xq[0][g] = sv[m][g];
                mlp_pairs_from_lds<G, 1>(wlds_b, xq, rq);
                const float r2[2] = {rq[0].x, rq[0].y};
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned d = 4 * (2 * m + i) + k;
                    const unsigned long long key = ((unsigned long long)__float_as_uint(pmn_sigmoid(r2[i])) << 32) |
                                                   (unsigned long long)(0xFFFFFFFFu - d);
                    best = key > best ? key : best;
                }
            }
            {   // the pixel's four lanes (n, 0..3)
                unsigned long long o = __shfl_xor(best, 16, 64);
                best = o > best ? o : best;
                o = __shfl_xor(best, 32, 64);
                best = o > best ? o : best;
            }
            const float vwp = __uint_as_float((unsigned)(best >> 32));
#pragma unroll
            for (int m = 0; m < IPL / 2; ++m)
#pragma unroll
                for (int g = 0; g < G; ++g) acc[m][g] = mul_add_unfused2(acc[m][g], sv[m][g], vwp);
            wsum += vwp;
            if (ok && k == 0) {
                const size_t o = ((size_t)b * N + v) * hw + p;
                a.vw_out[o] = vwp;
                if (a.vw_argmax) a.vw_argmax[o] = (int)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull));
            }
        } else {
            wsum += vw;
        }
    }

    if (!ok) return;
#pragma unroll
    for (int m = 0; m < IPL / 2; ++m)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            acc[m][g].x = acc[m][g].x / wsum;
            acc[m][g].y = acc[m][g].y / wsum;
        }
#pragma unroll
    for (int m0 = 0; m0 < IPL / 2; m0 += NP) {
        pmn_f2 xq[NP][G], oq[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int g = 0; g < G; ++g) xq[q][g] = acc[m0 + q][g];
        mlp_pairs_from_lds<G, NP>(wlds_a, xq, oq);
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int d = 4 * (2 * (m0 + q) + i) + k;
                if (a.sim_out) {
#pragma unroll
                    for (int g = 0; g < G; ++g) a.sim_out[(((size_t)b * G + g) * D + d) * hw + p] = acc[m0 + q][g][i];
                }
                a.out[((size_t)b * hw + p) * D + d] = oq[q][i];  // cost is hypothesis-last [B,h,w,D]
            }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------

template <int C, int G, int D, int DCH, int QP, int NW, bool PIXELWISE>
static int launch_corr_mfma(GatherArgs& a, hipStream_t stream) {
    constexpr int PP = 4 * QP + 4;
    const int hw = a.h * a.w;
    const int wtiles = (hw + 15) / 16;
    a.ntiles = (wtiles + NW - 1) / NW;
    const size_t lds = (size_t)((PIXELWISE ? 2 : 1) * MLP_LDS_FLOATS + NW * 16 * PP) * 4;
    auto kern = corr_mfma_kernel<C, G, D, DCH, QP, NW, PIXELWISE>;
    if (lds > 160 * 1024) return PMN_ERR_SHAPE;
    if (lds > 48 * 1024) {
        const int rc = pmn_raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
        if (rc != PMN_OK) return rc;
    }
    hipLaunchKernelGGL(kern, dim3(a.ntiles, a.B), dim3(64 * NW), lds, stream, a);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

#ifndef PMN_CM_QP1
#define PMN_CM_QP1 96
#endif
#ifndef PMN_CM_QP2
#define PMN_CM_QP2 96
#endif
#ifndef PMN_CM_QP3
#define PMN_CM_QP3 96
#endif
#ifndef PMN_CM_NW
#define PMN_CM_NW 1
#endif

// The cascade's own shapes; anything else returns PMN_ERR_SHAPE and the caller takes the streaming kernel.
int pmn_launch_corr_mfma(GatherArgs& a, int C, int G, bool pixelwise, hipStream_t stream) {
    if (!pixelwise) {
        if (C == 16 && G == 4 && a.D == 8) return launch_corr_mfma<16, 4, 8, 8, PMN_CM_QP1, PMN_CM_NW, false>(a, stream);
        if (C == 32 && G == 8 && a.D == 16) return launch_corr_mfma<32, 8, 16, 16, PMN_CM_QP2, PMN_CM_NW, false>(a, stream);
        if (C == 64 && G == 8 && a.D == 32) return launch_corr_mfma<64, 8, 32, 16, PMN_CM_QP3, PMN_CM_NW, false>(a, stream);
        return PMN_ERR_SHAPE;
    }
    if (C == 64 && G == 8 && a.D == 48) return launch_corr_mfma<64, 8, 48, 8, PMN_CM_QP3, PMN_CM_NW, true>(a, stream);
    return PMN_ERR_SHAPE;
}
