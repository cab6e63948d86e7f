// misc.hip -- layout change, confidence epilogue, stand-alone warping op, ABI housekeeping.
#include <cstring>

#include "pmn_common.hpp"

extern "C" int pmn_abi_version(void) { return PMN_ABI_VERSION; }

extern "C" const char* pmn_error_string(int code) {
    switch (code) {
        case PMN_OK: return "ok";
        case PMN_ERR_ARG: return "invalid argument (null pointer or size out of range)";
        case PMN_ERR_SHAPE: return "unsupported shape (channels/groups/neighbours/hypotheses)";
        case PMN_ERR_LAUNCH: return "HIP launch failed";
        default: return "unknown error";
    }
}

// ---- NCHW -> NHWC ------------------------------------------------------------------------------------------------
// 64-pixel x C-channel tile through LDS: reads coalesced along pixels, writes coalesced along channels.
__global__ __launch_bounds__(PMN_BLOCK) void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                int C, int hw) {
    extern __shared__ float tile[];  // [C][65]
    const int b = blockIdx.y, p0 = blockIdx.x * 64, tid = threadIdx.x;
    for (int idx = tid; idx < C * 64; idx += PMN_BLOCK) {
        const int c = idx >> 6, i = idx & 63;
        tile[c * 65 + i] = (p0 + i < hw) ? in[((size_t)b * C + c) * hw + p0 + i] : 0.0f;
    }
    __syncthreads();
    for (int idx = tid; idx < C * 64; idx += PMN_BLOCK) {
        const int i = idx / C, c = idx - i * C;
        if (p0 + i < hw) out[((size_t)b * hw + p0 + i) * C + c] = tile[c * 65 + i];
    }
}

extern "C" int pmn_nchw_to_nhwc(const float* in, float* out, int B, int C, int h, int w, void* stream) {
    if (!in || !out || B < 1 || C < 1 || h < 1 || w < 1) return PMN_ERR_ARG;
    if (C > 128) return PMN_ERR_SHAPE;
    const int hw = h * w;
    PMN_LAUNCH(nchw_to_nhwc_kernel, dim3((hw + 63) / 64, B), dim3(PMN_BLOCK), (size_t)C * 65 * sizeof(float),
                       (hipStream_t)stream, in, out, C, hw);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

// ---- confidence epilogue (reference models/net.py:288-299, models/module.py:184-196) ---------------------------------
__global__ __launch_bounds__(PMN_BLOCK) void confidence_kernel(const float* __restrict__ score, int D, int h, int w, int H,
                                                              int W, float* __restrict__ conf, int* __restrict__ dindex) {
#pragma clang fp contract(off)
    const int q = blockIdx.x * PMN_BLOCK + threadIdx.x, b = blockIdx.y;
    if (q >= H * W) return;
    const int Y = q / W, X = q - Y * W;
    // F.interpolate(mode="nearest"): src = min(floor(dst * (in/out)), in-1), scale in fp32
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const int y = min((int)floorf((float)Y * sy), h - 1);
    const int x = min((int)floorf((float)X * sx), w - 1);
    const size_t hw = (size_t)h * w;
    const float* sp = score + (size_t)b * D * hw + (size_t)y * w + x;
    float idxf = 0.0f;
    for (int d = 0; d < D; ++d) idxf = idxf + sp[d * hw] * (float)d;
    int idx = (int)idxf;  // .long(): truncation
    idx = min(max(idx, 0), D - 1);
    // 4 * avg_pool3d over the zero-padded (1 front, 2 back) volume = sum of p[idx-1 .. idx+2]
    float s = 0.0f;
    for (int j = idx - 1; j <= idx + 2; ++j) s = s + ((j >= 0 && j < D) ? sp[j * hw] : 0.0f);
    s = s / 4.0f * 4.0f;
    conf[(size_t)b * H * W + q] = s;
    // every output pixel mapped onto (y,x) computes the same index; the duplicate stores are benign
    if (dindex) dindex[(size_t)b * hw + (size_t)y * w + x] = idx;
}

// The cascade's own case, H = 2h and W = 2w (the nearest resize of net.py:298 then maps output (Y, X) onto (Y >> 1, X >> 1) exactly):
// one thread per SOURCE pixel evaluates the index and the window sum once and writes its 2 x 2 outputs as two 8-byte stores -- the
// general kernel evaluated them in each of the four output threads (12 strided reads and the regression per output pixel).
__global__ __launch_bounds__(PMN_BLOCK) void confidence2x_kernel(const float* __restrict__ score, int D, int h, int w,
                                                                float* __restrict__ conf, int* __restrict__ dindex) {
#pragma clang fp contract(off)
    const int q = blockIdx.x * PMN_BLOCK + threadIdx.x, b = blockIdx.y;
    if (q >= h * w) return;
    const int y = q / w, x = q - y * w;
    const size_t hw = (size_t)h * w;
    const float* sp = score + (size_t)b * D * hw + q;
    float idxf = 0.0f;
    for (int d = 0; d < D; ++d) idxf = idxf + sp[d * hw] * (float)d;
    int idx = (int)idxf;
    idx = min(max(idx, 0), D - 1);
    float s = 0.0f;
    for (int j = idx - 1; j <= idx + 2; ++j) s = s + ((j >= 0 && j < D) ? sp[j * hw] : 0.0f);
    s = s / 4.0f * 4.0f;
    float* o = conf + (size_t)b * 4 * hw + (size_t)(2 * y) * (2 * w) + 2 * x;
    *reinterpret_cast<float2*>(o) = make_float2(s, s);
    *reinterpret_cast<float2*>(o + 2 * w) = make_float2(s, s);
    if (dindex) dindex[(size_t)b * hw + q] = idx;
}

extern "C" int pmn_confidence(const float* score, int B, int D, int h, int w, int H, int W, float* confidence_out,
                              int* depth_index_out, void* stream) {
    if (!score || !confidence_out || B < 1 || D < 1 || h < 1 || w < 1 || H < 1 || W < 1) return PMN_ERR_ARG;
    if (H == 2 * h && W == 2 * w && (reinterpret_cast<uintptr_t>(confidence_out) & 7) == 0)
        PMN_LAUNCH(confidence2x_kernel, dim3((h * w + PMN_BLOCK - 1) / PMN_BLOCK, B), dim3(PMN_BLOCK), 0,
                           (hipStream_t)stream, score, D, h, w, confidence_out, depth_index_out);
    else
        PMN_LAUNCH(confidence_kernel, dim3((H * W + PMN_BLOCK - 1) / PMN_BLOCK, B), dim3(PMN_BLOCK), 0,
                           (hipStream_t)stream, score, D, h, w, H, W, confidence_out, depth_index_out);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

// ---- stand-alone differentiable_warping (reference models/module.py:130-181) ---------------------------------------
__global__ __launch_bounds__(PMN_BLOCK) void warping_kernel(const float* __restrict__ src, const float* __restrict__ proj,
                                                           const float* __restrict__ depth, int C, int D, int h, int w,
                                                           int hs, int ws, float* __restrict__ warped) {
    const int hw = h * w;
    const int i = blockIdx.x * PMN_BLOCK + threadIdx.x, b = blockIdx.y;
    if (i >= D * hw) return;
    const int d = i / hw, p = i - d * hw, y = p / w, x = p - y * w;
    float ix, iy;
    pmn_warp_position(proj + (size_t)b * 16, (float)x, (float)y, depth[((size_t)b * D + d) * hw + p], h, w, hs, ws, ix,
                      iy);
    const PmnTaps t = pmn_make_taps(ix, iy, hs, ws);
    const float* sp = src + (size_t)b * C * hs * ws + t.off;
    for (int c = 0; c < C; ++c) {
        const float* q = sp + (size_t)c * hs * ws;
        const float v = fmaf(q[ws + 1], t.w11, fmaf(q[ws], t.w10, fmaf(q[1], t.w01, q[0] * t.w00)));
        warped[(((size_t)b * C + c) * D + d) * hw + p] = v;
    }
}

extern "C" int pmn_differentiable_warping(const float* src_nchw, const float* rel_proj, const float* depth, int B, int C,
                                          int D, int h, int w, int hs, int ws, float* warped, void* stream) {
    if (!src_nchw || !rel_proj || !depth || !warped) return PMN_ERR_ARG;
    if (B < 1 || C < 1 || D < 1 || h < 2 || w < 2 || hs < 2 || ws < 2) return PMN_ERR_ARG;
    PMN_LAUNCH(warping_kernel, dim3((D * h * w + PMN_BLOCK - 1) / PMN_BLOCK, B), dim3(PMN_BLOCK), 0,
                       (hipStream_t)stream, src_nchw, rel_proj, depth, C, D, h, w, hs, ws, warped);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

// ---- per-stage relative projections (reference models/net.py:225-231 + models/module.py:148) -------------------------------
// For every stage s (scale 1/8, 1/4, 1/2 ...), batch b and source view v:
//   K_s = K with rows 0,1 scaled;  P = [[K_s E[:3,:4]], [E[3,:]]] ;  rel = P_src @ inverse(P_ref)
// One thread per (stage, b, v).  The products are formed in fp32 exactly like the reference's torch.matmul (k-ascending
// FMA-free sums); the 4x4 inverse is a double-precision Gauss-Jordan with partial pivoting rounded to fp32 (torch.inverse
// is an fp32 LU; both are within ~1e-7 relative of the exact inverse, i.e. ~1e-4 px at 800-px coordinates).
__global__ void stage_projections_kernel(const float* __restrict__ intr, const float* __restrict__ extr, int B, int V,
                                         int nstages, float scale0, float* __restrict__ rel) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nsrc = V - 1;
    if (i >= nstages * B * nsrc) return;
    const int v = i % nsrc + 1, b = (i / nsrc) % B, s = i / (nsrc * B);
    float scale = scale0;
    for (int k = 0; k < s; ++k) scale *= 2.0f;
    // Every array below is indexed with compile-time constants only (loops fully unrolled, the pivot row exchanged by predicated
    // swaps): the kernel must not touch SCRATCH.  Round 5: it was the one kernel of the default forward with a private-memory frame
    // (272 bytes per lane: `A[piv][c]` with a run-time row), and forwards replayed as HIP graphs CONCURRENTLY on several streams then
    // came out with slightly -- now and then entirely -- wrong projections (scripts/graph_overlap_probe.py: 63 of 300 rounds; none with
    // one hardware queue, none for eager launches, none for graphs of ATen kernels).  Same operations in the same order as before.
    float P[2][16];
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        const int view = w == 0 ? 0 : v;
        const float* K = intr + ((size_t)b * V + view) * 9;
        const float* E = extr + ((size_t)b * V + view) * 16;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float k0 = r < 2 ? K[r * 3 + 0] * scale : K[r * 3 + 0];
            const float k1 = r < 2 ? K[r * 3 + 1] * scale : K[r * 3 + 1];
            const float k2 = r < 2 ? K[r * 3 + 2] * scale : K[r * 3 + 2];
#pragma unroll
            for (int c = 0; c < 4; ++c) P[w][r * 4 + c] = (k0 * E[0 * 4 + c] + k1 * E[1 * 4 + c]) + k2 * E[2 * 4 + c];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) P[w][12 + c] = E[12 + c];
    }
    // inverse of P[0] in fp64: Gauss-Jordan with partial pivoting
    double A[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            A[r][c] = (double)P[0][r * 4 + c];
            A[r][4 + c] = r == c ? 1.0 : 0.0;
        }
#pragma unroll
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        double best = fabs(A[col][col]);
#pragma unroll
        for (int r = col + 1; r < 4; ++r) {  // (the first row with the strictly largest magnitude, as `fabs(A[r][col]) > fabs(A[piv][col])` picks)
            const double m = fabs(A[r][col]);
            if (m > best) {
                best = m;
                piv = r;
            }
        }
#pragma unroll
        for (int r = col + 1; r < 4; ++r) {  // exchange rows col and piv
            const bool sw = piv == r;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const double lo = A[col][c], hi = A[r][c];
                A[col][c] = sw ? hi : lo;
                A[r][c] = sw ? lo : hi;
            }
        }
        const double inv = 1.0 / A[col][col];
#pragma unroll
        for (int c = 0; c < 8; ++c) A[col][c] *= inv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            const double f = A[r][col];
#pragma unroll
            for (int c = 0; c < 8; ++c) A[r][c] -= f * A[col][c];
        }
    }
    float* o = rel + (((size_t)s * B + b) * nsrc + (v - 1)) * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = acc + P[1][r * 4 + k] * (float)A[k][4 + c];
            o[r * 4 + c] = acc;
        }
}

// intrinsics [B,V,3,3], extrinsics [B,V,4,4] (view 0 = reference) -> rel [nstages, B, V-1, 4, 4]; stage s uses intrinsics
// rows 0,1 scaled by scale0 * 2^s (reference net.py:221-232: scale0 = 0.125, three stages)
extern "C" int pmn_stage_projections(const float* intrinsics, const float* extrinsics, int B, int V, int nstages,
                                     float scale0, float* rel, void* stream) {
    if (!intrinsics || !extrinsics || !rel || B < 1 || V < 2 || nstages < 1) return PMN_ERR_ARG;
    const int total = nstages * B * (V - 1);
    PMN_LAUNCH(stage_projections_kernel, dim3((total + 63) / 64), dim3(64), 0, (hipStream_t)stream, intrinsics,
                       extrinsics, B, V, nstages, scale0, rel);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

// ---- Refinement's depth normalisation (reference models/net.py:104-106: (depth_0 - depth_min) / (depth_max - depth_min)) -----------
// Two ATen kernels in round 5, and the only launches of a forward that did not come from this library: with them here a whole
// forward is recordable as a launch plan (plan.hip).  Same operations as ATen's sub / sub / div on float32 tensors: IEEE
// subtraction and correctly rounded division (hipcc's default for `/`), so the same bits.
__global__ __launch_bounds__(PMN_BLOCK) void normalize_depth_kernel(const float* __restrict__ depth, const float* __restrict__ depth_min,
                                                                    const float* __restrict__ depth_max, int n, float* __restrict__ out) {
#pragma clang fp contract(off)
    const int b = blockIdx.y, i = blockIdx.x * PMN_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float lo = depth_min[b], span = depth_max[b] - lo;
    out[(size_t)b * n + i] = (depth[(size_t)b * n + i] - lo) / span;
}

extern "C" int pmn_normalize_depth(const float* depth, const float* depth_min, const float* depth_max, int B, int n, float* out,
                                   void* stream) {
    if (!depth || !depth_min || !depth_max || !out || B < 1 || n < 1) return PMN_ERR_ARG;
    PMN_LAUNCH(normalize_depth_kernel, dim3((n + PMN_BLOCK - 1) / PMN_BLOCK, B), dim3(PMN_BLOCK), 0, (hipStream_t)stream, depth,
               depth_min, depth_max, n, out);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

// ---- opt-in range check for the fp16-split entry points (include/pmn_hip.h, "fp16-split entry points: accepted magnitudes") -------------
// Sets bit 0 of *flag when some |x| >= 65504 or x is not finite: `hi = fp16(x)` would then be inf and the convolution's output inf / NaN
// where an fp32 convolution is finite.  Not part of any forward by default: patchmatchnet_amd/ops.py launches it on the inputs of the
// _f16s entry points when PMN_CHECK_F16_DOMAIN=1 (the domain was a caller's contract nothing could detect).
__global__ __launch_bounds__(PMN_BLOCK) void check_f16_domain_kernel(const float* __restrict__ x, long long n, int* __restrict__ flag) {
    bool bad = false;
    for (long long i = blockIdx.x * (long long)PMN_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * PMN_BLOCK) {
        const float v = fabsf(x[i]);
        bad |= !(v < 65504.0f);  // also true for NaN
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

extern "C" int pmn_check_f16_domain(const float* x, long long n, int* flag, void* stream) {
    if (!x || !flag || n < 1) return PMN_ERR_ARG;
    const long long blocks = (n + PMN_BLOCK - 1) / PMN_BLOCK;
    PMN_LAUNCH(check_f16_domain_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(PMN_BLOCK), 0, (hipStream_t)stream, x, n, flag);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}
