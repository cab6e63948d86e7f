// Shared device helpers for the PatchmatchNet MI355X (gfx950) kernels.
//
// Arithmetic conventions (parity with the PyTorch ops the reference calls):
//  * coordinate maths is compiled under `#pragma clang fp contract(off)` so hipcc does not contract it into
//    FMAs -- the reference materialises every intermediate tensor (one rounding per op), and bit-identical
//    sample positions make the tap sets identical to the oracle's;
//  * divisions are IEEE (hipcc default: correctly rounded fp32 divide), including -- since round 4 -- the warp's perspective
//    division and normalisation chain (pmn_pose_position); the exception left is the depth-weight sigmoid / normalisation of
//    pmn_aggregate_regress (v_exp_f32, v_rcp_f32 + one Newton step, relative error < 4e-7), measured NOT to move a pixel
//    (profiles/r04_ieee_attribution.md);
//  * bilinear taps follow ATen's grid_sampler_2d: corner weights (x1-ix)*(y1-iy) ..., corners accumulated in
//    the order nw, ne, sw, se, out-of-range corners contribute nothing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pmn_hip.h"

#define PMN_BLOCK 256

__device__ __forceinline__ float pmn_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- lesson 46: no packed fp32 instruction may take its SECOND source's HIGH register for the LOW half of the result ------------------
// Measured on MI355X (scripts/repro/pk_opsel_matrix.hip, pk_inplace_min.hip; profiles/r06_overlap/r06_pk_opsel_matrix.log): v_pk_add_f32 /
// v_pk_mul_f32 / v_pk_fma_f32 with op_sel:[x,1] / op_sel:[x,1,x] read that operand as ZERO now and then -- for a 16-lane pass -- while
// waves of another kernel on the same CU issue v_mfma_f32_16x16x32_f16 / _bf16 (this library's fp16-split convolutions).  Alone, or
// beside any other instruction mix, the same instruction is right two billion times out of two billion, which is why three rounds of
// single-stream parity tests never saw it and why overlapped forwards differed from the eager forward in "a few thousand pixels' fifth
// digit".  Every other op_sel / op_sel_hi combination is right beside the same MFMAs.  hipcc emits the form when a scalar that it
// knows to be the HIGH half of a 64-bit register pair (the .y / .w of an LDS vector load, of a float4 built by DPP moves ...) is
// broadcast into packed math; a scalar in a register of its own is broadcast with op_sel_hi (low half to both), which is safe.  So
// every scalar that is broadcast into packed math passes through pmn_settle first: an EMPTY inline asm with the value as a
// read-write operand -- no instruction; hipcc can no longer see which half of which pair the value came from and gives it a register
// of its own.  scripts/isa_pk_opsel.py lists the form in a binary; tests/test_isa_hazards.py holds libpmn_hip.so to ZERO sites (the
// guard proper: the pins are just how this source gets there); tests/test_overlap_gpu.py re-checks every launch beside the MFMA
// kernels on the device.  -DPMN_NO_SETTLE compiles the pins out (probe builds: the form comes back in ~20 kernels).
__device__ __forceinline__ float pmn_settle(float v) {
#ifndef PMN_NO_SETTLE
    asm volatile("" : "+v"(v));
#endif
    return v;
}
__device__ __forceinline__ float4 pmn_settle4(float4 v) {
    return make_float4(pmn_settle(v.x), pmn_settle(v.y), pmn_settle(v.z), pmn_settle(v.w));
}

// ---- bilinear tap set --------------------------------------------------------------------------------------
// Texel index of the (clamped) north-west corner plus the weights of the 2x2 block anchored there.  Corners that
// ATen would skip as out of range get weight 0 and a clamped (always legal) address; when the true NW corner is
// at -1 (or the true SE corner at `size`) the surviving weights are moved onto the clamped block so every
// in-range corner keeps exactly the weight ATen gives it, in the same nw,ne,sw,se order.
struct PmnTaps {
    int off;                   // y0c * ws + x0c
    float w00, w01, w10, w11;  // weights of (y0c,x0c) (y0c,x0c+1) (y0c+1,x0c) (y0c+1,x0c+1)
};

__device__ __forceinline__ void pmn_axis(float pos, int size, int& i0c, float& wa, float& wb) {
#pragma clang fp contract(off)
    float f0 = floorf(pos);
    float f1 = f0 + 1.0f;
    float w_lo = f1 - pos;  // weight of the low (west / north) corner
    float w_hi = pos - f0;  // weight of the high (east / south) corner
    // clamp before the int conversion so wild coordinates cannot overflow
    int i0 = (int)fminf(fmaxf(f0, -2.0f), (float)size);
    i0c = min(max(i0, 0), size - 2);
    wa = 0.0f;
    wb = 0.0f;
    if (i0 == i0c) {  // both corners addressable at their natural slots (hi corner may still be == size-1)
        wa = w_lo;
        wb = w_hi;
    } else if (i0 == -1) {  // low corner out of range; hi corner is texel 0 = slot a
        wa = w_hi;
    } else if (i0 == size - 1) {  // hi corner out of range; low corner is texel size-1 = slot b
        wb = w_lo;
    }
}

__device__ __forceinline__ PmnTaps pmn_make_taps(float ix, float iy, int hs, int ws) {
#pragma clang fp contract(off)
    int x0c, y0c;
    float ax, bx, ay, by;
    pmn_axis(ix, ws, x0c, ax, bx);
    pmn_axis(iy, hs, y0c, ay, by);
    PmnTaps t;
    t.off = y0c * ws + x0c;
    t.w00 = ax * ay;
    t.w01 = bx * ay;
    t.w10 = ax * by;
    t.w11 = bx * by;
    return t;
}

// The same tap set with the corner position kept as (x0, y0): the windowed gather kernel addresses an LDS window with it.
struct PmnTapsXY {
    int x0, y0;                // clamped north-west texel
    float w00, w01, w10, w11;  // weights of (y0,x0) (y0,x0+1) (y0+1,x0) (y0+1,x0+1)
};

__device__ __forceinline__ PmnTapsXY pmn_make_taps_xy(float ix, float iy, int hs, int ws) {
#pragma clang fp contract(off)
    float ax, bx, ay, by;
    PmnTapsXY t;
    pmn_axis(ix, ws, t.x0, ax, bx);
    pmn_axis(iy, hs, t.y0, ay, by);
    t.w00 = ax * ay;
    t.w01 = bx * ay;
    t.w10 = ax * by;
    t.w11 = bx * by;
    return t;
}

// ---- sample positions ---------------------------------------------------------------------------------------

// F.grid_sample un-normalisation (ATen grid_sampler_unnormalize)
__device__ __forceinline__ float pmn_unnorm_align(float c, int size) {
#pragma clang fp contract(off)
    return ((c + 1.0f) / 2.0f) * (float)(size - 1);
}
__device__ __forceinline__ float pmn_unnorm_noalign(float c, int size) {
#pragma clang fp contract(off)
    return ((c + 1.0f) * (float)size - 1.0f) / 2.0f;
}

// Homography warp of reference pixel (x,y) at depth d into the source map (reference models/module.py:161-181), in the
// reference's own sequence of IEEE operations: rot = R [x y 1]^T, p = rot * d + t, the perspective DIVISIONS, the normalisation to
// [-1, 1] with (size-1)/2 and grid_sample's un-normalisation.  Every kernel that warps uses these two functions since round 4:
// rounds 1-3 took the division as v_rcp + one Newton step with the normalise / un-normalise round trip folded into one scale --
// positions within 1e-4 px of these, and exactly that was 9 of 10 of the free-running outlier pixels against the reference's own
// output (6.5e-4 -> 7.3e-5 of the final-depth pixels beyond 1e-3 at 1600x1200; profiles/r04_ieee_attribution.md).
// P = relative projection src_proj @ inv(ref_proj), row-major 4x4.
struct PmnPose {
    float rx, ry, rz, tx, ty, tz;  // rot_xyz of one reference pixel for one view, and the translation
    float cx, cy, rcx, rcy;        // the normalisation divisors (w-1)/2, (h-1)/2 and their refined reciprocals
};

// IEEE fp32 division the way hipcc expands it (LLVM's LowerFDIV32: v_rcp_f32, one Newton refinement of the reciprocal, the quotient
// with two fma corrections), WITHOUT its v_div_scale / v_div_fmas / v_div_fixup shell: that shell only acts on operands whose
// exponents are extreme (denormal denominators, |numerator| < 2^-103, quotients near overflow) -- never on depths, pixel coordinates
// or image sizes -- so for every operand this file divides the sequence below IS the compiler's, bit for bit, at 8 instead of 12
// instructions, and its first three (the refined reciprocal) are shared by divisions with a common denominator.  The warp has four
// divisions per (pixel, hypothesis, view): 48 -> 23 instructions (round 4: profiles/r04_ieee_attribution.md).
__device__ __forceinline__ float pmn_uniform(float v) {  // a wave-uniform value, moved to a scalar register
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ float pmn_rcp_refined(float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = fmaf(-d, r0, 1.0f);
    return fmaf(e, r0, r0);
}
__device__ __forceinline__ float pmn_div_by(float n, float d, float r) {  // n / d with r = pmn_rcp_refined(d)
#pragma clang fp contract(off)
    const float q0 = n * r;
    const float e2 = fmaf(-d, q0, n);
    const float q1 = fmaf(e2, r, q0);
    const float e3 = fmaf(-d, q1, n);
    return fmaf(e3, r, q1);
}

__device__ __forceinline__ float pmn_div(float n, float d) { return pmn_div_by(n, d, pmn_rcp_refined(d)); }  // n / d, 8 instructions

__device__ __forceinline__ PmnPose pmn_make_pose(const float* __restrict__ P, float x, float y, int h, int w) {
#pragma clang fp contract(off)
    PmnPose q;
    q.rx = (P[0] * x + P[1] * y) + P[2];
    q.ry = (P[4] * x + P[5] * y) + P[6];
    q.rz = (P[8] * x + P[9] * y) + P[10];
    q.tx = P[3];
    q.ty = P[7];
    q.tz = P[11];
    // launch constants: held in SGPRs (hipcc leaves a uniform float that came out of the VALU in a vector register per lane)
    q.cx = pmn_uniform((float)(w - 1) / 2.0f);
    q.cy = pmn_uniform((float)(h - 1) / 2.0f);
    q.rcx = pmn_uniform(pmn_rcp_refined(q.cx));
    q.rcy = pmn_uniform(pmn_rcp_refined(q.cy));
    return q;
}

// returns false for a hypothesis behind / on the source camera: (w, h, 1) then lands outside every tap (the reference's sentinel)
__device__ __forceinline__ bool pmn_pose_position(const PmnPose& q, float d, int h, int w, int hs, int ws, float& ix, float& iy) {
#pragma clang fp contract(off)
    float px = q.rx * d + q.tx;
    float py = q.ry * d + q.ty;
    float pz = q.rz * d + q.tz;
    const bool front = pz > 1e-3f;
    if (!front) {
        px = (float)w;
        py = (float)h;
        pz = 1.0f;
    }
    const float rz = pmn_rcp_refined(pz);
    const float gx = pmn_div_by(px, pz, rz), gy = pmn_div_by(py, pz, rz);                  // proj_xyz[:, :2] / z
    const float xn = pmn_div_by(gx, q.cx, q.rcx) - 1.0f, yn = pmn_div_by(gy, q.cy, q.rcy) - 1.0f;  // x / ((w - 1) / 2) - 1
    ix = pmn_unnorm_align(xn, ws);
    iy = pmn_unnorm_align(yn, hs);
    return front;
}

__device__ __forceinline__ void pmn_warp_position(const float* __restrict__ P, float x, float y, float d, int h,
                                                  int w, int hs, int ws, float& ix, float& iy) {
    const PmnPose q = pmn_make_pose(P, x, y, h, w);
    pmn_pose_position(q, d, h, w, hs, ws, ix, iy);
}

// Neighbour k of pixel (x,y): fixed table offset + learned offset, normalised with (size-1)/2 (get_grid,
// reference models/patchmatch.py:409-421) but sampled with align_corners=False + border clip.  The two divisions by the launch
// constants (w-1)/2, (h-1)/2 are IEEE divisions in hipcc's own fma sequence (pmn_div_by: same bits as `X / c`, 5 instructions each
// with the refined reciprocal hoisted out of the neighbour loop, instead of 12).
__device__ __forceinline__ void pmn_neighbor_position(float x, float y, int dy, int dx, float offx, float offy,
                                                      int h, int w, float& ix, float& iy) {
#pragma clang fp contract(off)
    const float cx = pmn_uniform((float)(w - 1) / 2.0f), cy = pmn_uniform((float)(h - 1) / 2.0f);
    const float rcx = pmn_uniform(pmn_rcp_refined(cx)), rcy = pmn_uniform(pmn_rcp_refined(cy));  // loop-invariant, in SGPRs
    float X = x + ((float)dx + offx);
    float Y = y + ((float)dy + offy);
    float xn = pmn_div_by(X, cx, rcx) - 1.0f;
    float yn = pmn_div_by(Y, cy, rcy) - 1.0f;
    ix = fminf(fmaxf(pmn_unnorm_noalign(xn, w), 0.0f), (float)(w - 1));
    iy = fminf(fmaxf(pmn_unnorm_noalign(yn, h), 0.0f), (float)(h - 1));
}

// XCD-aware tile order: hardware places block b on XCD b % 8; give every XCD a contiguous run of tiles so
// neighbouring tiles (which share source texels) hit the same private L2.  Bijective for any tile count.
__device__ __forceinline__ int pmn_xcd_tile(int bid, int ntiles) {
    const int NX = 8;
    int per = ntiles / NX, rem = ntiles % NX;
    int xcd = bid % NX, idx = bid / NX;
    // XCDs [0, rem) own (per+1) tiles, the rest own `per`
    return xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
}

// ---- launches --------------------------------------------------------------------------------------------------------------
// Every kernel of the library is launched through PMN_LAUNCH (same argument order as hipLaunchKernelGGL).  Normally that is one
// hipLaunchKernel on the caller's stream.  While the calling THREAD records a launch plan (pmn_plan_begin .. pmn_plan_end, plan.hip)
// nothing is launched: the kernel's host symbol, its grid and a copy of its arguments -- all passed by value, neighbour tables
// included -- are appended to the plan, and pmn_plan_launch later replays the list with plain hipLaunchKernel calls from C: one
// library call per forward instead of ~55 from Python, and no HIP graph.
#include <tuple>
#include <utility>
struct PmnPlan;
extern thread_local PmnPlan* pmn_tls_plan;  // the plan this thread records into, or null (plan.hip)
int pmn_plan_append(PmnPlan* plan, const void* func, dim3 grid, dim3 block, size_t lds, int nargs, void* const* args,
                    const size_t* sizes, const size_t* aligns);

template <typename... P, size_t... I>
inline void pmn_launch_tuple(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t stream, std::tuple<P...>& params,
                             std::index_sequence<I...>) {
    void* argv[] = {static_cast<void*>(&std::get<I>(params))...};
    if (__builtin_expect(pmn_tls_plan != nullptr, 0)) {
        const size_t sizes[] = {sizeof(P)...}, aligns[] = {alignof(P)...};
        (void)pmn_plan_append(pmn_tls_plan, reinterpret_cast<const void*>(kernel), grid, block, lds, (int)sizeof...(P), argv, sizes,
                              aligns);  // (a failed append poisons the plan: pmn_plan_end reports it)
        return;
    }
    (void)hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, argv, lds, stream);  // PMN_CHECK_LAUNCH reads the error
}

template <typename... P, typename... A>
inline void pmn_launch(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t stream, A&&... args) {
    static_assert(sizeof...(P) == sizeof...(A), "PMN_LAUNCH: argument count differs from the kernel's parameter list");
    static_assert((std::is_trivially_copyable<P>::value && ...), "kernel parameters are copied bytewise into launch plans");
    std::tuple<P...> params(std::forward<A>(args)...);
    pmn_launch_tuple(kernel, grid, block, lds, stream, params, std::index_sequence_for<P...>{});
}
#define PMN_LAUNCH(kernel, ...) pmn_launch(kernel, __VA_ARGS__)

#define PMN_CHECK_LAUNCH()                                  \
    do {                                                    \
        if (pmn_tls_plan != nullptr) break; /* recording */ \
        hipError_t e_ = hipGetLastError();                  \
        if (e_ != hipSuccess) return PMN_ERR_LAUNCH;        \
    } while (0)

// Raises hipFuncAttributeMaxDynamicSharedMemorySize of ``func`` to ``bytes`` on the CURRENT device, once per (kernel, device,
// size) -- the attribute is per device and the call is a driver round trip, so it is cached, keyed by device (a process may
// launch on several GPUs: ops.py brackets every call with torch.cuda.device(tensor.device)) and guarded by a mutex (eval.py and
// bench.py launch from one thread per process, but nothing in the ABI forbids two).
#include <mutex>
#include <vector>
inline int pmn_raise_dynamic_lds(const void* func, size_t bytes) {
    struct Entry { const void* func; int device; size_t bytes; };
    static std::mutex mu;
    static std::vector<Entry> seen;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return PMN_ERR_LAUNCH;
    std::lock_guard<std::mutex> lock(mu);
    for (Entry& e : seen)
        if (e.func == func && e.device == dev) {
            if (e.bytes >= bytes) return PMN_OK;
            if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return PMN_ERR_LAUNCH;
            e.bytes = bytes;
            return PMN_OK;
        }
    if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return PMN_ERR_LAUNCH;
    seen.push_back({func, dev, bytes});
    return PMN_OK;
}
