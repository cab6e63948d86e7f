// Minimal stand-alone probe for DESIGN_LESSONS.md lesson 46: ONE file, no library, no PyTorch.
//
// The narrowing (profiles/r06_overlap/r06_settle_probes.log) ended at two instructions of the PixelwiseNet launch, both of the form
//     v_pk_add_f32 v[D:D+1], v[A:A+1], v[D:D+1] op_sel:[0,1]
// -- a packed fp32 add IN PLACE on its second source, whose HIGH register is broadcast to both halves (D.lo = A.lo + D.hi,
// D.hi = A.hi + D.hi).  This file issues that instruction (and its neighbours in form space) from inline asm on known operands in a
// self-checking loop, alone and beside a register-only loop of v_mfma_f32_16x16x32_f16 on another stream, and counts wrong results.
//
//   hipcc --offload-arch=gfx950 -O2 -o build/pk_inplace_min scripts/repro/pk_inplace_min.hip
//   build/pk_inplace_min [victim blocks=2048] [iterations=4000] [disturber blocks=1024]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                           \
        }                                                                                      \
    } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// the disturber: back-to-back fp16 MFMAs on registers, no memory, no LDS (scripts/repro/disturbers.hip: k_mfma_f16)
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(0.001f * (threadIdx.x + i));
        b[i] = (_Float16)(0.002f * (threadIdx.x - i));
    }
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.0f) out[threadIdx.x] = c0[0];
}

__device__ __forceinline__ float val(uint32_t a) {  // a small exactly representable float from a hash (|v| <= 2048: products and fmas exact)
    a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16;
    return (float)(int)(a & 0xFFF) - 2048.0f;
}

// FORM 0: v_pk_add_f32 D, A, D op_sel:[0,1]      in place on src1, high half broadcast   (the two instructions of the narrowing)
// FORM 1: v_pk_add_f32 E, A, D op_sel:[0,1]      the same, NOT in place
// FORM 2: v_pk_add_f32 D, A, D op_sel_hi:[1,0]   in place, LOW half broadcast
// FORM 3: v_pk_fma_f32 D, D, A, C op_sel:[1,0,0] in place on src0, high half broadcast (a form the clean builds also contain)
// FORM 4: FORM 0 with D loaded from LDS (ds_read_b64) right before
// FORM 5: FORM 1 behind 16 idle cycles (s_nop 7 twice): is it a forwarding hazard from the VALU instructions that wrote the operands?
// FORMS 6-10, the forms the library DOES contain (6, 7, 8) and the other two measured wrong (9, 10), in this harness's instruction
// context -- the one in which FORM 0 fails a thousand times more often than in pk_opsel_matrix.hip:
//   6: v_pk_add_f32 E, D, A op_sel:[1,0]     7: v_pk_mul_f32 E, D, A op_sel:[1,0]     8: v_pk_fma_f32 E, A, C, D op_sel:[0,0,1]
//   9: v_pk_mul_f32 E, A, D op_sel:[0,1]    10: v_pk_fma_f32 E, A, D, C op_sel:[0,1,0]
template <int FORM>
__global__ __launch_bounds__(256) void victim(unsigned long long* errs, int iters) {
    __shared__ f2 lds[256];
    const uint32_t t = threadIdx.x + blockIdx.x * 256u;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        const float a0 = val(t * 4u + it * 7919u), a1 = val(t * 4u + 1u + it * 7919u), d0 = val(t * 4u + 2u + it * 7919u),
                    d1 = val(t * 4u + 3u + it * 7919u);
        f2 A = {a0, a1}, D = {d0, d1}, E = {0.f, 0.f};
        float e0, e1;
        if constexpr (FORM == 0) {
            asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[0,1]" : "+v"(D) : "v"(A));
            E = D; e0 = a0 + d1; e1 = a1 + d1;
        } else if constexpr (FORM == 1) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(E) : "v"(A), "v"(D));
            e0 = a0 + d1; e1 = a1 + d1;
        } else if constexpr (FORM == 2) {
            asm volatile("v_pk_add_f32 %0, %1, %0 op_sel_hi:[1,0]" : "+v"(D) : "v"(A));
            E = D; e0 = a0 + d0; e1 = a1 + d0;
        } else if constexpr (FORM == 3) {
            const f2 C = {1.0f, 2.0f};
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0]" : "+v"(D) : "v"(A), "v"(C));
            E = D; e0 = __builtin_fmaf(d1, a0, 1.0f); e1 = __builtin_fmaf(d1, a1, 2.0f);
        } else if constexpr (FORM == 5) {
            asm volatile("s_nop 7\n\ts_nop 7\n\tv_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(E) : "v"(A), "v"(D));
            e0 = a0 + d1; e1 = a1 + d1;
        } else if constexpr (FORM == 6) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=&v"(E) : "v"(D), "v"(A));
            e0 = d1 + a0; e1 = d1 + a1;
        } else if constexpr (FORM == 7) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=&v"(E) : "v"(D), "v"(A));
            e0 = d1 * a0; e1 = d1 * a1;
        } else if constexpr (FORM == 8) {
            const f2 C = {3.0f, 5.0f};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=&v"(E) : "v"(A), "v"(C), "v"(D));
            e0 = __builtin_fmaf(a0, 3.0f, d1); e1 = __builtin_fmaf(a1, 5.0f, d1);
        } else if constexpr (FORM == 9) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(E) : "v"(A), "v"(D));
            e0 = a0 * d1; e1 = a1 * d1;
        } else if constexpr (FORM == 10) {
            const f2 C = {3.0f, 5.0f};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=&v"(E) : "v"(A), "v"(D), "v"(C));
            e0 = __builtin_fmaf(a0, d1, 3.0f); e1 = __builtin_fmaf(a1, d1, 5.0f);
        } else {
            lds[threadIdx.x] = D;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            f2 L = lds[(threadIdx.x & ~63) | ((threadIdx.x + 1) & 63)];  // the neighbour lane's pair
            asm volatile("s_waitcnt lgkmcnt(0)\n\tv_pk_add_f32 %0, %1, %0 op_sel:[0,1]" : "+v"(L) : "v"(A));
            E = L;
            const uint32_t tn = ((threadIdx.x & ~63) | ((threadIdx.x + 1) & 63)) + blockIdx.x * 256u;
            const float n1 = val(tn * 4u + 3u + it * 7919u);
            e0 = a0 + n1; e1 = a1 + n1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        bad += (E[0] != e0) | (E[1] != e1);
    }
    if (bad) atomicAdd(errs, (unsigned long long)bad);
}

template <int FORM>
static unsigned long long run(unsigned long long* d_errs, int blocks, int iters, hipStream_t s) {
    CK(hipMemsetAsync(d_errs, 0, 8, s));
    hipLaunchKernelGGL(victim<FORM>, dim3(blocks), dim3(256), 0, s, d_errs, iters);
    CK(hipStreamSynchronize(s));
    unsigned long long h = 0;
    CK(hipMemcpy(&h, d_errs, 8, hipMemcpyDeviceToHost));
    return h;
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 2048, iters = argc > 2 ? atoi(argv[2]) : 4000, dblocks = argc > 3 ? atoi(argv[3]) : 1024;
    unsigned long long* d_errs;
    float* d_out;
    CK(hipMalloc(&d_errs, 8));
    CK(hipMalloc(&d_out, 4096));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const char* names[11] = {"pk_add in place on src1, hi broadcast (op_sel:[0,1])", "pk_add hi broadcast, not in place", "pk_add in place, lo broadcast",
                            "pk_fma in place on src0, hi broadcast", "pk_add in place, hi broadcast, operand from LDS",
                            "pk_add hi broadcast, not in place, behind 16 idle cycles",
                            "pk_add op_sel:[1,0]   (src0 hi; in the library)", "pk_mul op_sel:[1,0]   (src0 hi; in the library)",
                            "pk_fma op_sel:[0,0,1] (src2 hi; in the library)", "pk_mul op_sel:[0,1]   (src1 hi)", "pk_fma op_sel:[0,1,0] (src1 hi)"};
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) hipLaunchKernelGGL(mfma_loop, dim3(dblocks), dim3(256), 0, sb, d_out, 6000000);
        unsigned long long e[11];
        e[0] = run<0>(d_errs, blocks, iters, sa);
        e[1] = run<1>(d_errs, blocks, iters, sa);
        e[2] = run<2>(d_errs, blocks, iters, sa);
        e[3] = run<3>(d_errs, blocks, iters, sa);
        e[4] = run<4>(d_errs, blocks, iters, sa);
        e[5] = run<5>(d_errs, blocks, iters, sa);
        e[6] = run<6>(d_errs, blocks, iters, sa);
        e[7] = run<7>(d_errs, blocks, iters, sa);
        e[8] = run<8>(d_errs, blocks, iters, sa);
        e[9] = run<9>(d_errs, blocks, iters, sa);
        e[10] = run<10>(d_errs, blocks, iters, sa);
        const bool running = pass == 1 && hipStreamQuery(sb) == hipErrorNotReady;
        for (int f = 0; f < 11; ++f)
            printf("%-18s %-62s wrong results: %llu of %llu\n", pass == 0 ? "alone" : (running ? "beside MFMA loop" : "beside (MFMA ENDED)"), names[f], e[f],
                   (unsigned long long)blocks * 256ull * iters);
        if (pass == 1) CK(hipStreamSynchronize(sb));
    }
    // time profile: the not-in-place form launched 60 times in a row while ONE long MFMA kernel runs -- constant rate, decaying, bursts?
    {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, sb));
        hipLaunchKernelGGL(mfma_loop, dim3(dblocks), dim3(256), 0, sb, d_out, 6000000);
        CK(hipEventRecord(e1, sb));
        printf("timeline (ms since the MFMA kernel was queued: wrong results of 2048 x 256 x 500):");
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 60; ++r) {
            const unsigned long long e = run<1>(d_errs, 2048, 500, sa);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf(" %.0f:%llu%s", ms, e, hipStreamQuery(sb) == hipErrorNotReady ? "" : "(ended)");
        }
        CK(hipStreamSynchronize(sb));
        float dur = 0.f;
        CK(hipEventElapsedTime(&dur, e0, e1));
        printf("\nthe MFMA kernel ran %.0f ms\n", dur);
    }
    return 0;
}
