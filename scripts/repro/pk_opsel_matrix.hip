// Which packed-fp32 instruction forms compute wrong results beside which matrix instructions?  (DESIGN_LESSONS.md lesson 46; the
// one-instruction reproducer is pk_inplace_min.hip, this is its matrix.)  One file, no library, no PyTorch.
//
// victim   v_pk_add_f32 / v_pk_mul_f32 with every op_sel x op_sel_hi combination of their two sources, v_pk_mov_b32 with its four, v_pk_fma_f32
//          with one selector changed at a time; operands = exactly representable small numbers written by VALU instructions; each result is compared
//          with the same selection done by scalar instructions; a wrong result is counted, the first one is kept.
// beside   a register-only loop of ONE matrix instruction (no memory, no LDS) launched on another stream, or a packed-VALU loop.
//
//   hipcc --offload-arch=gfx950 -O2 -o build/pk_opsel_matrix scripts/repro/pk_opsel_matrix.hip
//   build/pk_opsel_matrix [victim blocks=1024] [iterations=20000] [workgroups of the other kernel=512]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                           \
        }                                                                                      \
    } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef int i4 __attribute__((ext_vector_type(4)));

// ---- disturbers: four independent accumulators, back to back -------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void disturb(float* out, int iters, volatile int* flags) {
    float sink = 0.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) flags[0] = 1;  // started
    if constexpr (KIND == 0) {  // v_mfma_f32_16x16x32_f16
        h8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
        f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
        }
        sink = c0[0] + c1[1] + c2[2] + c3[3];
    } else if constexpr (KIND == 1) {  // v_mfma_f32_16x16x32_bf16
        b8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
        f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        }
        sink = c0[0] + c1[1] + c2[2] + c3[3];
    } else if constexpr (KIND == 2) {  // v_mfma_f32_32x32x16_f16
        h8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
        f16v c0, c1;
        for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        }
        sink = c0[0] + c1[1];
    } else if constexpr (KIND == 3) {  // v_mfma_f32_16x16x16_f16 (the gfx90a-era shape)
        h4 a, b;
        for (int i = 0; i < 4; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
        f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c3, 0, 0, 0);
        }
        sink = c0[0] + c1[1] + c2[2] + c3[3];
    } else if constexpr (KIND == 4) {  // v_mfma_f32_16x16x4_f32
        const float a = 0.001f * threadIdx.x, b = 0.002f * threadIdx.x;
        f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        }
        sink = c0[0] + c1[1] + c2[2] + c3[3];
    } else if constexpr (KIND == 5) {  // v_mfma_i32_16x16x64_i8
        i4 a = {(int)threadIdx.x, 3, 5, 7}, b = {11, (int)threadIdx.x, 13, 17};
        i4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
        }
        sink = (float)(c0[0] + c1[1] + c2[2] + c3[3]);
    } else {  // packed fp32 VALU only
        f2 x = {0.001f * threadIdx.x, 1.0f}, y = {1.0001f, 0.9999f}, z0 = {0, 0}, z1 = z0, z2 = z0, z3 = z0;
        for (int it = 0; it < iters; ++it) {
            z0 = __builtin_elementwise_fma(x, y, z0); z1 = __builtin_elementwise_fma(x, y, z1);
            z2 = __builtin_elementwise_fma(x, y, z2); z3 = __builtin_elementwise_fma(x, y, z3);
        }
        sink = z0[0] + z1[1] + z2[0] + z3[1];
    }
    if (sink == 12345.678f) out[threadIdx.x] = sink;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) flags[1] = 1;  // (one of) the last workgroups has finished
}

// ---- victim --------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float val(uint32_t a) {
    a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16;
    return (float)(int)(a & 0xFF) - 128.0f;  // |v| <= 128: sums, products and fmas of three of them are exact in fp32
}

struct First {
    unsigned long long count;
    float a[2], b[2], c[2], got[2], want[2];
};

#define PK2(MN, S0, S1, H0, H1) asm volatile(MN " %0, %1, %2 op_sel:[" #S0 "," #S1 "] op_sel_hi:[" #H0 "," #H1 "]" : "=&v"(E) : "v"(A), "v"(B))
#define PK3(S0, S1, S2, H0, H1, H2) \
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[" #S0 "," #S1 "," #S2 "] op_sel_hi:[" #H0 "," #H1 "," #H2 "]" : "=&v"(E) : "v"(A), "v"(B), "v"(C))

// form = op * 64 + sel * 8 + selhi   (op 0 add, 1 mul, 2 fma; sel / selhi: bit i = source i)
template <int form>
__device__ __forceinline__ f2 one(const f2 A, const f2 B, const f2 C) {
    f2 E = {0.f, 0.f};
    switch (form) {
#define ROW2(OP, MN)                                                                                                     \
    case OP * 64 + 0 * 8 + 3: PK2(MN, 0, 0, 1, 1); break; case OP * 64 + 1 * 8 + 3: PK2(MN, 1, 0, 1, 1); break;           \
    case OP * 64 + 2 * 8 + 3: PK2(MN, 0, 1, 1, 1); break; case OP * 64 + 3 * 8 + 3: PK2(MN, 1, 1, 1, 1); break;           \
    case OP * 64 + 0 * 8 + 2: PK2(MN, 0, 0, 0, 1); break; case OP * 64 + 1 * 8 + 2: PK2(MN, 1, 0, 0, 1); break;           \
    case OP * 64 + 2 * 8 + 2: PK2(MN, 0, 1, 0, 1); break; case OP * 64 + 3 * 8 + 2: PK2(MN, 1, 1, 0, 1); break;           \
    case OP * 64 + 0 * 8 + 1: PK2(MN, 0, 0, 1, 0); break; case OP * 64 + 1 * 8 + 1: PK2(MN, 1, 0, 1, 0); break;           \
    case OP * 64 + 2 * 8 + 1: PK2(MN, 0, 1, 1, 0); break; case OP * 64 + 3 * 8 + 1: PK2(MN, 1, 1, 1, 0); break;           \
    case OP * 64 + 0 * 8 + 0: PK2(MN, 0, 0, 0, 0); break; case OP * 64 + 1 * 8 + 0: PK2(MN, 1, 0, 0, 0); break;           \
    case OP * 64 + 2 * 8 + 0: PK2(MN, 0, 1, 0, 0); break; case OP * 64 + 3 * 8 + 0: PK2(MN, 1, 1, 0, 0); break;
            ROW2(0, "v_pk_add_f32")
            ROW2(1, "v_pk_mul_f32")
            case 3 * 64 + 0 * 8 + 3: asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=&v"(E) : "v"(A), "v"(B)); break;
            case 3 * 64 + 1 * 8 + 3: asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=&v"(E) : "v"(A), "v"(B)); break;
            case 3 * 64 + 2 * 8 + 3: asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=&v"(E) : "v"(A), "v"(B)); break;
            case 3 * 64 + 3 * 8 + 3: asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=&v"(E) : "v"(A), "v"(B)); break;
            case 2 * 64 + 0 * 8 + 7: PK3(0, 0, 0, 1, 1, 1); break;
            case 2 * 64 + 1 * 8 + 7: PK3(1, 0, 0, 1, 1, 1); break;
            case 2 * 64 + 2 * 8 + 7: PK3(0, 1, 0, 1, 1, 1); break;
            case 2 * 64 + 4 * 8 + 7: PK3(0, 0, 1, 1, 1, 1); break;
            case 2 * 64 + 0 * 8 + 6: PK3(0, 0, 0, 0, 1, 1); break;
            case 2 * 64 + 0 * 8 + 5: PK3(0, 0, 0, 1, 0, 1); break;
            case 2 * 64 + 0 * 8 + 3: PK3(0, 0, 0, 1, 1, 0); break;
            default: break;
    }
    return E;
}

template <int form>
__global__ __launch_bounds__(256) void victim(First* res, int iters, volatile int* flags) {
    const uint32_t t = threadIdx.x + blockIdx.x * 256u;
    constexpr int op = form >> 6, sel = (form >> 3) & 7, selhi = form & 7;
    unsigned bad = 0;
    for (int it = 0; it < iters; it += 4) {
      f2 As[4], Bs[4], Cs[4], Es[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t k = t * 8u + (it + u) * 7919u;
        As[u] = f2{val(k), val(k + 1)}; Bs[u] = f2{val(k + 2), val(k + 3)}; Cs[u] = f2{val(k + 4), val(k + 5)};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) Es[u] = one<form>(As[u], Bs[u], Cs[u]);  // four of the instruction back to back
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f2 A = As[u], B = Bs[u], C = Cs[u], E = Es[u];
        // the same selection with scalar instructions: low half takes source i's high half where sel bit i is set, high half takes
        // source i's low half where selhi bit i is clear
        const float al = (sel & 1) ? A[1] : A[0], bl = (sel & 2) ? B[1] : B[0], cl = (sel & 4) ? C[1] : C[0];
        const float ah = (selhi & 1) ? A[1] : A[0], bh = (selhi & 2) ? B[1] : B[0], ch = (selhi & 4) ? C[1] : C[0];
        // (v_pk_mov_b32: D.lo = op_sel[0] ? A.hi : A.lo, D.hi = op_sel[1] ? B.hi : B.lo)
        const float w0 = op == 0 ? al + bl : op == 1 ? al * bl : op == 2 ? __builtin_fmaf(al, bl, cl) : al;
        const float w1 = op == 0 ? ah + bh : op == 1 ? ah * bh : op == 2 ? __builtin_fmaf(ah, bh, ch) : ((sel & 2) ? B[1] : B[0]);
        if (E[0] != w0 || E[1] != w1) {
            if (bad == 0 && atomicAdd(&res->count, 1ull) == 0) {
                res->a[0] = A[0]; res->a[1] = A[1]; res->b[0] = B[0]; res->b[1] = B[1]; res->c[0] = C[0]; res->c[1] = C[1];
                res->got[0] = E[0]; res->got[1] = E[1]; res->want[0] = w0; res->want[1] = w1;
            } else {
                atomicAdd(&res->count, 1ull);
            }
            ++bad;
        }
      }
    }
}

static constexpr int FORMS[] = {
    // v_pk_add_f32 and v_pk_mul_f32: sel 0..3 x selhi 3,2,1,0
    0 * 64 + 0 * 8 + 3, 0 * 64 + 1 * 8 + 3, 0 * 64 + 2 * 8 + 3, 0 * 64 + 3 * 8 + 3, 0 * 64 + 0 * 8 + 2, 0 * 64 + 1 * 8 + 2, 0 * 64 + 2 * 8 + 2, 0 * 64 + 3 * 8 + 2,
    0 * 64 + 0 * 8 + 1, 0 * 64 + 1 * 8 + 1, 0 * 64 + 2 * 8 + 1, 0 * 64 + 3 * 8 + 1, 0 * 64 + 0 * 8 + 0, 0 * 64 + 1 * 8 + 0, 0 * 64 + 2 * 8 + 0, 0 * 64 + 3 * 8 + 0,
    1 * 64 + 0 * 8 + 3, 1 * 64 + 1 * 8 + 3, 1 * 64 + 2 * 8 + 3, 1 * 64 + 3 * 8 + 3, 1 * 64 + 0 * 8 + 2, 1 * 64 + 1 * 8 + 2, 1 * 64 + 2 * 8 + 2, 1 * 64 + 3 * 8 + 2,
    1 * 64 + 0 * 8 + 1, 1 * 64 + 1 * 8 + 1, 1 * 64 + 2 * 8 + 1, 1 * 64 + 3 * 8 + 1, 1 * 64 + 0 * 8 + 0, 1 * 64 + 1 * 8 + 0, 1 * 64 + 2 * 8 + 0, 1 * 64 + 3 * 8 + 0,
    3 * 64 + 0 * 8 + 3, 3 * 64 + 1 * 8 + 3, 3 * 64 + 2 * 8 + 3, 3 * 64 + 3 * 8 + 3,
    2 * 64 + 0 * 8 + 7, 2 * 64 + 1 * 8 + 7, 2 * 64 + 2 * 8 + 7, 2 * 64 + 4 * 8 + 7, 2 * 64 + 0 * 8 + 6, 2 * 64 + 0 * 8 + 5, 2 * 64 + 0 * 8 + 3};

template <int I>
static void launch_form(int form, First* res, int blocks, int iters, int* flags, hipStream_t s) {
    if constexpr (I < (int)(sizeof(FORMS) / sizeof(int))) {
        if (FORMS[I] == form) hipLaunchKernelGGL(victim<FORMS[I]>, dim3(blocks), dim3(256), 0, s, res, iters, flags);
        else launch_form<I + 1>(form, res, blocks, iters, flags, s);
    }
}

template <int KIND>
static void start(float* out, int blocks, int iters, int* flags, hipStream_t s) {
    hipLaunchKernelGGL(disturb<KIND>, dim3(blocks), dim3(256), 0, s, out, iters, flags);
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 1024, iters = argc > 2 ? atoi(argv[2]) : 20000;
    First* d_res;
    float* d_out;
    CK(hipMalloc(&d_res, sizeof(First)));
    CK(hipMalloc(&d_out, 4096));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const char* dn[8] = {"alone", "v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_32x32x16_f16", "v_mfma_f32_16x16x16_f16",
                         "v_mfma_f32_16x16x4_f32", "v_mfma_i32_16x16x64_i8", "v_pk_fma_f32 (VALU only)"};
    const char* on[4] = {"v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_pk_mov_b32"};
    const unsigned long long total = (unsigned long long)blocks * 256ull * iters;
    int* d_flags;
    CK(hipMalloc(&d_flags, 8));
    const int dblocks = argc > 3 ? atoi(argv[3]) : 512;  // two workgroups = two waves per SIMD of the other kernel: room for the victim
    for (int d = 0; d < 8; ++d) {
        int wrong_forms = 0, overlapped = 0;
        for (int form : FORMS) {
            CK(hipMemset(d_flags, 0, 8));
            CK(hipMemset(d_res, 0, sizeof(First)));
            const int di = 1000000;  // ~60-250 ms of matrix instructions per SIMD: longer than the victim (~10 ms)
            switch (d) {
                case 1: start<0>(d_out, dblocks, di, d_flags, sb); break;
                case 2: start<1>(d_out, dblocks, di, d_flags, sb); break;
                case 3: start<2>(d_out, dblocks, di, d_flags, sb); break;
                case 4: start<3>(d_out, dblocks, 2 * di, d_flags, sb); break;
                case 5: start<4>(d_out, dblocks, di / 2, d_flags, sb); break;
                case 6: start<5>(d_out, dblocks, di, d_flags, sb); break;
                case 7: start<6>(d_out, dblocks, 4 * di, d_flags, sb); break;
                default: break;
            }
            if (d > 0) {  // the victim goes out once the other kernel's first workgroup has reported in
                int started = 0;
                for (int spin = 0; spin < 100000 && !started; ++spin) CK(hipMemcpy(&started, d_flags, 4, hipMemcpyDeviceToHost));
            }
            launch_form<0>(form, d_res, blocks, iters, d_flags, sa);
            CK(hipStreamSynchronize(sa));
            const bool other_still_running = hipStreamQuery(sb) == hipErrorNotReady;  // ... and it had started before the victim was queued
            CK(hipStreamSynchronize(sb));
            First r;
            CK(hipMemcpy(&r, d_res, sizeof(First), hipMemcpyDeviceToHost));
            overlapped += d == 0 || other_still_running;
            if (r.count) {
                ++wrong_forms;
                const int op = form >> 6, sel = (form >> 3) & 7, sh = form & 7;
                if (op != 2)
                    printf("beside %-26s %s op_sel:[%d,%d] op_sel_hi:[%d,%d]  wrong %llu of %llu   e.g. A = (%g, %g) B = (%g, %g): got (%g, %g), want (%g, %g)\n",
                           dn[d], on[op], sel & 1, (sel >> 1) & 1, sh & 1, (sh >> 1) & 1, r.count, total, r.a[0], r.a[1], r.b[0], r.b[1], r.got[0], r.got[1],
                           r.want[0], r.want[1]);
                else
                    printf("beside %-26s %s op_sel:[%d,%d,%d] op_sel_hi:[%d,%d,%d]  wrong %llu of %llu   e.g. A = (%g, %g) B = (%g, %g) C = (%g, %g): got (%g, %g), "
                           "want (%g, %g)\n",
                           dn[d], on[op], sel & 1, (sel >> 1) & 1, (sel >> 2) & 1, sh & 1, (sh >> 1) & 1, (sh >> 2) & 1, r.count, total, r.a[0], r.a[1], r.b[0],
                           r.b[1], r.c[0], r.c[1], r.got[0], r.got[1], r.want[0], r.want[1]);
            }
        }
        printf("beside %-26s %d of %zu forms with wrong results; the other kernel was running before the victim was queued and after it had finished in %d of them\n",
               dn[d], wrong_forms, sizeof(FORMS) / sizeof(int), overlapped);
    }
    return 0;
}
