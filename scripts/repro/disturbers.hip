// Synthetic disturber kernels for scripts/overlap_micro.py (DESIGN_LESSONS.md lesson 46): each one exercises ONE feature of the
// fp16-split convolution kernels, so that the co-running feature that corrupts pmn_feature_weight / the PixelwiseNet launch can be named.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libdisturb.so disturbers.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// 0: fp16 MFMA only (registers)
__global__ __launch_bounds__(256) void k_mfma_f16(float* out, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i * 0.01f); }
    f32x4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc1, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[1];
}

// 1: fp32 MFMA only
__global__ __launch_bounds__(256) void k_mfma_f32(float* out, int iters) {
    float a = threadIdx.x * 0.001f, b = 0.5f;
    f32x4_t acc0 = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0];
}

// 2: fp32 -> fp16 conversions (v_cvt_pk_f16_f32 and back), VALU only
__global__ __launch_bounds__(256) void k_cvt(float* out, int iters) {
    f32x2_t x = {threadIdx.x * 0.37f, threadIdx.x * 0.11f + 1.0f};
    f32x2_t s = {0, 0};
    for (int it = 0; it < iters; ++it) {
        const f16x2_t h = __builtin_convertvector(x, f16x2_t);
        const f32x2_t back = __builtin_convertvector(h, f32x2_t);
        s += (x - back) * 2048.0f;
        x += 0.001f;
    }
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1];
}

// 3: LDS traffic: ds_write_b64 of fp16 quads, barrier, ds_read_b128 -- `lds_bytes` of dynamic LDS
__global__ __launch_bounds__(256) void k_lds(float* out, int iters, int lds_bytes) {
    extern __shared__ float4 smem[];
    _Float16* p = reinterpret_cast<_Float16*>(smem);
    const int nq = lds_bytes / 8;  // 8-byte slots
    float s = 0.f;
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < nq; i += 256) {
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            const f16x4 v = {(_Float16)(i & 255), (_Float16)it, (_Float16)1, (_Float16)2};
            *reinterpret_cast<f16x4*>(p + 4 * i) = v;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nq / 2; i += 256) {
            const f16x8 v = *reinterpret_cast<const f16x8*>(p + 8 * i);
            s += (float)v[0] + (float)v[7];
        }
        __syncthreads();
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// 4: LDS + fp16 MFMA fed from LDS (the convolution's k loop without its global loads)
__global__ __launch_bounds__(256) void k_lds_mfma(float* out, int iters, int lds_bytes) {
    extern __shared__ float4 smem[];
    _Float16* p = reinterpret_cast<_Float16*>(smem);
    const int nh = lds_bytes / 2;
    for (int i = threadIdx.x; i < nh; i += 256) p[i] = (_Float16)((i & 63) * 0.01f);
    __syncthreads();
    f32x4_t acc = {0, 0, 0, 0};
    f16x8 b;
    for (int i = 0; i < 8; ++i) b[i] = (_Float16)(0.25f + i);
    const int n16 = lds_bytes / 16;
    for (int it = 0; it < iters; ++it) {
        const f16x8 a = *reinterpret_cast<const f16x8*>(p + 8 * ((threadIdx.x * 7 + it * 13) % n16));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0];
}

// 5: global float4 loads + stores only (streaming copy)
__global__ __launch_bounds__(256) void k_copy(const float4* in, float4* out, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

extern "C" int disturb_launch(int which, float* buf, size_t buf_floats, int blocks, int iters, int lds_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (which) {
        case 0: hipLaunchKernelGGL(k_mfma_f16, dim3(blocks), dim3(256), 0, st, buf, iters); break;
        case 1: hipLaunchKernelGGL(k_mfma_f32, dim3(blocks), dim3(256), 0, st, buf, iters); break;
        case 2: hipLaunchKernelGGL(k_cvt, dim3(blocks), dim3(256), 0, st, buf, iters); break;
        case 3: hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), lds_bytes, st, buf, iters, lds_bytes); break;
        case 4: hipLaunchKernelGGL(k_lds_mfma, dim3(blocks), dim3(256), lds_bytes, st, buf, iters, lds_bytes); break;
        case 5: hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const float4*>(buf),
                                   reinterpret_cast<float4*>(buf + buf_floats / 2), buf_floats / 8); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
