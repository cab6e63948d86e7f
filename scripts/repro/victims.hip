// Self-checking victim micro-kernels for DESIGN_LESSONS.md lesson 46: each exercises ONE instruction class in a loop and counts, on
// the device, how often the result is not the arithmetically known value.  Run alone they must count 0; scripts/repro/micro_matrix.py
// runs them beside the register-only v_mfma_f32_16x16x32_f16 loop of disturbers.hip.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libvictims.so victims.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t mix(uint32_t a) {
    a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16;
    return a;
}

// 0: LDS b32 -- lane writes a word, reads the word of lane+1 (wave-private rows)
__global__ __launch_bounds__(256) void v_lds_b32(unsigned long long* errs, int iters) {
    __shared__ uint32_t s[256];
    const int t = threadIdx.x, peer = (t & ~63) | ((t + 1) & 63);
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        s[t] = mix(it * 977u + t + blockIdx.x * 131u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t got = s[peer];
        bad += got != mix(it * 977u + peer + blockIdx.x * 131u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (bad) atomicAdd(errs, (unsigned long long)bad);
}

// 1: LDS b128 -- lane writes 16 bytes, reads the 16 bytes of lane+1
__global__ __launch_bounds__(256) void v_lds_b128(unsigned long long* errs, int iters) {
    __shared__ uint4 s[256];
    const int t = threadIdx.x, peer = (t & ~63) | ((t + 1) & 63);
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t b = mix(it * 977u + t + blockIdx.x * 131u);
        s[t] = make_uint4(b, b + 1, b + 2, b + 3);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint4 got = s[peer];
        const uint32_t e = mix(it * 977u + peer + blockIdx.x * 131u);
        bad += (got.x != e) | (got.y != e + 1) | (got.z != e + 2) | (got.w != e + 3);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (bad) atomicAdd(errs, (unsigned long long)bad);
}

// 2: LDS records the way MODE_NEIGHBOR uses them: phase A writes [float4 + int] records, workgroup barrier, every lane group of 8 lanes
//    reads ITS pixel's records (broadcast inside the group) one after the other
__global__ __launch_bounds__(256) void v_lds_records(unsigned long long* errs, int iters) {
    __shared__ uint4 rw[32 * 9];
    __shared__ uint32_t ro[32 * 9];
    const int t = threadIdx.x, grp = t / 8;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        for (int i = t; i < 32 * 9; i += 256) {
            const uint32_t b = mix(it * 7919u + i + blockIdx.x * 131u);
            rw[i] = make_uint4(b, b ^ 1, b ^ 2, b ^ 3);
            ro[i] = b + 7;
        }
        __syncthreads();
        for (int d = 0; d < 9; ++d) {
            const int i = d * 32 + grp;
            const uint4 w = rw[i];
            const uint32_t o = ro[i];
            const uint32_t e = mix(it * 7919u + i + blockIdx.x * 131u);
            bad += (w.x != e) | (w.y != (e ^ 1)) | (w.z != (e ^ 2)) | (w.w != (e ^ 3)) | (o != e + 7);
        }
        __syncthreads();
    }
    if (bad) atomicAdd(errs, (unsigned long long)bad);
}

// 3: global gather: 4 x global_load_dwordx4 at data-dependent offsets of a buffer with buf[i] = mix(i)
__global__ __launch_bounds__(256) void v_gather(unsigned long long* errs, int iters, const uint4* buf, uint32_t n16) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t i0 = mix(t * 31u + it) % (n16 - 600);
        const uint4 a = buf[i0], b = buf[i0 + 1], c = buf[i0 + 512], d = buf[i0 + 513];
        bad += (a.x != mix(4 * i0)) | (a.w != mix(4 * i0 + 3)) | (b.y != mix(4 * (i0 + 1) + 1)) | (c.z != mix(4 * (i0 + 512) + 2)) |
               (d.x != mix(4 * (i0 + 513)));
    }
    if (bad) atomicAdd(errs, (unsigned long long)bad);
}

// 4: DPP row_newbcast (the record broadcast of MODE_VIEWS / the PixelwiseNet launch)
__global__ __launch_bounds__(256) void v_dpp(unsigned long long* errs, int iters) {
    const int t = threadIdx.x, lane = t & 63;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        const int v = (int)mix(it * 613u + t);
        const int got5 = __builtin_amdgcn_mov_dpp(v, 0x150 + 5, 0xF, 0xF, true);
        const int got11 = __builtin_amdgcn_mov_dpp(v, 0x150 + 11, 0xF, 0xF, true);
        const int e5 = (int)mix(it * 613u + ((t & ~15) | 5)), e11 = (int)mix(it * 613u + ((t & ~15) | 11));
        bad += (got5 != e5) | (got11 != e11);
        (void)lane;
    }
    if (bad) atomicAdd(errs, (unsigned long long)bad);
}

// 5: packed fp32 math against the scalar form
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void v_pk(unsigned long long* errs, int iters) {
    const int t = threadIdx.x;
    unsigned bad = 0;
    float x = 1.0f + t * 0.001f;
    for (int it = 0; it < iters; ++it) {
        const f2 a = {x, x * 0.5f}, b = {0.75f + it * 1e-4f, 1.25f}, c = {0.1f, -0.2f};
        const f2 r = __builtin_elementwise_fma(a, b, c);
        float r0, r1;
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(a.x), "v"(b.x), "v"(c.x));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r1) : "v"(a.y), "v"(b.y), "v"(c.y));
        bad += (r.x != r0) | (r.y != r1);
        x = x * 1.0001f + 1e-3f;
        if (x > 100.f) x = 1.0f;
    }
    if (bad) atomicAdd(errs, (unsigned long long)bad);
}

// 6: LDS b64 / b128 rows the way the PixelwiseNet launch uses them: owner lanes write float4, after a wave barrier the item role
//    reads float2 from another lane's row
__global__ __launch_bounds__(256) void v_lds_rows(unsigned long long* errs, int iters) {
    __shared__ uint4 s[4][64];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t b = mix(it * 977u + t + blockIdx.x * 131u);
        if ((lane & 1) == 0) s[wave][lane] = make_uint4(b, b + 1, b + 2, b + 3);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int src = (lane * 2) & 62;  // an even lane's row
        const uint2 got = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint32_t*>(&s[wave][src]) + 2 * (lane >> 5));
        const uint32_t e = mix(it * 977u + ((t & ~63) | src) + blockIdx.x * 131u) + 2 * (lane >> 5);
        bad += (got.x != e) | (got.y != e + 1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (bad) atomicAdd(errs, (unsigned long long)bad);
}

// 7: expf + IEEE division (the sigmoid) against values computed by a loop-free reference path earlier in the same thread
__global__ __launch_bounds__(256) void v_sigmoid(unsigned long long* errs, int iters) {
    const int t = threadIdx.x;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        const float x = -4.0f + ((it * 37 + t) & 1023) * (8.0f / 1024.0f);
        const float a = 1.0f / (1.0f + expf(-x));
        float xx = x;
        asm volatile("" : "+v"(xx));  // a second, separately computed copy
        const float b = 1.0f / (1.0f + expf(-xx));
        bad += __float_as_uint(a) != __float_as_uint(b);
    }
    if (bad) atomicAdd(errs, (unsigned long long)bad);
}

// 8: SHORT-LIVED workgroups (what the real victims are: thousands of workgroups that each make one pass): write LDS, one workgroup
//    barrier, read what another WAVE wrote, a second round, exit.  `iters` rounds per workgroup (1-2 = short-lived).
__global__ __launch_bounds__(256, 5) void v_short_wg(unsigned long long* errs, int iters) {
    extern __shared__ uint32_t dyn[];
    const int t = threadIdx.x, peer = (t + 64) & 255;  // the same lane of the next wave
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        dyn[t] = mix(it * 977u + t + blockIdx.x * 131u);
        dyn[256 + 9 * t] = mix(it * 31u + t + blockIdx.x * 7u);  // a second, scattered array further up
        __syncthreads();
        bad += dyn[peer] != mix(it * 977u + peer + blockIdx.x * 131u);
        bad += dyn[256 + 9 * peer] != mix(it * 31u + peer + blockIdx.x * 7u);
        __syncthreads();
    }
    if (bad) atomicAdd(errs, (unsigned long long)bad);
}

// 9 / 10: an LDS read RETURNING WHILE GLOBAL LOADS ARE IN FLIGHT (the FeatureWeightNet launch's inner loop): phase A parks 9 records per
//    pixel in LDS; per record the lane group issues four global_load_dwordx4 and THEN the ds_read_b128 of the record (order 9), or the
//    ds_read_b128 first and the loads after its data has arrived (order 10); the record is checked against its known value and the
//    loaded data is consumed so that the loads stay.
template <bool LDS_FIRST>
__global__ __launch_bounds__(256, 5) void v_lds_under_vmem(unsigned long long* errs, int iters, const uint4* buf, uint32_t n16) {
    __shared__ uint4 rw[32 * 9];
    const int t = threadIdx.x, grp = t / 8, lc = t & 7;
    unsigned bad = 0;
    uint32_t sink = 0;
    for (int it = 0; it < iters; ++it) {
        for (int i = t; i < 32 * 9; i += 256) {
            const uint32_t b = mix(it * 7919u + i + blockIdx.x * 131u);
            rw[i] = make_uint4(b, b ^ 0x1111u, b ^ 0x2222u, b ^ 0x3333u);
        }
        __syncthreads();
        for (int d = 0; d < 9; ++d) {
            const int i = d * 32 + grp;
            const uint32_t i0 = (mix(blockIdx.x * 977u + it * 31u + i) % (n16 - 1100)) + lc;
            uint4 w, a, b, c, e4;
            if (LDS_FIRST) {
                w = rw[i];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                a = buf[i0]; b = buf[i0 + 8]; c = buf[i0 + 1024]; e4 = buf[i0 + 1032];
            } else {
                a = buf[i0]; b = buf[i0 + 8]; c = buf[i0 + 1024]; e4 = buf[i0 + 1032];
                asm volatile("" ::: "memory");
                w = rw[i];
            }
            const uint32_t e = mix(it * 7919u + i + blockIdx.x * 131u);
            bad += (w.x != e) | (w.y != (e ^ 0x1111u)) | (w.z != (e ^ 0x2222u)) | (w.w != (e ^ 0x3333u));
            sink += a.x + b.y + c.z + e4.w;
            bad += (a.x != mix(4 * i0)) | (e4.w != mix(4 * (i0 + 1032) + 3));
        }
        __syncthreads();
    }
    if (bad | (sink == 0x12345678u)) atomicAdd(errs, (unsigned long long)bad);
}

extern "C" int victim_launch(int which, unsigned long long* errs, int blocks, int iters, const void* buf, unsigned n16, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (which) {
        case 0: hipLaunchKernelGGL(v_lds_b32, dim3(blocks), dim3(256), 0, st, errs, iters); break;
        case 1: hipLaunchKernelGGL(v_lds_b128, dim3(blocks), dim3(256), 0, st, errs, iters); break;
        case 2: hipLaunchKernelGGL(v_lds_records, dim3(blocks), dim3(256), 0, st, errs, iters); break;
        case 3: hipLaunchKernelGGL(v_gather, dim3(blocks), dim3(256), 0, st, errs, iters, reinterpret_cast<const uint4*>(buf), n16); break;
        case 4: hipLaunchKernelGGL(v_dpp, dim3(blocks), dim3(256), 0, st, errs, iters); break;
        case 5: hipLaunchKernelGGL(v_pk, dim3(blocks), dim3(256), 0, st, errs, iters); break;
        case 6: hipLaunchKernelGGL(v_lds_rows, dim3(blocks), dim3(256), 0, st, errs, iters); break;
        case 7: hipLaunchKernelGGL(v_sigmoid, dim3(blocks), dim3(256), 0, st, errs, iters); break;
        case 9: hipLaunchKernelGGL(v_lds_under_vmem<false>, dim3(blocks), dim3(256), 0, st, errs, iters, reinterpret_cast<const uint4*>(buf), n16); break;
        case 10: hipLaunchKernelGGL(v_lds_under_vmem<true>, dim3(blocks), dim3(256), 0, st, errs, iters, reinterpret_cast<const uint4*>(buf), n16); break;
        case 8: hipLaunchKernelGGL(v_short_wg, dim3(blocks * 64), dim3(256), 18432, st, errs, iters); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

__global__ void fill_mix(uint32_t* p, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = mix((uint32_t)i);
}
extern "C" int victim_fill(void* buf, size_t n_words, void* stream) {
    hipLaunchKernelGGL(fill_mix, dim3(4096), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<uint32_t*>(buf), n_words);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
