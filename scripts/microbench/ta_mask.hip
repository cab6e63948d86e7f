// Microbenchmark (development aid): does the vector-memory pipe (TA / L1) charge a dwordx4 wave load by ACTIVE lanes?
// Every 8-lane group reads 128-byte "texels" of an L2-resident map at a slowly advancing position (the gather kernel's access
// shape at C = 32); variant k keeps only groups with (group + step) % k == 0 ... active for each load instruction.
//   hipcc --offload-arch=gfx950 -O3 -o ta_mask ta_mask.hip && ./ta_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KEEP_NUM, int KEEP_DEN>  // active fraction of lane groups per load = KEEP_NUM / KEEP_DEN
__global__ __launch_bounds__(256) void k(const float4* __restrict__ map, int texels, int steps, float* out) {
    const int lane8 = threadIdx.x & 7, grp = (blockIdx.x * 256 + threadIdx.x) >> 3;
    unsigned pos = (unsigned)grp * 3u;
    float4 acc = make_float4(0, 0, 0, 0);
    constexpr int NC = 4;  // independent chains: NC masked loads in flight per wave
    float4 cur[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) cur[c] = acc;
    for (int s = 0; s < steps; s += NC) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            pos += 1 + ((grp + s + c) & 1);
            const unsigned t = pos & (unsigned)(texels - 1);
            const bool on = ((unsigned)(grp + s + c) % KEEP_DEN) < KEEP_NUM;
            if (on) cur[c] = map[(size_t)t * 8 + lane8];
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) { acc.x += cur[c].x; acc.y += cur[c].y; acc.z += cur[c].z; acc.w += cur[c].w; }
    }
    if (acc.x == 123.456f) out[grp] = acc.x + acc.y + acc.z + acc.w;
}

template <int U>  // distinct texels read by the 8 lane groups of a wave per load instruction
__global__ __launch_bounds__(256) void kdup(const float4* __restrict__ map, int texels, int steps, float* out) {
    const int lane8 = threadIdx.x & 7, grp = (blockIdx.x * 256 + threadIdx.x) >> 3;
    const int wave = grp >> 3, gw = grp & 7;
    unsigned pos = (unsigned)wave * 29u;
    float4 acc = make_float4(0, 0, 0, 0);
    constexpr int NC = 4;
    float4 cur[NC];
    for (int s = 0; s < steps; s += NC) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            pos += 9;
            const unsigned t = (pos + (unsigned)(gw % U)) & (unsigned)(texels - 1);
            cur[c] = map[(size_t)t * 8 + lane8];
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) { acc.x += cur[c].x; acc.y += cur[c].y; acc.z += cur[c].z; acc.w += cur[c].w; }
    }
    if (acc.x == 123.456f) out[grp] = acc.x + acc.y + acc.z + acc.w;
}

template <int U>
static void rundup(const float4* map, int texels, float* out) {
    const int steps = 512, blocks = 256 * 16;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((kdup<U>), dim3(blocks), dim3(256), 0, 0, map, texels, steps, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r && ms < best) best = ms;
    }
    const double wave_loads = (double)blocks * 4 * steps;
    printf("%d distinct texels per wave load: %8.3f ms   %6.2f clk/wave-load/CU\n", U, best, best * 1e-3 * 2.4e9 / (wave_loads / 256));
}

template <int A, int B>
static void run(const float4* map, int texels, float* out, const char* name) {
    const int steps = 512, blocks = 256 * 16;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<A, B>), dim3(blocks), dim3(256), 0, 0, map, texels, steps, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r && ms < best) best = ms;
    }
    const double wave_loads = (double)blocks * 4 * steps, active_bytes = wave_loads * 1024.0 * A / B;
    printf("%-14s %8.3f ms   %6.2f clk/wave-load/CU   active bytes %7.1f GB/s  (%5.1f B/clk/CU)\n", name, best,
           best * 1e-3 * 2.4e9 / (wave_loads / 256), active_bytes / best * 1e-6, active_bytes / (best * 1e-3 * 2.4e9 * 256));
}

int main() {
    const int texels = 1 << 17;  // 16 MB: stays in L2 / MALL
    float4* map; float* out;
    hipMalloc(&map, (size_t)texels * 128);
    hipMemset(map, 0, (size_t)texels * 128);
    hipMalloc(&out, 1 << 24);
    run<1, 1>(map, texels, out, "all lanes");
    run<3, 4>(map, texels, out, "3/4 groups");
    run<1, 2>(map, texels, out, "1/2 groups");
    run<1, 4>(map, texels, out, "1/4 groups");
    run<1, 8>(map, texels, out, "1/8 groups");
    rundup<8>(map, texels, out); rundup<4>(map, texels, out); rundup<2>(map, texels, out); rundup<1>(map, texels, out);
    return 0;
}
