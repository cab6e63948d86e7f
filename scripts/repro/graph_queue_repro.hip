// Torch-free reproducer for DESIGN_LESSONS.md lesson 45: does a chain of producer -> consumer kernels, captured into a HIP graph and
// replayed on stream A, still give the eager chain's bytes while other streams of the process replay THEIR graphs on other
// hardware queues?  No PyTorch, no allocator, no library: hipMalloc'd buffers, three slots, each with its own stream, buffers and
// graph; every replay is preceded by a small pinned host -> device copy on the same stream (the per-sample address table / cameras
// of patchmatchnet_amd/graph.py: _fill) that the chain's first kernel reads.
//
//   graph_queue_repro <mode> [slots] [rounds] [MB] [chain]
//     mode: graph   one hipGraphLaunch per slot and round (stream capture of the chain)
//           eager   the same kernels launched one by one
//           manual  the graph built with hipGraphAddKernelNode (explicit linear dependencies) instead of stream capture
//   prints, per slot, how many rounds produced a final buffer that differs from the expected one (computed by an eager,
//   single-stream run of the same chain with the same seed) and the first differing element.
//
// hipcc --offload-arch=gfx950 -O2 -o graph_queue_repro graph_queue_repro.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                           \
        }                                                                                      \
    } while (0)

// out[i] = mix(in[(i * stride + seed) % n], i): every workgroup reads lines written by far-away workgroups of the previous kernel
// (other XCDs, other L2s), like a gather over a feature map the previous kernel produced.
__global__ void stage_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, const uint32_t* __restrict__ seed_dev,
                             uint32_t n, uint32_t stride, uint32_t salt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t seed = seed_dev[0];
    const uint64_t j = ((uint64_t)i * stride + seed + salt) % n;
    uint32_t v = in[j] ^ (i * 2654435761u);
    v = (v << 7) | (v >> 25);
    out[i] = v + seed + salt;
}

__global__ void first_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ seed_dev, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i * 40503u + seed_dev[0] * 97u;
}

struct Slot {
    hipStream_t stream;
    uint32_t *a, *b, *seed_dev, *seed_host, *result_host;
    hipGraph_t graph;
    hipGraphExec_t exec;
};

static void launch_chain(const Slot& s, uint32_t n, int chain, hipStream_t st) {
    const dim3 block(256), grid((n + 255) / 256);
    hipLaunchKernelGGL(first_kernel, grid, block, 0, st, s.a, s.seed_dev, n);
    uint32_t *in = s.a, *out = s.b;
    for (int k = 0; k < chain; ++k) {
        hipLaunchKernelGGL(stage_kernel, grid, block, 0, st, in, out, s.seed_dev, n, 1000003u + 2u * k, (uint32_t)k);
        uint32_t* t = in;
        in = out;
        out = t;
    }
}

static uint32_t* final_buffer(const Slot& s, int chain) { return (chain % 2) ? s.b : s.a; }

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "graph";
    const int slots = argc > 2 ? atoi(argv[2]) : 3;
    const int rounds = argc > 3 ? atoi(argv[3]) : 100;
    const size_t mb = argc > 4 ? atoi(argv[4]) : 32;
    const int chain = argc > 5 ? atoi(argv[5]) : 12;
    const uint32_t n = (uint32_t)(mb * 1024 * 1024 / 4);
    const bool use_graph = strcmp(mode, "eager") != 0, manual = strcmp(mode, "manual") == 0;
    std::vector<Slot> S(slots);
    for (Slot& s : S) {
        CK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        CK(hipMalloc(&s.a, (size_t)n * 4));
        CK(hipMalloc(&s.b, (size_t)n * 4));
        CK(hipMalloc(&s.seed_dev, 64));
        CK(hipHostMalloc(&s.seed_host, 64 * rounds));
        CK(hipHostMalloc(&s.result_host, (size_t)n * 4));
        CK(hipMemset(s.seed_dev, 0, 64));
    }
    CK(hipDeviceSynchronize());
    if (use_graph) {
        for (Slot& s : S) {
            if (!manual) {
                CK(hipStreamBeginCapture(s.stream, hipStreamCaptureModeThreadLocal));
                launch_chain(s, n, chain, s.stream);
                CK(hipStreamEndCapture(s.stream, &s.graph));
            } else {
                CK(hipGraphCreate(&s.graph, 0));
                const dim3 block(256), grid((n + 255) / 256);
                hipGraphNode_t prev;
                {
                    uint32_t nn = n;
                    void* args[] = {&s.a, &s.seed_dev, &nn};
                    hipKernelNodeParams p{};
                    p.func = (void*)first_kernel;
                    p.gridDim = grid;
                    p.blockDim = block;
                    p.kernelParams = args;
                    CK(hipGraphAddKernelNode(&prev, s.graph, nullptr, 0, &p));
                }
                uint32_t *in = s.a, *out = s.b;
                for (int k = 0; k < chain; ++k) {
                    uint32_t nn = n, stride = 1000003u + 2u * k, salt = (uint32_t)k;
                    void* args[] = {&in, &out, &s.seed_dev, &nn, &stride, &salt};
                    hipKernelNodeParams p{};
                    p.func = (void*)stage_kernel;
                    p.gridDim = grid;
                    p.blockDim = block;
                    p.kernelParams = args;
                    hipGraphNode_t node;
                    CK(hipGraphAddKernelNode(&node, s.graph, &prev, 1, &p));
                    prev = node;
                    uint32_t* t = in;
                    in = out;
                    out = t;
                }
            }
            CK(hipGraphInstantiate(&s.exec, s.graph, nullptr, nullptr, 0));
        }
    }
    // expected results per (slot, round): an eager, fully synchronised run with the same seeds
    std::vector<std::vector<uint64_t>> want(slots, std::vector<uint64_t>(rounds));
    std::vector<uint32_t> host(n);
    auto checksum = [&](const uint32_t* p) {
        uint64_t h = 1469598103934665603ull;
        for (uint32_t i = 0; i < n; ++i) h = (h ^ p[i]) * 1099511628211ull;
        return h;
    };
    const int distinct = rounds < 4 ? rounds : 4;  // seeds cycle: four reference results per slot are enough
    for (int k = 0; k < slots; ++k)
        for (int r = 0; r < distinct; ++r) {
            Slot& s = S[k];
            s.seed_host[16 * r] = 12345u + 1000u * k + 7u * (r % distinct);
            CK(hipMemcpy(s.seed_dev, &s.seed_host[16 * r], 4, hipMemcpyHostToDevice));
            launch_chain(s, n, chain, s.stream);
            CK(hipStreamSynchronize(s.stream));
            CK(hipMemcpy(host.data(), final_buffer(s, chain), (size_t)n * 4, hipMemcpyDeviceToHost));
            want[k][r] = checksum(host.data());
        }
    // the overlapped run: slots interleaved, nothing between the launches but the small upload
    std::vector<int> bad(slots, 0);
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    float total_ms = 0.0f;
    for (int r = 0; r < rounds; ++r) {
        for (int k = 0; k < slots; ++k) {
            Slot& s = S[k];
            s.seed_host[16 * r] = 12345u + 1000u * k + 7u * (r % distinct);
            CK(hipMemcpyAsync(s.seed_dev, &s.seed_host[16 * r], 4, hipMemcpyHostToDevice, s.stream));
            if (use_graph) CK(hipGraphLaunch(s.exec, s.stream));
            else launch_chain(s, n, chain, s.stream);
            CK(hipMemcpyAsync(s.result_host, final_buffer(s, chain), (size_t)n * 4, hipMemcpyDeviceToHost, s.stream));
        }
        for (int k = 0; k < slots; ++k) {
            CK(hipStreamSynchronize(S[k].stream));
            if (checksum(S[k].result_host) != want[k][r % distinct]) ++bad[k];
        }
    }
    (void)total_ms;
    int total = 0;
    for (int k = 0; k < slots; ++k) total += bad[k];
    printf("mode=%s slots=%d rounds=%d MB=%zu chain=%d GPU_MAX_HW_QUEUES=%s HSA_ENABLE_SDMA=%s : rounds that differ per slot =", mode, slots,
           rounds, mb, chain, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "unset",
           getenv("HSA_ENABLE_SDMA") ? getenv("HSA_ENABLE_SDMA") : "unset");
    for (int k = 0; k < slots; ++k) printf(" %d", bad[k]);
    printf("  total=%d\n", total);
    return total ? 1 : 0;
}
