// lds_poison.hip -- test aid (scripts/lds_poison_probe.py): workgroups that fill their whole LDS allocation with a bit pattern and
// exit, so that whatever runs on that CU next finds the pattern in any LDS word it reads without having written it.
#include <hip/hip_runtime.h>
__global__ void lds_poison_kernel(unsigned pattern, int words, unsigned* sink) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = pattern;
    __syncthreads();
    if (sink && lds[(threadIdx.x * 97) % words] != pattern) sink[0] = 1;  // keeps the stores alive
}
extern "C" int lds_poison(int blocks, int bytes, unsigned pattern, unsigned* sink, void* stream) {
    static bool raised = false;
    if (!raised) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(lds_poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        raised = true;
    }
    hipLaunchKernelGGL(lds_poison_kernel, dim3(blocks), dim3(256), bytes, (hipStream_t)stream, pattern, bytes / 4, sink);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
