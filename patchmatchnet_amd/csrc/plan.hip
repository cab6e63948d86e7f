// Launch plans: a forward recorded once, replayed from C with plain kernel launches (include/pmn_hip.h: pmn_plan_*).
//
// Why this exists.  One PatchmatchNet forward is ~55 kernel launches.  Issued from Python they cost about as much interpreter time as
// the kernels take to run, which is what stops several samples from being in flight; rounds 2-5 captured the forward into a HIP graph
// and replayed that.  A plan keeps what the graph bought -- one call per forward, no interpreter between the launches -- without the
// graph: pmn_plan_launch is a loop of hipLaunchKernel calls on the caller's stream, exactly what the entry points themselves do, so it
// depends on nothing but kernel launches (no capture modes, no graph memory pools, no runtime-version-specific replay path) and its
// contents can be listed.  Same replay rate as the graph (bench.py --launch graph | plan).
//
// Recording is per THREAD: between pmn_plan_begin and pmn_plan_end every PMN_LAUNCH of the calling thread (pmn_common.hpp) appends
// {kernel symbol, grid, block, dynamic LDS, argument bytes} to the plan instead of launching; other threads keep launching.  Kernel
// arguments are passed by value (pointers, sizes, small structs with the neighbour tables inside), so a plan is self-contained: it
// holds no reference to the caller's host memory, only the DEVICE addresses that were passed while recording -- the caller keeps
// those buffers alive and unmoved for the life of the plan (patchmatchnet_amd/graph.py records under a private torch memory pool).
//
// Nothing here synchronises, allocates device memory or copies: begin / end / launch only touch host memory and enqueue.
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "pmn_common.hpp"

struct PmnPlanEntry {
    const void* func;
    dim3 grid, block;
    size_t lds;
    int nargs;
    size_t first_arg;  // index into PmnPlan::arg_offsets
};

struct PmnPlan {
    uint32_t magic;
    bool recording, poisoned, sealed;
    int device;  // the device that was current while recording (per-device kernel attributes were raised there)
    std::vector<PmnPlanEntry> entries;
    std::vector<size_t> arg_offsets;  // byte offset of every argument in `blob`
    std::vector<unsigned char> blob;  // the arguments' bytes, each at its natural alignment
    std::vector<void*> argv;          // built by pmn_plan_end: pointers into `blob`, entry e's start at entries[e].first_arg
};

static const uint32_t PMN_PLAN_MAGIC = 0x504d4e50u;  // "PMNP"

thread_local PmnPlan* pmn_tls_plan = nullptr;

static PmnPlan* as_plan(void* p) {
    PmnPlan* plan = static_cast<PmnPlan*>(p);
    return (plan && plan->magic == PMN_PLAN_MAGIC) ? plan : nullptr;
}

int pmn_plan_append(PmnPlan* plan, const void* func, dim3 grid, dim3 block, size_t lds, int nargs, void* const* args,
                    const size_t* sizes, const size_t* aligns) {
    if (!plan || plan->poisoned) return PMN_ERR_ARG;
    try {
        PmnPlanEntry e{func, grid, block, lds, nargs, plan->arg_offsets.size()};
        for (int i = 0; i < nargs; ++i) {
            const size_t al = aligns[i] ? aligns[i] : 1;
            size_t off = (plan->blob.size() + al - 1) / al * al;
            plan->blob.resize(off + sizes[i]);
            std::memcpy(plan->blob.data() + off, args[i], sizes[i]);
            plan->arg_offsets.push_back(off);
        }
        plan->entries.push_back(e);
    } catch (const std::bad_alloc&) {
        plan->poisoned = true;
        return PMN_ERR_ARG;
    }
    return PMN_OK;
}

extern "C" int pmn_plan_create(void** plan_out) {
    if (!plan_out) return PMN_ERR_ARG;
    PmnPlan* plan = new (std::nothrow) PmnPlan();
    if (!plan) return PMN_ERR_ARG;
    plan->magic = PMN_PLAN_MAGIC;
    plan->recording = plan->poisoned = plan->sealed = false;
    plan->device = -1;
    *plan_out = plan;
    return PMN_OK;
}

extern "C" int pmn_plan_begin(void* p) {
    PmnPlan* plan = as_plan(p);
    if (!plan || plan->sealed || plan->recording || pmn_tls_plan != nullptr) return PMN_ERR_ARG;  // one recording per thread, once per plan
    if (hipGetDevice(&plan->device) != hipSuccess) plan->device = -1;  // (no device: recording itself never touches one)
    plan->recording = true;
    pmn_tls_plan = plan;
    return PMN_OK;
}

extern "C" int pmn_plan_end(void* p) {
    PmnPlan* plan = as_plan(p);
    if (!plan || !plan->recording || pmn_tls_plan != plan) return PMN_ERR_ARG;
    pmn_tls_plan = nullptr;
    plan->recording = false;
    if (plan->poisoned) return PMN_ERR_ARG;
    // the blob no longer moves: resolve the argument pointers once (16-byte alignment of the blob's base: std::vector<unsigned char>
    // allocates through operator new, which aligns to max_align_t; the largest kernel parameter alignment in the library is 8)
    try {
        plan->argv.resize(plan->arg_offsets.size());
    } catch (const std::bad_alloc&) {
        plan->poisoned = true;
        return PMN_ERR_ARG;
    }
    for (size_t i = 0; i < plan->arg_offsets.size(); ++i) plan->argv[i] = plan->blob.data() + plan->arg_offsets[i];
    plan->sealed = true;
    return PMN_OK;
}

extern "C" int pmn_plan_count(const void* p) {
    const PmnPlan* plan = as_plan(const_cast<void*>(p));
    if (!plan) return PMN_ERR_ARG;
    return (int)plan->entries.size();
}

extern "C" const char* pmn_plan_kernel_name(const void* p, int index) {
    const PmnPlan* plan = as_plan(const_cast<void*>(p));
    if (!plan || index < 0 || index >= (int)plan->entries.size()) return nullptr;
    return hipKernelNameRefByPtr(plan->entries[index].func, nullptr);
}

extern "C" int pmn_plan_launch(const void* p, void* stream) {
    const PmnPlan* plan = as_plan(const_cast<void*>(p));
    if (!plan || !plan->sealed) return PMN_ERR_ARG;
    if (pmn_tls_plan != nullptr) return PMN_ERR_ARG;  // a plan is not recorded into a plan
    hipStream_t st = (hipStream_t)stream;
    void* const* argv = plan->argv.data();
    for (const PmnPlanEntry& e : plan->entries) {
        if (hipLaunchKernel(e.func, e.grid, e.block, const_cast<void**>(argv + e.first_arg), e.lds, st) != hipSuccess) {
            (void)hipGetLastError();
            return PMN_ERR_LAUNCH;
        }
    }
    return PMN_OK;
}

extern "C" int pmn_plan_destroy(void* p) {
    PmnPlan* plan = as_plan(p);
    if (!plan) return PMN_ERR_ARG;
    if (pmn_tls_plan == plan) pmn_tls_plan = nullptr;
    plan->magic = 0;
    delete plan;
    return PMN_OK;
}
