// Grid-barrier probe for gfx950: what does one device-wide barrier cost inside a persistent kernel, and which
// fence / cache-policy combination makes data written by one workgroup visible to workgroups on other XCDs
// (each XCD has its own L2)?  Informs the persistent decode kernel (DESIGN.md "decode loop").
//
// Every barrier has its own counter (zeroed before the launch), so there is no generation logic, and every spin
// is bounded: a barrier that does not complete within SPIN_LIMIT polls records a timeout and falls through, so
// the probe can never hang the GPU.
//
// Between barriers k and k+1 every workgroup w publishes a value derived from (k, w) in `buf`, and after the
// barrier checks the values of 8 other workgroups.  mode selects how:
//   0  plain stores/loads, no fences: lower bound for the barrier itself (expected to show stale reads)
//   1  plain stores/loads, __threadfence() (agent-scope release/acquire) around the barrier
//   2  nontemporal (streaming) stores and loads, no fences
//   3  agent-scope relaxed atomic stores and loads, no fences
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr int SPIN_LIMIT = 1 << 18;

// `dead`: a barrier already timed out in this workgroup; stop waiting (the run is void, just get out quickly)
__device__ __forceinline__ bool grid_barrier(int* counter, int n_wg, bool dead) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0 && !dead) {
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_wg) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { ok = false; break; }
        }
    }
    __syncthreads();
    return ok;
}

__device__ __forceinline__ uint32_t value_of(int k, int w) { return 0x9E3779B9u * (uint32_t)(k + 1) + (uint32_t)w * 0x85EBCA6Bu; }

__global__ void probe_grid_barrier_kernel(int* counters, uint32_t* buf, int n_barriers, int mode, int* stats) {
    extern __shared__ uint32_t lds_pad[];
    const int w = blockIdx.x, n_wg = gridDim.x;
    int mism = 0, timeouts = 0;
    for (int k = 0; k < n_barriers; ++k) {
        // publish: 256 threads write 256 consecutive dwords of this workgroup's line set
        uint32_t* mine = buf + ((size_t)(k & 1) * n_wg + w) * blockDim.x + threadIdx.x;
        const uint32_t v = value_of(k, w) + threadIdx.x;
        if (mode == 2) __builtin_nontemporal_store(v, mine);
        else if (mode == 3) __hip_atomic_store(mine, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *mine = v;
        if (mode == 1) __threadfence();
        if (!grid_barrier(counters + k, n_wg, timeouts > 0)) ++timeouts;
        if (mode == 1) __threadfence();
#pragma unroll
        for (int j = 1; j <= 8; ++j) {
            const int o = (w + j * 37 + k) % n_wg;
            const uint32_t* theirs = buf + ((size_t)(k & 1) * n_wg + o) * blockDim.x + threadIdx.x;
            uint32_t got;
            if (mode == 2) got = __builtin_nontemporal_load(theirs);
            else if (mode == 3) got = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else got = *theirs;
            if (got != value_of(k, o) + threadIdx.x) ++mism;
        }
    }
    if (mism) atomicAdd(stats, mism);
    if (timeouts && threadIdx.x == 0) atomicAdd(stats + 1, timeouts);
    if (mode == 99) stats[2] = (int)lds_pad[threadIdx.x];           // keep the dynamic LDS allocation referenced
}

}  // namespace

// ms_out: kernel time; stats_out[0] = stale/mismatched reads, [1] = barrier timeouts.  Returns 0 or a hipError_t.
extern "C" int dots_probe_grid_barrier(int n_wg, int threads, int n_barriers, int mode, int lds_bytes, float* ms_out, int32_t* stats_out) {
    if (n_wg < 1 || n_wg > 1024 || threads < 64 || threads > 1024 || n_barriers < 1 || n_barriers > 4096 || mode < 0 || mode > 3) return -1;
    int *counters = nullptr, *stats = nullptr;
    uint32_t* buf = nullptr;
    hipError_t e;
    if ((e = hipMalloc(&counters, n_barriers * sizeof(int))) != hipSuccess) return (int)e;
    if ((e = hipMalloc(&stats, 4 * sizeof(int))) != hipSuccess) return (int)e;
    if ((e = hipMalloc(&buf, (size_t)2 * n_wg * threads * 4)) != hipSuccess) return (int)e;
    (void)hipMemset(counters, 0, n_barriers * sizeof(int));
    (void)hipMemset(stats, 0, 4 * sizeof(int));
    (void)hipMemset(buf, 0, (size_t)2 * n_wg * threads * 4);
    if (lds_bytes > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)probe_grid_barrier_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a, nullptr);
    hipLaunchKernelGGL(probe_grid_barrier_kernel, dim3(n_wg), dim3(threads), lds_bytes, nullptr, counters, buf, n_barriers, mode, stats);
    (void)hipEventRecord(b, nullptr);
    e = hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    int32_t st[4] = {0, 0, 0, 0};
    (void)hipMemcpy(st, stats, sizeof(st), hipMemcpyDeviceToHost);
    if (ms_out) *ms_out = ms;
    if (stats_out) { stats_out[0] = st[0]; stats_out[1] = st[1]; }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipFree(counters);
    (void)hipFree(stats);
    (void)hipFree(buf);
    return (int)e;
}

// ------------------------------------------------------------------------------------------------
// CU-mask probe: which CUs does a stream created with hipExtStreamCreateWithCUMask run on?  Every workgroup records its
// raw HW_ID and XCC_ID registers and spins ~20 us so that the dispatcher has to spread the grid; the host counts
// distinct (xcc, se, sh, cu) tuples.  Used to choose the masks of the decode / admission streams.
namespace {
__global__ void probe_cu_id_kernel(uint32_t* out, int spin) {
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = __builtin_amdgcn_s_getreg(4 | (31 << 11));        // HW_REG_HW_ID
        out[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_getreg(20 | (31 << 11));   // HW_REG_XCC_ID
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);
}
}  // namespace

// mask == nullptr: the null stream (all CUs).  ids_out: uint32 [n_wg][2] raw register values.
extern "C" int dots_probe_cu_mask(const uint32_t* mask, int words, int n_wg, int threads, int lds_bytes, uint32_t* ids_out) {
    if (n_wg < 1 || n_wg > 65536 || !ids_out) return -1;
    hipStream_t s = nullptr;
    hipError_t e;
    if (mask && (e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask)) != hipSuccess) return (int)e;
    uint32_t* d = nullptr;
    if ((e = hipMalloc(&d, (size_t)n_wg * 8)) != hipSuccess) return (int)e;
    if (lds_bytes > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)probe_cu_id_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(probe_cu_id_kernel, dim3(n_wg), dim3(threads), lds_bytes, s, d, 200);
    e = hipStreamSynchronize(s);
    (void)hipMemcpy(ids_out, d, (size_t)n_wg * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (s) (void)hipStreamDestroy(s);
    return (int)e;
}
