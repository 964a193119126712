// Streaming / latency / launch-boundary probe for the decode-step design (standalone, no Python):
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bw_probe.hip -o tools/bin/bw_probe && tools/bin/bw_probe
// Answers, on one MI355X:
//   A  how fast ONE workgroup / CU can pull a contiguous weight stream, as a function of how many CUs pull at once
//      (is the per-CU rate ~ chip bandwidth / 256, or can a lone CU go faster?)
//   B  time of one whole "GEMV-sized" stream (4.7 / 6.3 / 27.5 / 55 MB) for different grid shapes, HBM-cold
//   C  cost of a dependent kernel boundary inside a captured graph, with 0 / 1 / 2 dependent loads in the kernel
//   D  dependent-load latency: L2-resident, Infinity-Cache-resident, HBM
// Every number is "graph of N launches / N", HBM-cold buffers rotate through a 6 GiB arena.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <functional>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// wave streams `chunks` KiB starting at base + wave_global_index * chunks KiB, DEPTH loads in flight
template <int DEPTH, bool NT>
__global__ __launch_bounds__(1024) void stream_kernel(const u32x4* __restrict__ base, int chunks, uint32_t* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const u32x4* p = base + ((size_t)(blockIdx.x * nw + wv) * chunks) * 64 + lane;
    u32x4 acc = {0, 0, 0, 0};
    for (int c = 0; c < chunks; c += DEPTH) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int j = 0; j < DEPTH; ++j)
            if (c + j < chunks) v[j] = NT ? __builtin_nontemporal_load(p + (size_t)(c + j) * 64) : p[(size_t)(c + j) * 64];
#pragma unroll
        for (int j = 0; j < DEPTH; ++j)
            if (c + j < chunks) acc ^= v[j];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

__global__ void empty_kernel(uint32_t* p) { if (p == nullptr) __builtin_trap(); }
// one / two dependent loads then a store (what a "trivial" bookkeeping kernel of the decode step does)
__global__ void dep1_kernel(const uint32_t* __restrict__ a, uint32_t* __restrict__ out) {
    out[blockIdx.x * 64 + threadIdx.x] = a[blockIdx.x * 64 + threadIdx.x] + 1;
}
__global__ void dep2_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t* __restrict__ out) {
    const uint32_t i = a[blockIdx.x * 64 + threadIdx.x] & 1023;
    out[blockIdx.x * 64 + threadIdx.x] = b[i * 64 + threadIdx.x] + 1;
}
// pointer chase, one lane
__global__ void chase_kernel(const uint32_t* __restrict__ next, int steps, uint32_t* out, long long* cycles) {
    uint32_t i = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) i = __builtin_nontemporal_load(next + (size_t)i * 32);
    const long long t1 = __builtin_readcyclecounter();
    out[0] = i;
    cycles[0] = t1 - t0;
}
__global__ void touch_kernel(const u32x4* __restrict__ p, size_t n16, uint32_t* sink) {
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

static hipStream_t S;
static double time_graph(const std::function<void(int)>& launch, int n, int reps = 3) {
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(S, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; ++i) launch(i);
    CK(hipStreamEndCapture(S, &g));
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipGraphLaunch(ex, S)); CK(hipStreamSynchronize(S));
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a, S)); CK(hipGraphLaunch(ex, S)); CK(hipEventRecord(b, S)); CK(hipStreamSynchronize(S));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    hipGraphExecDestroy(ex); hipGraphDestroy(g); hipEventDestroy(a); hipEventDestroy(b);
    return best * 1e3 / n;   // us per launch
}

template <int DEPTH, bool NT>
static double run_stream(const u32x4* arena, size_t arena_bytes, int wgs, int waves, int chunks, uint32_t* sink, int n = 24) {
    const size_t bytes = (size_t)wgs * waves * chunks * 1024;
    const size_t slots = arena_bytes / bytes;
    return time_graph([&](int i) {
        const u32x4* base = arena + ((size_t)(i % slots) * bytes) / 16;
        hipLaunchKernelGGL((stream_kernel<DEPTH, NT>), dim3(wgs), dim3(waves * 64), 0, S, base, chunks, sink);
    }, n);
}

int main(int argc, char** argv) {
    CK(hipSetDevice(0));
    CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
    const size_t ARENA = (size_t)6 << 30;
    u32x4* arena; uint32_t* sink;
    CK(hipMalloc(&arena, ARENA)); CK(hipMalloc(&sink, 1 << 20));
    CK(hipMemset(arena, 1, ARENA)); CK(hipMemset(sink, 0, 1 << 20));
    CK(hipDeviceSynchronize());

    printf("== A: per-workgroup pull rate vs number of workgroups (16 waves x 8 KiB in flight, nt loads, 1 MiB per WG)\n");
    for (int wgs : {1, 8, 32, 64, 128, 256, 512}) {
        const int chunks = 64;                      // 64 KiB per wave, 1 MiB per WG
        double us = run_stream<8, true>(arena, ARENA, wgs, 16, chunks, sink);
        double us2 = run_stream<8, true>(arena, ARENA, wgs, 16, 2 * chunks, sink);
        printf("  wgs %4d: 1MiB/WG %8.2f us, 2MiB/WG %8.2f us -> marginal %.1f GB/s per WG, %.2f TB/s total\n", wgs, us, us2,
               1048576.0 / ((us2 - us) * 1e-6) / 1e9, wgs * 1048576.0 / ((us2 - us) * 1e-6) / 1e12);
    }
    printf("== A2: same with 4 waves x 24 KiB in flight\n");
    for (int wgs : {1, 64, 256, 512, 1024}) {
        double us = run_stream<24, true>(arena, ARENA, wgs, 4, 96, sink);
        double us2 = run_stream<24, true>(arena, ARENA, wgs, 4, 192, sink);
        printf("  wgs %4d: 384KiB/WG %8.2f us, 768KiB/WG %8.2f us -> marginal %.1f GB/s per WG, %.2f TB/s total\n", wgs, us, us2,
               393216.0 / ((us2 - us) * 1e-6) / 1e9, wgs * 393216.0 / ((us2 - us) * 1e-6) / 1e12);
    }

    printf("== B: one GEMV-sized stream, HBM-cold, us per launch (includes the launch boundary)\n");
    struct Shape { const char* name; int total_kib; };
    for (Shape sh : {Shape{"o_proj 4.7MB", 4608}, Shape{"qkv 6.3MB", 6144}, Shape{"down 27.5MB", 26880}, Shape{"gateup 55MB", 53760}}) {
        printf("  %s:", sh.name);
        for (int wgs : {64, 96, 128, 192, 256, 512, 768, 1024}) {
            for (int waves : {4, 8, 16}) {
                const int per_wave = sh.total_kib / (wgs * waves);
                if (per_wave < 1 || per_wave * wgs * waves != sh.total_kib) continue;
                double us = per_wave <= 4 ? run_stream<4, true>(arena, ARENA, wgs, waves, per_wave, sink)
                          : per_wave <= 8 ? run_stream<8, true>(arena, ARENA, wgs, waves, per_wave, sink)
                          : per_wave <= 12 ? run_stream<12, true>(arena, ARENA, wgs, waves, per_wave, sink)
                                           : run_stream<16, true>(arena, ARENA, wgs, waves, per_wave, sink);
                printf("  [%dx%dw %dK] %.2f", wgs, waves, per_wave, us);
            }
        }
        printf("\n");
    }
    printf("== B2: plain (allocating) loads instead of nt, 256 WGs x 16 waves\n");
    for (int total : {4608, 6144, 26880, 53760}) {
        const int per_wave = total / (256 * 16) > 0 ? total / (256 * 16) : 1;
        double a = run_stream<8, true>(arena, ARENA, 256, 16, per_wave, sink), b = run_stream<8, false>(arena, ARENA, 256, 16, per_wave, sink);
        printf("  %5d KiB (%d KiB/wave): nt %.2f us, plain %.2f us\n", 256 * 16 * per_wave, per_wave, a, b);
    }
    printf("== B3: imbalance: 55 MB as 560 WGs x 4 waves x 24 KiB vs 512 WGs and 768 WGs of the same size\n");
    for (int wgs : {256, 512, 560, 640, 768}) printf("  %d WGs: %.2f us (%.2f TB/s)\n", wgs, run_stream<24, true>(arena, ARENA, wgs, 4, 24, sink),
                                                   wgs * 4 * 24 * 1024.0 / (run_stream<24, true>(arena, ARENA, wgs, 4, 24, sink) * 1e-6) / 1e12);
    printf("== B4: repeated reads of the SAME 55 MB / 6.3 MB (Infinity Cache resident?), plain loads\n");
    for (int total : {6144, 53760}) {
        const int per_wave = total / (256 * 16);
        double us = time_graph([&](int) { hipLaunchKernelGGL((stream_kernel<8, false>), dim3(256), dim3(1024), 0, S, arena, per_wave, sink); }, 24);
        double usn = time_graph([&](int) { hipLaunchKernelGGL((stream_kernel<8, true>), dim3(256), dim3(1024), 0, S, arena, per_wave, sink); }, 24);
        printf("  %5d KiB same buffer: plain %.2f us, nt %.2f us\n", 256 * 16 * per_wave, us, usn);
    }

    printf("== C: kernel boundary inside a graph\n");
    uint32_t *a, *b, *o;
    CK(hipMalloc(&a, 1 << 22)); CK(hipMalloc(&b, 1 << 22)); CK(hipMalloc(&o, 1 << 22));
    CK(hipMemset(a, 0, 1 << 22)); CK(hipMemset(b, 0, 1 << 22));
    for (int wgs : {8, 256, 1024}) {
        double e = time_graph([&](int) { hipLaunchKernelGGL(empty_kernel, dim3(wgs), dim3(64), 0, S, sink); }, 200);
        double d1 = time_graph([&](int i) { hipLaunchKernelGGL(dep1_kernel, dim3(wgs), dim3(64), 0, S, (i & 1) ? a : o, (i & 1) ? o : a); }, 200);
        double d2 = time_graph([&](int i) { hipLaunchKernelGGL(dep2_kernel, dim3(wgs), dim3(64), 0, S, (i & 1) ? a : o, b, (i & 1) ? o : a); }, 200);
        printf("  %4d WGs x 64: empty %.2f us, load->store chain %.2f us, load->load->store %.2f us\n", wgs, e, d1, d2);
    }
    for (int thr : {256, 1024}) {
        double e = time_graph([&](int) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(thr), 0, S, sink); }, 200);
        printf("  256 WGs x %d threads: empty %.2f us\n", thr, e);
    }

    printf("== D: dependent-load latency (one lane, stride 128 B, nt loads)\n");
    {
        uint32_t* nxt; long long* cyc;
        CK(hipMalloc(&cyc, 64));
        for (size_t foot : {(size_t)256 << 10, (size_t)2 << 20, (size_t)64 << 20, (size_t)2 << 30}) {
            const size_t n = foot / 128;
            std::vector<uint32_t> h(n * 32, 0);
            // random cyclic permutation
            std::vector<uint32_t> perm(n);
            for (size_t i = 0; i < n; ++i) perm[i] = (uint32_t)i;
            uint64_t s = 88172645463325252ull;
            for (size_t i = n - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; size_t j = s % (i + 1); std::swap(perm[i], perm[j]); }
            for (size_t i = 0; i < n; ++i) h[(size_t)perm[i] * 32] = perm[(i + 1) % n];
            CK(hipMalloc(&nxt, n * 128));
            CK(hipMemcpy(nxt, h.data(), n * 128, hipMemcpyHostToDevice));
            const int steps = 2000;
            hipLaunchKernelGGL(chase_kernel, dim3(1), dim3(1), 0, S, nxt, steps, o, cyc);   // warm
            hipLaunchKernelGGL(chase_kernel, dim3(1), dim3(1), 0, S, nxt, steps, o, cyc);
            CK(hipStreamSynchronize(S));
            long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            printf("  footprint %8zu KiB: %.0f cycles per load (s_memtime / readcyclecounter units)\n", foot >> 10, (double)c / steps);
            CK(hipFree(nxt));
        }
    }
    return 0;
}
