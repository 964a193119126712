// Dataflow probe: is ONE launch holding several dependent phases ("roles", selected by block id; a later role prefetches its
// independent weight / KV stream, then spins — bounded — on the earlier role's completion counter) faster than one launch per
// phase on MI355X?  Synthetic roles with the byte counts of the dots.ocr decode layer at B = 8, ctx 5.7k:
//   test 1  [qkv -> attn]             A: 256 WGs x 24 KB weights + 24 KB rows, writes q;  B: 400 WGs x 4 waves x 32 KB KV, needs q
//   test 2  [comb -> o -> gateup -> down]   48 / 192 / 560 / 768 WGs, 0 / 24 / 98 / 36 KB streams, 25 / 24 / 24 / 36 KB fresh inputs
// Cross-workgroup payloads: write-through (sc1) stores, drain, relaxed agent counter add; consumers poll relaxed (one wave,
// s_sleep), then read with sc1 loads (guide §6 G16 R1).  Every spin is bounded; a timeout sets err and the run is reported bad.
// Each graph = 28 iterations over distinct (HBM-cold) stream buffers, like the 28 layers of a step; the small buffers are
// reused.  Results are verified: role B's checksum must contain the q words role A wrote in THIS launch (seed changes per launch).
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dataflow_probe.hip -o tools/bin/dataflow_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <functional>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define DEVI __device__ __forceinline__
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int L = 28;
constexpr unsigned SPIN_MAX = 200000;     // x ~0.5 us

DEVI u32x4 ld_sc1(const void* base, uint32_t byte_off) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);
}
DEVI uint32_t fold(u32x4 v) { return v[0] ^ v[1] ^ v[2] ^ v[3]; }
DEVI int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

// Hand-off protocol v2 (v1 — every consumer polling the counter the producers add to — measured 2x SLOWER than separate launches:
// 256 adds + thousands of polls on one word serialise at ~12 ns each):
//   arrive: 16 shard counters (128 B apart, shard = block & 15) with RETURNING adds; the last arriver of a shard adds to the top
//           counter; the last of those stores the READY flag into NREP replicas (128 B apart);
//   wait:   ONE wave per consumer WG polls ONE replica (block & (NREP-1)): a line that is written once — L2 hits until it flips.
// Sync block layout (uint32 words): [0..15]*32 shards, [16*32] top, [17*32 + r*32] replicas.
constexpr int NSH = 16, NREP = 16, SYNC_WORDS = (NSH + 1 + NREP) * 32;
DEVI bool wait_ready(uint32_t* sync, uint32_t* err) {
    uint32_t* flag = sync + (NSH + 1 + (blockIdx.x & (NREP - 1))) * 32;
    unsigned spins = 0;
    while (__hip_atomic_load(flag, RLX) == 0u) {
        __builtin_amdgcn_s_sleep(16);
        if (++spins > SPIN_MAX) { if ((threadIdx.x & 63) == 0) __hip_atomic_store(err, 1u, RLX); return false; }
    }
    asm volatile("" ::: "memory");
    return true;
}
// called by ONE lane after every storing wave of the WG has drained; n_total arrivals expected (a multiple of NSH)
DEVI void arrive(uint32_t* sync, int b, int n_total) {
    const uint32_t per = (uint32_t)n_total / NSH;
    if (__hip_atomic_fetch_add(sync + (b & (NSH - 1)) * 32, 1u, RLX) != per - 1) return;
    if (__hip_atomic_fetch_add(sync + NSH * 32, 1u, RLX) != NSH - 1) return;
#pragma unroll
    for (int r = 0; r < NREP; ++r) __hip_atomic_store(sync + (NSH + 1 + r) * 32, 1u, RLX);
}
DEVI uint32_t wave_xor(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o, 64);
    return v;
}


// Issue N 16-B nt loads keeping at most TH outstanding per wave (TH >= N: no throttle).  The extra waits only slow the ISSUE; the
// data stays in registers.  Purpose: bound the bytes the chip has in flight (queue depth = latency of every hand-off round trip).
template <int N, int TH>
DEVI void stream_issue(u32x4 (&dst)[N > 0 ? N : 1], const u32x4* src) {
#pragma unroll
    for (int c = 0; c < N; ++c) {
        dst[c] = __builtin_nontemporal_load(src + c * 64);
        if (TH < N && c >= TH - 1 && c < N - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TH - 1) : "memory");
    }
}

struct T1 {
    const u32x4* wA;      // [256 WG][4 waves][6][64]   24 KB per WG
    const u32x4* kv;      // [1472 chunks][32][64]      32 KB per wave
    uint32_t* h;          // 24 KB rows (reused)
    uint32_t* q;          // [256][32] words (reused)
    uint32_t* part;       // [400][768] words (reused)
    uint32_t* chk;        // [400] per iteration
    uint32_t* sync;       // [4][SYNC_WORDS] per iteration
    uint32_t* err;
    const uint32_t* seed;
    unsigned long long* starts;   // [656][4] (start, flag seen / before arrive, end) or null
    int delay;            // role B waits delay x 64 clocks before its prefetch flood
};

// MODE 0 fused (prefetch, late wave polls then loads its own page), 1 role A only, 2 role B only (no wait), 3 fused without prefetch (wait first),
// 4 fused, ALL waves prefetch, wave 0 polls afterwards (its poll returns behind its own page: in-order)
template <int MODE, int TH = 32>
__global__ __launch_bounds__(256) void k_qkv_attn(T1 t) {
    __shared__ uint32_t lds[4 * 64 + 128];
    const int lane = threadIdx.x & 63, w = wave_id();
    int b = blockIdx.x;
    unsigned long long* ts = t.starts ? t.starts + (size_t)(b + (MODE == 2 ? 256 : 0)) * 4 : nullptr;
    if (ts && threadIdx.x == 0) ts[0] = wall_clock64();
    const bool roleA = MODE == 1 || ((MODE == 0 || MODE == 3 || MODE == 4) && b < 256);
    if (roleA) {
        u32x4 rows[6], wt[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) rows[c] = ld_sc1(t.h, ((w * 6 + c) * 64 + lane) * 16);
#pragma unroll
        for (int c = 0; c < 6; ++c) wt[c] = __builtin_nontemporal_load(t.wA + ((size_t)(b * 4 + w) * 6 + c) * 64 + lane);
        uint32_t acc = 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) acc ^= fold(rows[c]) ^ fold(wt[c]);
        lds[w * 64 + lane] = acc;
        __syncthreads();
        if (w != 3) return;
        const uint32_t v = (lds[lane] ^ lds[64 + lane] ^ lds[128 + lane] ^ lds[192 + lane]) & 0;     // data-dependent zero: keeps the loads live
        if (lane < 32) __hip_atomic_store(t.q + b * 32 + lane, (v | (uint32_t)(b * 32 + lane)) ^ *t.seed, RLX);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) { if (ts) ts[1] = wall_clock64(); arrive(t.sync, b, 256); if (ts) ts[2] = wall_clock64(); }
        return;
    }
    if (MODE == 0 || MODE == 3 || MODE == 4) b -= 256;
    const int split = b % 25, bh = b / 25;
    if (split >= 23) { if (threadIdx.x == 0) arrive(t.sync + SYNC_WORDS, b, 400); return; }
    const u32x4* src = t.kv + ((size_t)((bh * 23 + split) * 4 + w) * 32) * 64 + lane;
    u32x4 kvr[32];
    bool ok = true;
    const bool late = (w == 0);
    if (MODE == 3) {            // everyone waits first
        if (late) { ok = wait_ready(t.sync, t.err); if (ts && lane == 0) ts[1] = wall_clock64(); }
        __syncthreads();
    }
    if (MODE == 0 || MODE == 4) for (int d = 0; d < t.delay; ++d) __builtin_amdgcn_s_sleep(1);
    if (!late || MODE == 3 || MODE == 4) stream_issue<32, TH>(kvr, src);
    if (late) {
        if (MODE == 0 || MODE == 4) { ok = wait_ready(t.sync, t.err); if (ts && lane == 0) ts[1] = wall_clock64(); }
        // q of this (b, hkv): 6 heads x 128 bf16 = 1.5 KB = 384 words: here words [bh * 384 .. +384) of the 8192-word q buffer
        uint32_t qv[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) qv[c] = __hip_atomic_load(t.q + bh * 384 + c * 64 + lane, RLX);
        uint32_t s = 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) { s ^= qv[c]; if (c < 2) lds[256 + c * 64 + lane] = qv[c]; }
        s = wave_xor(s);
        if (lane == 0) t.chk[b] = s;
        if (MODE != 3 && MODE != 4) stream_issue<32, TH>(kvr, src);
    }
    __syncthreads();
    uint32_t acc = lds[256 + lane] ^ lds[320 + lane];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc ^= fold(kvr[c]);
    lds[w * 64 + lane] = acc;
    __syncthreads();
    const uint32_t r = lds[lane] ^ lds[64 + lane] ^ lds[128 + lane] ^ lds[192 + lane];
#pragma unroll
    for (int c = 0; c < 3; ++c) __hip_atomic_store(t.part + (size_t)b * 768 + c * 256 + threadIdx.x, r + c, RLX);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) { arrive(t.sync + SYNC_WORDS, b, 400); if (ts) ts[2] = wall_clock64(); }
    (void)ok;
}

// ---- test 2: 4-role chain.  Role r: n[r] WGs, stream[r] KB per WG (4 waves x NS chunks of 1 KB... see table), fresh input in[r] KB per WG
struct T2 {
    const u32x4* w[4];      // streams of roles 0..3
    uint32_t* buf[5];       // buf[r] = input of role r (fresh, written by role r-1; buf[0] = partials written by the previous launch), buf[4] = final out
    uint32_t* sync;         // [4][SYNC_WORDS]
    uint32_t* err;
    uint32_t* chk;          // [4][1024]
    const uint32_t* seed;
    int delay, gate2;       // roles > 0: wait delay x 64 clocks before prefetching; gate2: role r >= 2 starts its prefetch when role r-2 is done
};
constexpr int R_N[4] = {48, 192, 560, 768};
constexpr int R_STREAM[4] = {0, 6, 24, 9};       // 1 KB chunks per wave (x4 waves): 0 / 24 / 96 / 36 KB per WG
constexpr int R_IN[4] = {6, 6, 6, 9};            // fresh-input chunks (1 KB) per wave: 24 / 24 / 24 / 36 KB per WG
constexpr int R_OUTW[4] = {128, 32, 64, 64};     // output words per WG (role 0: 48 x 128 = 6144 words = 24 KB X image, ...)

template <int R, bool FUSED, bool PALL, int TH = 32>
DEVI void chain_role(const T2& t, int b, uint32_t* lds) {
    const int lane = threadIdx.x & 63, w = wave_id();
    constexpr int NS = R_STREAM[R], NI = R_IN[R];
    u32x4 st[NS > 0 ? NS : 1];
    const bool late = (w == 0) && FUSED && R > 0 && !PALL;
    const u32x4* src = t.w[R] + ((size_t)(b * 4 + w) * (NS > 0 ? NS : 1)) * 64 + lane;
    if (FUSED && R > 0) {
        if (R >= 2 && t.gate2) { if (w == 0) wait_ready(t.sync + (R - 2) * SYNC_WORDS, t.err); __syncthreads(); }
        else for (int d = 0; d < t.delay; ++d) __builtin_amdgcn_s_sleep(1);
    }
    if (!late) stream_issue<NS, TH>(st, src);
    __builtin_amdgcn_sched_barrier(0);
    if (FUSED && R > 0) {
        if (w == 0) {
            wait_ready(t.sync + (R - 1) * SYNC_WORDS, t.err);
            if (late) stream_issue<NS, TH>(st, src);
        }
        __syncthreads();
    }
    // fresh input: every WG reads the same in[r] KB (all-gather), L1-bypassing
    u32x4 in[NI];
#pragma unroll
    for (int c = 0; c < NI; ++c) in[c] = ld_sc1(t.buf[R], ((w * NI + c) * 64 + lane) * 16);
    __builtin_amdgcn_sched_barrier(0);
    uint32_t s_in = 0, acc = 0;
#pragma unroll
    for (int c = 0; c < NI; ++c) s_in ^= fold(in[c]);
#pragma unroll
    for (int c = 0; c < NS; ++c) acc ^= fold(st[c]);
    lds[w * 64 + lane] = s_in;
    lds[256 + w * 64 + lane] = acc;
    __syncthreads();
    if (w != 3) return;
    uint32_t sx = lds[lane] ^ lds[64 + lane] ^ lds[128 + lane] ^ lds[192 + lane];
    const uint32_t z = (lds[256 + lane] ^ lds[320 + lane] ^ lds[384 + lane] ^ lds[448 + lane]) & 0;
    sx = wave_xor(sx);                                  // xor of the whole fresh input: identical in every WG of the role
    if (lane == 0) t.chk[R * 1024 + b] = sx;
    // output: word index i of the next buffer <- f(seed, role, i); covers buf[R+1] completely when summed over the role's WGs
    constexpr int OW = R_OUTW[R];
    for (int i = lane; i < OW; i += 64) __hip_atomic_store(t.buf[R + 1] + b * OW + i, (z | (uint32_t)(b * OW + i)) * 2654435761u ^ (*t.seed + R), RLX);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) arrive(t.sync + R * SYNC_WORDS, b, R_N[R]);
}

template <int MODE, int TH = 32>      // 0 fused (late wave), 5 fused (all waves prefetch, wave 0 polls afterwards), 1..4 = role MODE-1 alone
__global__ __launch_bounds__(256) void k_chain(T2 t) {
    __shared__ uint32_t lds[512];
    const int b = blockIdx.x;
    if (MODE == 0 || MODE == 5) {
        constexpr bool P = MODE == 5;
        if (b < 48) chain_role<0, true, P, TH>(t, b, lds);
        else if (b < 48 + 192) chain_role<1, true, P, TH>(t, b - 48, lds);
        else if (b < 48 + 192 + 560) chain_role<2, true, P, TH>(t, b - 240, lds);
        else chain_role<3, true, P, TH>(t, b - 800, lds);
    } else if (MODE == 1) chain_role<0, false, false>(t, b, lds);
    else if (MODE == 2) chain_role<1, false, false>(t, b, lds);
    else if (MODE == 3) chain_role<2, false, false>(t, b, lds);
    else chain_role<3, false, false>(t, b, lds);
}

static hipStream_t S;
static double time_graph(const std::function<void()>& body, const std::function<void()>& before, int reps = 7) {
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(S, hipStreamCaptureModeThreadLocal));
    body();
    CK(hipStreamEndCapture(S, &g));
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    before(); CK(hipGraphLaunch(ex, S)); CK(hipStreamSynchronize(S));
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        before();
        CK(hipEventRecord(a, S)); CK(hipGraphLaunch(ex, S)); CK(hipEventRecord(b, S)); CK(hipStreamSynchronize(S));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    hipGraphExecDestroy(ex); hipGraphDestroy(g);
    return best * 1e3;
}
template <typename T> static T* dalloc(size_t n, int fill = 0x3c) { void* p; CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, fill, n * sizeof(T))); return (T*)p; }

int main() {
    CK(hipSetDevice(0));
    CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
    uint32_t* seed = dalloc<uint32_t>(1, 0);
    uint32_t* err = dalloc<uint32_t>(1, 0);
    uint32_t* cnt = dalloc<uint32_t>((size_t)L * 4 * SYNC_WORDS, 0);
    uint32_t h_seed = 1000;
    auto bump = [&]() { ++h_seed; CK(hipMemcpyAsync(seed, &h_seed, 4, hipMemcpyHostToDevice, S)); };
    auto zero_cnt = [&]() { CK(hipMemsetAsync(cnt, 0, (size_t)L * 4 * SYNC_WORDS * 4, S)); };
    auto check_err = [&](const char* what) { uint32_t e; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost)); if (e) { printf("  !! %s: SPIN TIMEOUT\n", what); CK(hipMemset(err, 0, 4)); } };

    // ---------------- test 1
    {
        const size_t wA_it = (size_t)256 * 4 * 6 * 64, kv_it = (size_t)1472 * 32 * 64;
        u32x4* wA = dalloc<u32x4>(wA_it * L);
        u32x4* kv = dalloc<u32x4>(kv_it * L);
        uint32_t *h = dalloc<uint32_t>(6144), *q = dalloc<uint32_t>(8192), *part = dalloc<uint32_t>(400 * 768), *chk = dalloc<uint32_t>(L * 400, 0);
        unsigned long long* starts = dalloc<unsigned long long>(4096, 0);
        auto mk0 = [&](int i, bool st) { T1 t{wA + wA_it * i, kv + kv_it * i, h, q, part, chk + 400 * i, cnt + (size_t)i * 4 * SYNC_WORDS, err, seed, st ? starts : nullptr, 0}; return t; };
        auto verify = [&](const char* what) {
            std::vector<uint32_t> c(L * 400);
            CK(hipMemcpy(c.data(), chk, c.size() * 4, hipMemcpyDeviceToHost));
            int bad = 0;
            for (int i = 0; i < L; ++i)
                for (int b = 0; b < 400; ++b) {
                    if (b % 25 >= 23) continue;
                    const int bh = b / 25;
                    uint32_t want = 0;
                    for (int k = 0; k < 384; ++k) want ^= (uint32_t)(bh * 384 + k) ^ h_seed;
                    if (c[i * 400 + b] != want) ++bad;
                }
            printf("  %-34s q seen by role B: %s (%d stale of %d)\n", what, bad ? "STALE" : "fresh", bad, L * 368);
            check_err(what);
        };
        printf("== test 1: [qkv 256 WG x 24 KB] -> [attn 400 WG x 128 KB], %d iterations per graph\n", L);
        const double t_sep = time_graph([&]() { zero_cnt(); for (int i = 0; i < L; ++i) { hipLaunchKernelGGL(k_qkv_attn<1>, dim3(256), dim3(256), 0, S, mk0(i, false)); hipLaunchKernelGGL(k_qkv_attn<2>, dim3(400), dim3(256), 0, S, mk0(i, false)); } }, bump);
        verify("separate launches");
        const double t_a = time_graph([&]() { zero_cnt(); for (int i = 0; i < L; ++i) hipLaunchKernelGGL(k_qkv_attn<1>, dim3(256), dim3(256), 0, S, mk0(i, false)); }, bump);
        const double t_b = time_graph([&]() { zero_cnt(); for (int i = 0; i < L; ++i) hipLaunchKernelGGL(k_qkv_attn<2>, dim3(400), dim3(256), 0, S, mk0(i, false)); }, bump);
        int g_delay = 0;
        auto mk = [&](int i, bool st) { T1 t = mk0(i, st); t.delay = g_delay; return t; };
        printf("  per iteration: A alone %.2f us, B alone %.2f us, A then B (2 launches) %.2f us\n", t_a / L, t_b / L, t_sep / L);
        const double t_n = time_graph([&]() { zero_cnt(); for (int i = 0; i < L; ++i) hipLaunchKernelGGL(k_qkv_attn<3>, dim3(656), dim3(256), 0, S, mk(i, false)); }, bump);
        verify("fused, wait before loading");
        printf("  fused without prefetch %.2f us\n", t_n / L);
        const int delays[] = {0, 32};
        for (int d : delays) {
            g_delay = d;
            const double t_f = time_graph([&]() { zero_cnt(); for (int i = 0; i < L; ++i) hipLaunchKernelGGL(k_qkv_attn<0>, dim3(656), dim3(256), 0, S, mk(i, false)); }, bump);
            verify("fused, late wave");
            const double t_p = time_graph([&]() { zero_cnt(); for (int i = 0; i < L; ++i) hipLaunchKernelGGL(k_qkv_attn<4>, dim3(656), dim3(256), 0, S, mk(i, false)); }, bump);
            verify("fused, all prefetch + poll after");
            printf("  delay %2d x 64 clk: FUSED late-wave %.2f us, all-prefetch %.2f us\n", d, t_f / L, t_p / L);
        }
        g_delay = 0;
#define T1_TH(TH) do { \
            const double tl = time_graph([&]() { zero_cnt(); for (int i = 0; i < L; ++i) hipLaunchKernelGGL((k_qkv_attn<0, TH>), dim3(656), dim3(256), 0, S, mk(i, false)); }, bump); \
            verify("fused late wave, throttled"); \
            const double tp = time_graph([&]() { zero_cnt(); for (int i = 0; i < L; ++i) hipLaunchKernelGGL((k_qkv_attn<4, TH>), dim3(656), dim3(256), 0, S, mk(i, i == L - 1 && TH == 8)); }, bump); \
            verify("fused all prefetch, throttled"); \
            const double tb = time_graph([&]() { zero_cnt(); for (int i = 0; i < L; ++i) { hipLaunchKernelGGL((k_qkv_attn<1, TH>), dim3(256), dim3(256), 0, S, mk0(i, false)); hipLaunchKernelGGL((k_qkv_attn<2, TH>), dim3(400), dim3(256), 0, S, mk0(i, false)); } }, bump); \
            printf("  throttle %2d loads/wave: FUSED late-wave %.2f us, all-prefetch %.2f us; separate launches with the same throttle %.2f us\n", TH, tl / L, tp / L, tb / L); } while (0)
        T1_TH(2); T1_TH(4); T1_TH(8); T1_TH(16);
        std::vector<unsigned long long> st(4096);
        CK(hipMemcpy(st.data(), starts, 4096 * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < 656; ++b) if (st[b * 4]) t0 = std::min(t0, st[b * 4]);
        auto stat = [&](int lo, int hi, int slot, const char* name) {
            double mn = 1e30, mx = 0, sum = 0; int n = 0;
            for (int b = lo; b < hi; ++b) { if (!st[b * 4 + slot]) continue; const double us = (double)(st[b * 4 + slot] - t0) * 0.01; mn = std::min(mn, us); mx = std::max(mx, us); sum += us; ++n; }
            if (n) printf("    %-28s %6.2f / %6.2f / %6.2f us (min / mean / max over %d WGs)\n", name, mn, sum / n, mx, n);
        };
        printf("  timeline of the last all-prefetch iteration at throttle 8 (us since the first WG started):\n");
        stat(0, 256, 0, "role A start"); stat(0, 256, 1, "role A stores drained"); stat(0, 256, 2, "role A arrive returned");
        stat(256, 656, 0, "role B start"); stat(256, 656, 1, "role B saw READY"); stat(256, 656, 2, "role B end");
    }
    // ---------------- test 2
    {
        size_t w_it[4];
        u32x4* w[4];
        for (int r = 0; r < 4; ++r) { w_it[r] = (size_t)R_N[r] * 4 * (R_STREAM[r] ? R_STREAM[r] : 1) * 64; w[r] = dalloc<u32x4>(w_it[r] * L); }
        uint32_t* buf[5];
        for (int r = 0; r < 5; ++r) buf[r] = dalloc<uint32_t>(65536);
        uint32_t* chk = dalloc<uint32_t>(4096, 0);
        int g_delay2 = 0, g_gate2 = 0;
        auto mk = [&](int i) { T2 t; for (int r = 0; r < 4; ++r) t.w[r] = w[r] + w_it[r] * i; for (int r = 0; r < 5; ++r) t.buf[r] = buf[r]; t.sync = cnt + (size_t)i * 4 * SYNC_WORDS; t.err = err; t.chk = chk; t.seed = seed; t.delay = g_delay2; t.gate2 = g_gate2; return t; };
        auto verify = [&](const char* what) {
            std::vector<uint32_t> c(4096);
            CK(hipMemcpy(c.data(), chk, c.size() * 4, hipMemcpyDeviceToHost));
            int bad = 0, tot = 0;
            for (int r = 1; r < 4; ++r) {      // role r read buf[r] = role r-1's output: words [0, R_IN[r]*4 KB / 4)
                const int words = R_IN[r] * 4 * 256;
                uint32_t want = 0;
                const int prod = R_N[r - 1] * R_OUTW[r - 1];
                for (int i = 0; i < words; ++i) want ^= i < prod ? ((uint32_t)i * 2654435761u ^ (h_seed + (r - 1))) : 0x3c3c3c3cu;
                for (int b = 0; b < R_N[r]; ++b) { ++tot; if (c[r * 1024 + b % 1024] != want) ++bad; }
            }
            printf("  %-34s inputs seen: %s (%d stale of %d)\n", what, bad ? "STALE" : "fresh", bad, tot);
            check_err(what);
        };
        printf("== test 2: [comb 48] -> [o 192 x 24 KB] -> [gateup 560 x 96 KB] -> [down 768 x 36 KB], %d iterations per graph\n", L);
        const double t_sep = time_graph([&]() { zero_cnt(); for (int i = 0; i < L; ++i) {
            hipLaunchKernelGGL(k_chain<1>, dim3(48), dim3(256), 0, S, mk(i)); hipLaunchKernelGGL(k_chain<2>, dim3(192), dim3(256), 0, S, mk(i));
            hipLaunchKernelGGL(k_chain<3>, dim3(560), dim3(256), 0, S, mk(i)); hipLaunchKernelGGL(k_chain<4>, dim3(768), dim3(256), 0, S, mk(i)); } }, bump);
        verify("separate launches");
        double t_r[4];
        t_r[0] = time_graph([&]() { for (int i = 0; i < L; ++i) hipLaunchKernelGGL(k_chain<1>, dim3(48), dim3(256), 0, S, mk(i)); }, bump);
        t_r[1] = time_graph([&]() { for (int i = 0; i < L; ++i) hipLaunchKernelGGL(k_chain<2>, dim3(192), dim3(256), 0, S, mk(i)); }, bump);
        t_r[2] = time_graph([&]() { for (int i = 0; i < L; ++i) hipLaunchKernelGGL(k_chain<3>, dim3(560), dim3(256), 0, S, mk(i)); }, bump);
        t_r[3] = time_graph([&]() { for (int i = 0; i < L; ++i) hipLaunchKernelGGL(k_chain<4>, dim3(768), dim3(256), 0, S, mk(i)); }, bump);
        printf("  per iteration: roles alone %.2f / %.2f / %.2f / %.2f us, 4 launches %.2f us\n", t_r[0] / L, t_r[1] / L, t_r[2] / L, t_r[3] / L, t_sep / L);
        const int cfgs[][2] = {{0, 0}, {0, 1}};
        for (auto& c : cfgs) {
            g_delay2 = c[0]; g_gate2 = c[1];
            const double t_f = time_graph([&]() { zero_cnt(); for (int i = 0; i < L; ++i) hipLaunchKernelGGL(k_chain<0>, dim3(48 + 192 + 560 + 768), dim3(256), 0, S, mk(i)); }, bump);
            verify("fused 4 roles, late wave");
            const double t_p = time_graph([&]() { zero_cnt(); for (int i = 0; i < L; ++i) hipLaunchKernelGGL(k_chain<5>, dim3(48 + 192 + 560 + 768), dim3(256), 0, S, mk(i)); }, bump);
            verify("fused 4 roles, all prefetch");
            printf("  delay %2d gate2 %d: FUSED late-wave %.2f us, all-prefetch %.2f us\n", c[0], c[1], t_f / L, t_p / L);
        }
#define T2_TH(TH, G) do { g_delay2 = 0; g_gate2 = G; \
            const double tl = time_graph([&]() { zero_cnt(); for (int i = 0; i < L; ++i) hipLaunchKernelGGL((k_chain<0, TH>), dim3(48 + 192 + 560 + 768), dim3(256), 0, S, mk(i)); }, bump); \
            verify("fused late wave, throttled"); \
            const double tp = time_graph([&]() { zero_cnt(); for (int i = 0; i < L; ++i) hipLaunchKernelGGL((k_chain<5, TH>), dim3(48 + 192 + 560 + 768), dim3(256), 0, S, mk(i)); }, bump); \
            verify("fused all prefetch, throttled"); \
            printf("  throttle %2d gate2 %d: FUSED late-wave %.2f us, all-prefetch %.2f us\n", TH, G, tl / L, tp / L); } while (0)
        T2_TH(2, 0); T2_TH(4, 0); T2_TH(8, 0); T2_TH(4, 1); T2_TH(8, 1);
    }
    return 0;
}
