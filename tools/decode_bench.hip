// Decode-step microbenchmark at the BASELINE dimensions (standalone, no Python):
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idots_ocr_amd/csrc -Iinclude tools/decode_bench.hip \
//           -Ldots_ocr_amd/lib -ldots_ocr_hip -Wl,-rpath,'$ORIGIN/../../dots_ocr_amd/lib' -o tools/bin/decode_bench
//     tools/bin/decode_bench [B=8] [ctx=5700] [max_seq_len=6224]
// It calls the engine's own kernel launchers (kernels.h) in the order of engine.hip's decode_step_launches over 28 layers
// of DISTINCT weight buffers (HBM-cold, like the real step: 3.1 GB of weights per step), inside a captured graph, and
// reports  (1) the whole step,  (2) each kernel kind alone: a graph of 28 launches over the 28 layers' buffers.
// Values are constant-filled; the decode kernels are bandwidth / latency bound and data independent.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <cmath>
#include <functional>
#include <vector>

#include "kernels.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static hipStream_t S;

#ifdef DOTS_TRACE
void dots_trace_set_fused(unsigned long long* buf);
void dots_trace_set_decode(unsigned long long* buf);
void dots_trace_set_b64(unsigned long long* buf);
static unsigned long long* g_trace = nullptr;
constexpr size_t TRACE_WORDS = (size_t)4096 * 16 * 8;
// per-slot statistics of the LAST launch that wrote the trace buffer: offsets in us from the earliest slot-0 stamp
static void trace_report(const char* name, int n_slots) {
    std::vector<unsigned long long> h(TRACE_WORDS);
    CK(hipMemcpy(h.data(), g_trace, TRACE_WORDS * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (size_t w = 0; w < TRACE_WORDS / 8; ++w) if (h[w * 8]) t0 = std::min(t0, h[w * 8]);
    printf("    trace %-20s", name);
    for (int sl = 0; sl < n_slots; ++sl) {
        double sum = 0, mx = 0, mn = 1e30; size_t n = 0;
        for (size_t w = 0; w < TRACE_WORDS / 8; ++w) {
            if (!h[w * 8 + sl] || !h[w * 8]) continue;
            const double us = (double)(h[w * 8 + sl] - t0) * 0.01;
            sum += us; mx = std::max(mx, us); mn = std::min(mn, us); ++n;
        }
        if (n) printf("  [%d] %.2f/%.2f/%.2f", sl, mn, sum / n, mx);
    }
    printf("   (min/mean/max us since the first wave started, over waves)\n");
    CK(hipMemset(g_trace, 0, TRACE_WORDS * 8));
}
#else
static void trace_report(const char*, int) {}
#endif

static double time_graph(const std::function<void()>& body, int reps = 5) {
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(S, hipStreamCaptureModeThreadLocal));
    body();
    CK(hipStreamEndCapture(S, &g));
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipGraphLaunch(ex, S)); CK(hipStreamSynchronize(S));
    double best = 1e30, sum = 0;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a, S)); CK(hipGraphLaunch(ex, S)); CK(hipEventRecord(b, S)); CK(hipStreamSynchronize(S));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
        sum += ms;
    }
    hipGraphExecDestroy(ex); hipGraphDestroy(g); hipEventDestroy(a); hipEventDestroy(b);
    return best * 1e3;   // us
}

template <typename T> static T* dalloc(size_t n, int fill = 0) {
    void* p; CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, fill, n * sizeof(T))); return (T*)p;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8;
    const int ctx = argc > 2 ? atoi(argv[2]) : 5700;
    const int max_seq = argc > 3 ? atoi(argv[3]) : 6224;
    const int L = 28, H = 1536, Hq = 12, Hkv = 2, I = 8960, V = 151936, Nq = Hq * 128, NQKV = (Hq + 2 * Hkv) * 128;
    const float eps = 1e-6f;
    CK(hipSetDevice(0));
    if (const char* cr = getenv("DOTS_BENCH_CUS")) {          // run on the first N CU-mask bits only (N / 8 CUs of every XCD): the decode partition of the pipelined step
        uint32_t words[8] = {0};
        for (int b = 0; b < atoi(cr) && b < 256; ++b) words[b / 32] |= 1u << (b % 32);
        CK(hipExtStreamCreateWithCUMask(&S, 8, words));
        printf("stream masked to %d CUs\n", atoi(cr));
    } else
    CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
#ifdef DOTS_TRACE
    CK(hipMalloc(&g_trace, TRACE_WORDS * 8));
    CK(hipMemset(g_trace, 0, TRACE_WORDS * 8));
    dots_trace_set_fused(g_trace);
    dots_trace_set_decode(g_trace);
    dots_trace_set_b64(g_trace);
#endif
    const int max_pages = (max_seq + 63) / 64;
    const int n_splits = decode_attn_splits(max_seq);
    const int R = std::max(16, (B + 15) / 16 * 16);          // rows the buffers hold (round 5: up to 64 rows = the pipelined step's batch)
    // weights: 0x3c3c = bf16 0.0115.  DOTS_BENCH_FP8=1: the e4m3 instantiations (1 byte per weight, 0x3c = 1.5; scales 0x3c3c3c3c = 0.0115)
    const bool fp8 = getenv("DOTS_BENCH_FP8") != nullptr;
    const size_t wb = fp8 ? 1 : 2;
    std::vector<bf16_t*> ln1(L), ln2(L), bias(L);
    std::vector<uint8_t*> qkv(L), o(L), w13(L), down(L);
    for (int i = 0; i < L; ++i) {
        qkv[i] = dalloc<uint8_t>((size_t)NQKV * H * wb, 0x3c); o[i] = dalloc<uint8_t>((size_t)H * Nq * wb, 0x3c);
        w13[i] = dalloc<uint8_t>((size_t)2 * I * H * wb, 0x3c); down[i] = dalloc<uint8_t>((size_t)H * I * wb, 0x3c);
        ln1[i] = dalloc<bf16_t>(H, 0x3f); ln2[i] = dalloc<bf16_t>(H, 0x3f); bias[i] = dalloc<bf16_t>(NQKV, 0x3c);
    }
    uint8_t* lm = dalloc<uint8_t>((size_t)V * H * wb, 0x3c);
    float* wsc = fp8 ? dalloc<float>((size_t)V, 0x3c) : nullptr;
    printf("weights: %s\n", fp8 ? "e4m3 + per-channel scale" : "bf16");
    bf16_t* embed = dalloc<bf16_t>((size_t)V * H, 0x3c);
    bf16_t* fnorm = dalloc<bf16_t>(H, 0x3f);
    const size_t pool_layer = (size_t)B * max_pages * Hkv * 2 * 8192;
    bf16_t* pool = dalloc<bf16_t>(pool_layer * L, 0x3c);
    std::vector<int32_t> h_tab((size_t)R * max_pages), h_ctx(R, ctx);
    for (int b = 0; b < R; ++b) for (int p = 0; p < max_pages; ++p) h_tab[(size_t)b * max_pages + p] = (b % B) * max_pages + p;
    int32_t *tab = dalloc<int32_t>(h_tab.size()), *ctx_len = dalloc<int32_t>(R), *cur = dalloc<int32_t>(R), *out_ids = dalloc<int32_t>(R * 64),
            *out_lens = dalloc<int32_t>(R), *fin = dalloc<int32_t>(R), *am_idx = dalloc<int32_t>(R * 64);
    CK(hipMemcpy(tab, h_tab.data(), h_tab.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ctx_len, h_ctx.data(), (size_t)R * 4, hipMemcpyHostToDevice));
    float* am_val = dalloc<float>(R * 64);
    bf16_t *h0 = dalloc<bf16_t>((size_t)R * H, 0x3c), *h1 = dalloc<bf16_t>((size_t)R * H, 0x3c), *dq = dalloc<bf16_t>((size_t)R * Nq), *att = dalloc<bf16_t>((size_t)R * Nq), *act = dalloc<bf16_t>((size_t)R * I);
    float *slabs = dalloc<float>((size_t)4 * R * H), *po = dalloc<float>((size_t)R * Hq * 64 * 128), *pml = dalloc<float>((size_t)R * Hq * 64 * 2),
          *logits = dalloc<float>((size_t)R * V);
    std::vector<float> f(64);
    for (int i = 0; i < 64; ++i) f[i] = 1.0f / powf(1e6f, (float)(2 * i) / 128.0f);
    float* inv_freq = dalloc<float>(64);
    CK(hipMemcpy(inv_freq, f.data(), 256, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    const float scale = 1.0f / sqrtf(128.0f);
    StepState st{};
    st.cur_tokens = cur; st.ctx_len = ctx_len; st.out_ids = out_ids; st.out_lens = out_lens; st.finished = fin; st.eos_ids = nullptr; st.sel = nullptr;
    st.max_len = nullptr; st.n_eos = 0; st.out_stride = 64; st.cap = 1; st.advance_ctx = 0;       // cap 1: rows finish at once, ctx stays put

    const int full = getenv("DOTS_BENCH_FULL") ? 1 : 0;      // whole-tile projections (the half-chip launch plan)
    const int part_cus = full ? (getenv("DOTS_BENCH_CUS") ? atoi(getenv("DOTS_BENCH_CUS")) : 128) : 0;      // the partition plan caps gate|up's grid at what the partition holds
    printf("decode attention: %s\n", decode_attn_stream_wgs(B, Hkv, n_splits, max_pages, part_cus) ? "streaming kernel (one resident workgroup per CU)" : "one workgroup per (row, kv head, split)");
    bf16_t* xn = getenv("DOTS_BENCH_NO_XN") ? nullptr : dalloc<bf16_t>((size_t)64 * H);      // scratch of the round-6 four-tile kernels (decode_b64.hip)
    // DOTS_BENCH_SHARE_SMALL=1: every layer reads layer 0's qkv and o_proj weights (11 MB: resident in the XCDs' L2s after the first layer) while gate|up / down
    // stay distinct — the UPPER BOUND of any scheme that prefetches the small kernels' weights into L2 ahead of their launch (VERDICT r5 #2b)
    if (getenv("DOTS_BENCH_SHARE_SMALL")) { for (int i = 1; i < L; ++i) { qkv[i] = qkv[0]; o[i] = o[0]; } printf("qkv / o_proj weights shared by all layers (L2-resident)\n"); }
    if (getenv("DOTS_BENCH_SHARE_ALL")) { for (int i = 1; i < L; ++i) { qkv[i] = qkv[0]; o[i] = o[0]; w13[i] = w13[0]; down[i] = down[0]; } printf("all layer weights shared (Infinity-Cache-resident)\n"); }
    float* part_h = getenv("DOTS_BENCH_NO_XN") ? nullptr : dalloc<float>((size_t)DEC_KSPLIT_PARTS * 64 * H);      // K-quarter sums of the projections above 32 rows (decode_b64.hip)
    // (the pending residual update of the previous layer's K-half down_proj rides in this layer's norm launch, as in engine.hip)
    auto k_qkv = [&](int i) { CK(launch_dec_qkv(S, h0, ln1[i], qkv[i], wsc, bias[i], inv_freq, ctx_len, tab, max_pages, pool + pool_layer * i, dq, B, H, Hq, Hkv, eps, part_cus, xn,
                                                 part_h && dec_proj_ksplit_supports(B, H, I) ? part_h : nullptr, wsc)); };
    auto k_attn = [&](int i) { CK(launch_decode_attn(S, dq, pool + pool_layer * i, ctx_len, tab, max_pages, po, pml, B, Hq, Hkv, n_splits, scale, part_cus)); };
    auto k_comb = [&](int) { CK(launch_decode_attn_combine(S, po, pml, ctx_len, att, B, Hq, Hkv, n_splits)); };
    bool pend_o = false;
    auto k_o = [&](int i) { CK(launch_dec_proj(S, att, o[i], wsc, h0, B, H, Nq, part_cus, part_h, &pend_o)); };
    auto k_gu = [&](int i) { CK(launch_dec_gateup(S, h0, ln2[i], w13[i], wsc, act, B, H, I, eps, part_cus, xn, part_h && dec_proj_ksplit_supports(B, H, Nq) ? part_h : nullptr, wsc)); };
    bool pend = false;
    auto k_down = [&](int i) { CK(launch_dec_proj(S, act, down[i], wsc, h0, B, H, I, part_cus, part_h, &pend)); };
    auto k_lm = [&]() { CK(launch_dec_lmhead(S, h0, fnorm, lm, wsc, logits, B, H, V, eps, part_cus, xn, part_h && dec_proj_ksplit_supports(B, H, I) ? part_h : nullptr, wsc)); };
    // skip: bit mask of kernel kinds left out (marginal cost of a kind inside the real, HBM-cold step = full - skipped)
    auto step_skip = [&](int skip) {
        CK(launch_dec_embed(S, cur, embed, h0, B, H));
        for (int i = 0; i < L; ++i) {
            if (!(skip & 1)) k_qkv(i);
            if (!(skip & 2)) k_attn(i);
            if (!(skip & 4)) k_comb(i);
            if (!(skip & 8)) k_o(i);
            if (!(skip & 16)) k_gu(i);
            if (!(skip & 32)) k_down(i);
        }
        if (!(skip & 64)) k_lm();
        CK(launch_argmax_step(S, logits, V, V, B, am_val, am_idx, st));
    };
    auto step = [&]() { step_skip(0); };
    const double w_bytes = (double)wb * (L * ((double)NQKV * H + (double)Nq * H + 3.0 * H * I) + (double)V * H);
    const double kv_bytes = (double)B * (ctx + 1) * L * Hkv * 128 * 2 * 2;
    printf("decode_bench: B=%d ctx=%d max_seq_len=%d (n_splits %d); algorithmic bytes/step %.1f MB weights + %.1f MB KV\n", B, ctx, max_seq, n_splits,
           w_bytes / 1e6, kv_bytes / 1e6);
    if (argc > 4) {            // "once": a few plain replays of the whole step and nothing else (rocprofv3 --pmc runs: tools/pmc_decode.sh)
        const double t = time_graph(step, 2);
        printf("whole step %.1f us (3 replays under the profiler)\n", t);
        return 0;
    }
    const double t_step = time_graph(step);
    printf("whole step              %9.1f us   %.2f TB/s algorithmic (%.1f %% of 8 TB/s)\n", t_step, (w_bytes + kv_bytes) / t_step / 1e6,
           (w_bytes + kv_bytes) / t_step / 1e6 / 8 * 100);
    {
        const char* names[] = {"dec_qkv", "decode_attn", "decode_attn_combine", "dec_proj o", "dec_gateup", "dec_proj down", "dec_lmhead"};
        const double mbs[] = {wb * 1.0 * NQKV * H / 1e6, kv_bytes / L / 1e6, 0, wb * 1.0 * Nq * H / 1e6, wb * 2.0 * I * H / 1e6, wb * 1.0 * I * H / 1e6, wb * 1.0 * V * H / 1e6};
        double tot = 0;
        for (int k = 0; k < 7; ++k) {
            const double t = time_graph([&]() { step_skip(1 << k); });
            const double per = (t_step - t) / (k == 6 ? 1 : L);
            printf("  marginal %-22s %8.2f us / launch   %6.1f MB   %.2f TB/s\n", names[k], per, mbs[k], mbs[k] ? mbs[k] / per : 0.0);
            if (k < 6) tot += per;
        }
        printf("  sum of per-layer marginals %.2f us (step / 28 = %.2f us)\n", tot, t_step / L);
    }
    struct Row { const char* name; std::function<void(int)> fn; double mb; };
    const Row rows[] = {
        {"dec_qkv", k_qkv, wb * 1.0 * NQKV * H / 1e6},
        {"decode_attn", k_attn, kv_bytes / L / 1e6},
        {"decode_attn_combine", k_comb, 0},
        {"dec_proj o", k_o, wb * 1.0 * Nq * H / 1e6},
        {"dec_gateup", k_gu, wb * 2.0 * I * H / 1e6},
        {"dec_proj down", k_down, wb * 1.0 * I * H / 1e6},
    };
    double sum = 0;
    for (const Row& r : rows) {
        const double us = time_graph([&]() { for (int i = 0; i < L; ++i) r.fn(i); }) / L;
        printf("%-24s %8.2f us / launch   %6.1f MB   %.2f TB/s\n", r.name, us, r.mb, r.mb ? r.mb / us : 0.0);
        trace_report(r.name, 8);
    }
    const double lm_us = time_graph([&]() { for (int i = 0; i < 4; ++i) k_lm(); }) / 4;
    printf("%-24s %8.2f us / launch   %6.1f MB   %.2f TB/s\n", "dec_lmhead", lm_us, wb * 1.0 * V * H / 1e6, wb * 1.0 * V * H / 1e6 / lm_us);
    const double misc = time_graph([&]() { for (int i = 0; i < 8; ++i) { CK(launch_dec_embed(S, cur, embed, h0, B, H)); CK(launch_argmax_step(S, logits, V, V, B, am_val, am_idx, st)); } }) / 8;
    printf("%-24s %8.2f us (embed + argmax partial + argmax step)\n", "step glue", misc);
    return 0;
}
