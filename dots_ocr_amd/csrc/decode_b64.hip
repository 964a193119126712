// Round 6: the wide-N dense kernels of the decode step for batches above 32 rows (the 64-row step of the pipelined bench): gate|up and lm_head
// with ALL FOUR 16-row batch tiles in one workgroup, so that every weight byte crosses a CU's load path ONCE (round 4's two-tile kernels
// pulled W13 and the lm_head through the L2s twice at 64 rows: PMC traffic 1.24 x algorithmic, profiles/r05_decode_traffic_64rows.json).
//
// What stood in the way: the normalised activations of 64 rows are 64 x 1536 bf16 = 192 KiB, more than a CU's 160 KiB of LDS, and the
// round-4 structure (waves = K slices running side by side) needs all of them at once.  Here
//   * a small kernel normalises the residual rows ONCE per GEMM into the X image in global memory (dec_norm_ximg_kernel: the per-tile
//     prologue's own row_rstd / norm8, so the bits are theirs); it stays in L2;
//   * a wave owns a PAIR of 16-row weight tiles (gate tile + up tile / two vocabulary tiles) for all 64 batch rows and walks the whole K
//     in order; the X image passes through LDS in K chunks of one gate|up slice (H / 128 k-steps x 4 tiles = 48 KiB at H = 1536), a
//     ring of three buffers filled by LDS-DMA two chunks ahead (no registers, no VALU), one raw s_barrier per chunk;
//   * the weights stream straight into a register ring D k-steps deep per wave (2 fragments per k-step, non-temporal), refilled right
//     behind the MFMAs that consumed a slot and never drained: the ring runs across the chunk barriers and across a wave's work items;
//   * per output element the arithmetic is the per-tile kernels' exactly — gate|up: four K slices, each an MFMA chain in k order from
//     zero, the slice sums added in order (dec_gateup_kernel: ag + red[1] + red[2] + red[3]); lm_head: even k-steps on one accumulator, odd
//     ones on another, their sum (dec_lmhead_kernel) — so a row's bits do not depend on the batch it shares
//     (tests/test_decode_kernels_gpu.py: the 40- / 64-row calls equal calls of <= 16 rows bit for bit).
// Every wave of a workgroup runs the same instruction stream (a wave without a work item streams a chunk of zeros: its MFMAs add +0),
// so the counted `s_waitcnt vmcnt` in front of each barrier is exact: the youngest 2 D memory operations are always ring loads.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "decode_layout.h"
#include "kernels.h"

namespace {
TRACE_DECL
#include "decode_dev.h"

constexpr int S64_TT = 4;        // batch tiles per workgroup (64 rows)
constexpr int S64_NBUF = 3;      // X chunk ring
enum { S64_GATEUP = 0, S64_LMHEAD = 1 };

// rows [B, H] -> rmsnorm -> X image [ceil(B / 16)][H / 8][16][8] in global memory.  One wave per row; bits of dec_*_kernel's prologue.
__global__ __launch_bounds__(256) void dec_norm_ximg_kernel(const bf16_t* __restrict__ h, const bf16_t* __restrict__ ln_w, bf16_t* __restrict__ xn,
                                                            int B, int H, float eps) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave_id();                                     // wave-uniform
    Rows<1, NC_MAX> R;
    rows_issue<1, NC_MAX>(R, h, ln_w, B, H, r, 0, lane);
    if (r < B) row_norm_to_lds<NC_MAX>(R.v[0], R.w, r & 15, H, eps, xn + (size_t)(r >> 4) * 16 * H, 16, lane);
}

// SwiGLU of one lane's 4 features of a (gate, up) accumulator pair -> X image.  The expression of dec_gateup_kernel's epilogue.
DEVI void swiglu_store4(bf16_t* __restrict__ act_tile, int m, int col, f32x4 gs, f32x4 us) {
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = gs[r] / (1.0f + __expf(-gs[r])) * us[r];
    store_frag4(act_tile, m, col, 16, o[0], o[1], o[2], o[3]);
}

// grid.x workgroups x NWV waves.  Work items (gate|up: the I / 16 tile pairs of the packed W13; lm_head: pairs of vocabulary tiles) are
// split evenly over the workgroups (contiguous ranges); wave w of a workgroup takes items w, w + NWV, ... of its range.
template <int MODE, typename WT, int NWV, int D, int L>
__global__ __launch_bounds__(NWV * 64) void dec_stream64_kernel(const bf16_t* __restrict__ Xn, const WT* __restrict__ Wd, const float* __restrict__ wscale,
                                                                void* __restrict__ out, int B, int K, int N, int n_items) {
    // L = k-steps per K chunk = K / 128 (compile time: every slot's offsets are immediates / scalar adds; 12 = dots.ocr, 6 = the tests' small model)
    static_assert((4 * L) % D == 0, "the ring position of a slot must not depend on the round");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int Q = 4 * L;                                                      // slots (k-steps) per round
    constexpr int PW = (S64_TT * L + NWV - 1) / NWV;                              // DMA pieces per wave and chunk
    const int lane = threadIdx.x & 63, wv = wave_id();
    const int m = lane & 15, g = lane >> 4;
    constexpr int KS = 4 * L;                                                     // K == 128 L (launcher)
    const int n_bt = (B + 15) >> 4;                                               // batch tiles that exist (3 or 4)
    constexpr int chunk_bytes = S64_TT * L * 1024;
    const int i0 = (int)((long long)blockIdx.x * n_items / gridDim.x), i1 = (int)((long long)(blockIdx.x + 1) * n_items / gridDim.x);
    const int rounds = max(1, (i1 - i0 + NWV - 1) / NWV), n_seq = 4 * rounds;
    const int n_wt = N / 16;                                                      // weight tiles (lm_head)
    const int ls = lane_slot<WT>(g, m);
    const WT* zc = reinterpret_cast<const WT*>(g_zero_chunk);

    // ---- X chunk c of the image -> ring buffer `buf`; piece = (batch tile, k-step) = 1 KiB.  A wave's PW pieces have fixed (tile, k-step
    // inside the chunk): wave-uniform 64-bit sources computed once, the chunk is a compile-time byte offset on the lane's 32-bit offset.
    const bf16_t* psrc[PW];
    int pdst[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int p = (wv + i * NWV) % (S64_TT * L);                              // fewer pieces than slots: one is copied again (same bytes)
        const int t = p / L, j = p - t * L;
        psrc[i] = Xn + (size_t)min(t, n_bt - 1) * 16 * K + (size_t)j * 512;
        pdst[i] = p * 1024;
    }
    const uint32_t smem_base = (uint32_t)(size_t)smem;
    auto dma = [&](int buf, int c) {
        // Inline asm, not __builtin_amdgcn_global_load_lds: the compiler cannot tell the ring buffer being filled from the one being read and
        // answers every LDS-DMA builtin with `s_waitcnt vmcnt(0)` in front of the next ds_read — which drains the weight ring once per chunk.
        // (It does not count these operations in its own vmcnt arithmetic for the ring loads: its waits are then at most PW operations
        // stricter than needed, never laxer.)  LDS destination (wave-uniform; the hardware adds 16 B per lane) through M0, which nothing else in
        // this kernel uses (gfx950 DS instructions do not need it; M0 is reserved, so it is not on the clobber list).
        const uint32_t voff = lane * 16u + (uint32_t)c * (L * 1024);
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const uint32_t lds_addr = smem_base + (uint32_t)(buf * chunk_bytes + pdst[i]);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(psrc[i]) : "memory");
        }
    };
    // ---- the two weight tiles of an item; a wave's weight stream is ONE sequence of k-steps (item after item) read through two running
    // wave-uniform pointers, `stride` 64 fragments per k-step — or the chunk of zeros with stride 0 when the wave has no (further) item
    auto item_of = [&](int r) { const int it = i0 + wv + r * NWV; return it < i1 ? it : -1; };
    auto tile_a = [&](int it) { return MODE == S64_GATEUP ? (it >> 1) * 4 + (it & 1) : 2 * it; };
    auto tile_b = [&](int it) { return MODE == S64_GATEUP ? (it >> 1) * 4 + 2 + (it & 1) : min(2 * it + 1, n_wt - 1); };
    auto base_of = [&](int tile) { return Wd + (size_t)tile * KS * 64; };

    TRACE(0);
    dma(0, 0);
    dma(1, 1);
    int it = item_of(0);
    const WT* fa = it >= 0 ? base_of(tile_a(it)) : zc;
    const WT* fb = it >= 0 ? base_of(tile_b(it)) : zc;
    int stride = it >= 0 ? 64 : 0;
    WT ra[D], rb[D];
#pragma unroll
    for (int q = 0; q < D; ++q) {
        if (q == Q) {                                                             // (D == Q: the ring holds a whole item)
            const int i1n = rounds > 1 ? item_of(1) : -1;
            fa = i1n >= 0 ? base_of(tile_a(i1n)) : zc; fb = i1n >= 0 ? base_of(tile_b(i1n)) : zc; stride = i1n >= 0 ? 64 : 0;
        }
        ra[q] = __builtin_nontemporal_load(fa + ls);
        rb[q] = __builtin_nontemporal_load(fb + ls);
        fa += stride; fb += stride;
    }
    __builtin_amdgcn_sched_barrier(0);

    for (int r = 0; r < rounds; ++r) {
        const int it_n = r + 1 < rounds ? item_of(r + 1) : -1;
        // fp8: per-output-channel scales of this item's rows 4g .. 4g + 3 of either tile (small operands, back long before the epilogue)
        f32x4 sca = {1.f, 1.f, 1.f, 1.f}, scb = {1.f, 1.f, 1.f, 1.f};
        if constexpr (is_fp8<WT>::value) {
            const int ita = max(it, 0);
            sca = *reinterpret_cast<const f32x4*>(wscale + tile_a(ita) * 16 + 4 * g);
            scb = *reinterpret_cast<const f32x4*>(wscale + tile_b(ita) * 16 + 4 * g);
        }
        // accumulators: x0 / x1 = the chain(s) being built, y0 / y1 = gate|up: the running sum of finished slices; lm_head: the odd k-steps
        f32x4 x0[S64_TT], x1[S64_TT], y0[S64_TT], y1[S64_TT];
#pragma unroll
        for (int t = 0; t < S64_TT; ++t) { x0[t] = f32x4{0, 0, 0, 0}; x1[t] = f32x4{0, 0, 0, 0}; y0[t] = f32x4{0, 0, 0, 0}; y1[t] = f32x4{0, 0, 0, 0}; }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int seq = 4 * r + c;
            // own pieces of chunk seq (and seq + 1) are in: the youngest 2 D operations are ring loads, all younger than those DMAs
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * D) : "memory");
            __builtin_amdgcn_s_barrier();                                         // everybody's pieces of chunk seq are in; everybody is done with chunk seq - 1
            TRACE(1 + c);
            dma((seq + 2) % S64_NBUF, (c + 2) & 3);                               // into the buffer of chunk seq - 1 (past the last chunk: a copy nobody reads)
            const bf16x8* xb = reinterpret_cast<const bf16x8*>(smem + (seq % S64_NBUF) * chunk_bytes) + lane;
            // B fragments: NBX = 2: one slot ahead of their MFMAs (two register sets); NBX = 1 (the 12-wave plan: 168 registers per lane, and
            // three waves per SIMD to cover each other's LDS latency): read when needed.  The scheduling barriers keep hipcc from hoisting
            // further (it did, and spilled).
            constexpr int NBX = NWV >= 12 ? 1 : 2;
            bf16x8 bx[NBX][S64_TT];
            if constexpr (NBX == 2) {
#pragma unroll
                for (int t = 0; t < S64_TT; ++t) bx[0][t] = xb[(size_t)(t * L) * 64];
            }
#pragma unroll
            for (int j = 0; j < L; ++j) {
                const int q = c * L + j;                                          // compile-time after unrolling
                if constexpr (NBX == 2) {
                    if (j + 1 < L) {
#pragma unroll
                        for (int t = 0; t < S64_TT; ++t) bx[(j + 1) & 1][t] = xb[(size_t)(t * L + j + 1) * 64];
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < S64_TT; ++t) bx[0][t] = xb[(size_t)(t * L + j) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 wa = as_a(ra[q % D]), wb = as_a(rb[q % D]);
                // gate|up: slice 0 builds its chain in y (the running sum to be), slices 1-3 in x; lm_head: even k-steps in x, odd ones in y
                if ((MODE == S64_GATEUP && c > 0) || (MODE == S64_LMHEAD && (j & 1) == 0)) {
#pragma unroll
                    for (int t = 0; t < S64_TT; ++t) {
                        x0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, bx[j & (NBX - 1)][t], x0[t], 0, 0, 0);
                        x1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, bx[j & (NBX - 1)][t], x1[t], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < S64_TT; ++t) {
                        y0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, bx[j & (NBX - 1)][t], y0[t], 0, 0, 0);
                        y1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, bx[j & (NBX - 1)][t], y1[t], 0, 0, 0);
                    }
                }
                // refill the slot: D k-steps ahead in the wave's stream (the next item's first k-steps near the end of this one)
                if (q + D == Q) {
                    fa = it_n >= 0 ? base_of(tile_a(it_n)) : zc; fb = it_n >= 0 ? base_of(tile_b(it_n)) : zc; stride = it_n >= 0 ? 64 : 0;
                }
                ra[q % D] = __builtin_nontemporal_load(fa + ls);
                rb[q % D] = __builtin_nontemporal_load(fb + ls);
                fa += stride; fb += stride;
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (MODE == S64_GATEUP) {                                   // slice c is complete: gs = ((s0 + s1) + s2) + s3
                if (c > 0) {
#pragma unroll
                    for (int t = 0; t < S64_TT; ++t) {
                        y0[t] += x0[t]; y1[t] += x1[t];
                        // anchored here: LLVM otherwise sinks these adds into the (conditional) epilogue, keeps every slice's chain alive until
                        // then and spills them
                        asm volatile("" : "+v"(y0[t]), "+v"(y1[t]));
                        x0[t] = f32x4{0, 0, 0, 0}; x1[t] = f32x4{0, 0, 0, 0};
                    }
                }
            }
        }
        TRACE(5);
        // ---- epilogue of this item
        if (it >= 0) {
            if constexpr (MODE == S64_GATEUP) {
                bf16_t* act = reinterpret_cast<bf16_t*>(out);
                const int col = (it >> 1) * 32 + (it & 1) * 16 + 4 * g;
#pragma unroll
                for (int t = 0; t < S64_TT; ++t) {
                    if (m + 16 * t < B) {
                        f32x4 gs = y0[t], us = y1[t];
                        if constexpr (is_fp8<WT>::value) { gs *= sca; us *= scb; }
                        swiglu_store4(act + (size_t)t * 16 * N, m, col, gs, us);
                    }
                }
            } else {
                // wave-uniform 64-bit part (batch tile, vocabulary tile) + one 32-bit lane offset (row m, features 4g ..): nothing but that offset
                // stays in vector registers across the K loop
                float* logits = reinterpret_cast<float*>(out);
                const int ta = tile_a(it), tb = 2 * it + 1;
                const uint32_t lane_off = (uint32_t)m * (uint32_t)N + 4u * g;
#pragma unroll
                for (int t = 0; t < S64_TT; ++t) {
                    f32x4 a0 = x0[t] + y0[t], a1 = x1[t] + y1[t];
                    if constexpr (is_fp8<WT>::value) { a0 *= sca; a1 *= scb; }
                    if (m + 16 * t < B) {
                        float* lp = logits + (size_t)16 * t * N + (size_t)ta * 16;
                        *reinterpret_cast<f32x4*>(lp + lane_off) = a0;
                        if (tb < n_wt) *reinterpret_cast<f32x4*>(lp + 16 + lane_off) = a1;
                    }
                }
            }
        }
        it = it_n;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                              // the two copies past the last chunk: nothing may land in LDS after the workgroup ends
    TRACE(6);
}

template <typename Kern>
hipError_t ensure_lds64(Kern kern, size_t bytes, uint32_t* done_mask) {
    if (bytes <= 64 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint32_t bit = 1u << (dev & 31);
    if (__atomic_load_n(done_mask, __ATOMIC_ACQUIRE) & bit) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    __atomic_fetch_or(done_mask, bit, __ATOMIC_RELEASE);
    return hipSuccess;
}

int device_cus() {
    static int n_cus = 0;
    if (n_cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) { (void)hipGetLastError(); v = 256; }
        n_cus = v;
    }
    return n_cus;
}

template <int MODE, typename WT>
hipError_t stream64_launch(hipStream_t s, const bf16_t* xn, const WT* Wd, const float* wscale, void* out, int B, int K, int N, int n_items, int cus) {
    static uint32_t attr[6] = {0, 0, 0, 0, 0, 0};
    const int L = K / 128;
    const size_t lds = (size_t)S64_NBUF * S64_TT * L * 1024;
    const int G = std::max(1, std::min(n_items, cus));
    const int per_wg = (n_items + G - 1) / G;
    auto go = [&](auto kern, int nwv, uint32_t* done) -> hipError_t {
        hipError_t e = ensure_lds64(kern, lds, done);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(G), dim3(nwv * 64), lds, s, xn, Wd, wscale, out, B, K, N, n_items);
        return hipGetLastError();
    };
    // few items per workgroup (the whole chip): few waves with a deep ring each; many (a CU partition; the lm_head): 12 waves
    static const int force = [] { const char* e = getenv("DOTS_OCR_S64_WAVES"); return e ? atoi(e) : 0; }();      // A/B switch: 4 / 8 / 12
    const int nwv = force ? force : (per_wg <= 4 ? 4 : per_wg <= 8 ? 8 : 12);
    if (L == 12) {
        if (nwv == 4) return go(dec_stream64_kernel<MODE, WT, 4, 24, 12>, 4, &attr[0]);
        if (nwv == 8) return go(dec_stream64_kernel<MODE, WT, 8, 12, 12>, 8, &attr[1]);
        if constexpr (MODE == S64_GATEUP) return go(dec_stream64_kernel<MODE, WT, 12, 8, 12>, 12, &attr[2]);
        else return go(dec_stream64_kernel<MODE, WT, 12, 6, 12>, 12, &attr[2]);
    }
    if (nwv == 4) return go(dec_stream64_kernel<MODE, WT, 4, 24, 6>, 4, &attr[3]);
    if (nwv == 8) return go(dec_stream64_kernel<MODE, WT, 8, 12, 6>, 8, &attr[4]);
    return go(dec_stream64_kernel<MODE, WT, 12, 6, 6>, 12, &attr[5]);
}

}  // namespace

#ifdef DOTS_TRACE
void dots_trace_set_b64(unsigned long long* buf) { (void)hipMemcpyToSymbol(HIP_SYMBOL(dots_trace_buf), &buf, sizeof(buf)); }
#endif

bool dec_stream64_supports(int B, int H) {
    static const bool off = getenv("DOTS_OCR_DEC_S64") && atoi(getenv("DOTS_OCR_DEC_S64")) == 0;      // A/B switch: the round-4 two-tile kernels
    return !off && B > 32 && B <= MAX_DECODE_ROWS && (H == 128 * 12 || H == 128 * 6);       // K chunk = 12 k-steps (dots.ocr) or 6 (the tests' small model)
}

hipError_t launch_dec_norm_ximg(hipStream_t s, const bf16_t* h, const bf16_t* ln_w, bf16_t* xn, int B, int H, float eps) {
    if (H % 8 || H > 512 * NC_MAX || B < 1 || B > MAX_DECODE_ROWS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(dec_norm_ximg_kernel, dim3((B + 3) / 4), dim3(256), 0, s, h, ln_w, xn, B, H, eps);
    return hipGetLastError();
}

// act = silu(gate) * up of rmsnorm(h), B in (32, 64]: norm kernel -> xn (scratch: 64 x H bf16), then the streaming kernel.  cus: CUs the stream may use.
hipError_t launch_dec_gateup64(hipStream_t s, const bf16_t* h, const bf16_t* ln_w, const void* W13d, const float* wscale, bf16_t* act, bf16_t* xn,
                               int B, int H, int I, float eps, int cus) {
    if (!dec_stream64_supports(B, H) || I % 32 || !xn) return hipErrorInvalidValue;
    HIP_CHECK_RET(launch_dec_norm_ximg(s, h, ln_w, xn, B, H, eps));
    if (cus <= 0) cus = device_cus();
    return wscale ? stream64_launch<S64_GATEUP>(s, xn, (const u32x2*)W13d, wscale, act, B, H, I, I / 16, cus)
                  : stream64_launch<S64_GATEUP>(s, xn, (const bf16x8*)W13d, wscale, act, B, H, I, I / 16, cus);
}

hipError_t launch_dec_lmhead64(hipStream_t s, const bf16_t* h, const bf16_t* ln_w, const void* Wd, const float* wscale, float* logits, bf16_t* xn,
                               int B, int H, int V, float eps, int cus) {
    if (!dec_stream64_supports(B, H) || V % 16 || !xn) return hipErrorInvalidValue;
    HIP_CHECK_RET(launch_dec_norm_ximg(s, h, ln_w, xn, B, H, eps));
    if (cus <= 0) cus = device_cus();
    const int n_items = (V / 16 + 1) / 2;
    return wscale ? stream64_launch<S64_LMHEAD>(s, xn, (const u32x2*)Wd, wscale, logits, B, H, V, n_items, cus)
                  : stream64_launch<S64_LMHEAD>(s, xn, (const bf16x8*)Wd, wscale, logits, B, H, V, n_items, cus);
}
