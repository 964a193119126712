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
// so the counted `s_waitcnt vmcnt` in front of each barrier is exact: what was issued behind a chunk's DMA pieces is known at compile time.
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
// PART (round 6, the K-split projections below): the rows are not final yet — the projection left its four K-quarter sums in `part` ([4][64][H] fp32) instead of
// adding them to the residual stream.  This kernel does that first, with proj_epilogue's own arithmetic ((q0 + q1) + (q2 + q3), x the fp8 weight scale, + residual, one
// rounding to bf16), writes the rows back to h and normalises what it wrote.  xn == nullptr: only the residual update (the single-kernel test entry).
template <bool PART>
__global__ __launch_bounds__(256) void dec_norm_ximg_kernel(bf16_t* __restrict__ h, const bf16_t* __restrict__ ln_w, bf16_t* __restrict__ xn,
                                                            const float* __restrict__ part, const float* __restrict__ pscale, int B, int H, float eps) {
#pragma clang fp contract(off)      // the residual update must round exactly like proj_epilogue (decode_fused.hip)
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave_id();                                     // wave-uniform
    Rows<1, NC_MAX> R;
    rows_issue<1, NC_MAX>(R, h, ln_w ? ln_w : h, B, H, r, 0, lane);
    if constexpr (PART) {
        if (r < B) {
#pragma unroll
            for (int c = 0; c < NC_MAX; ++c) {
                const int k = c * 512 + lane * 8;
                if (k < H) {
                    f32x4 q0[DEC_KSPLIT_PARTS], q1[DEC_KSPLIT_PARTS];
#pragma unroll
                    for (int i = 0; i < DEC_KSPLIT_PARTS; ++i) {
                        const float* pp = part + ((size_t)i * MAX_DECODE_ROWS + r) * H + k;
                        q0[i] = *reinterpret_cast<const f32x4*>(pp);
                        q1[i] = *reinterpret_cast<const f32x4*>(pp + 4);
                    }
                    f32x4 s0 = (q0[0] + q0[1]) + (q0[2] + q0[3]), s1 = (q1[0] + q1[1]) + (q1[2] + q1[3]);      // proj_sum16's last three additions
                    if (pscale) {
                        s0 *= *reinterpret_cast<const f32x4*>(pscale + k);
                        s1 *= *reinterpret_cast<const f32x4*>(pscale + k + 4);
                    }
                    const u32x4 res = R.v[0][c];
                    const u32x4 o = {pack_bf2(lo_bf(res[0]) + s0[0], hi_bf(res[0]) + s0[1]), pack_bf2(lo_bf(res[1]) + s0[2], hi_bf(res[1]) + s0[3]),
                                     pack_bf2(lo_bf(res[2]) + s1[0], hi_bf(res[2]) + s1[1]), pack_bf2(lo_bf(res[3]) + s1[2], hi_bf(res[3]) + s1[3])};
                    R.v[0][c] = o;
                    *reinterpret_cast<u32x4*>(h + (size_t)r * H + k) = o;
                }
            }
        }
    }
    if (xn && r < B) row_norm_to_lds<NC_MAX>(R.v[0], R.w, r & 15, H, eps, xn + (size_t)(r >> 4) * 16 * H, 16, lane);
}

// ------------------------------------------------------------------------------------------------
// The projections above 32 rows as FOUR K quarters (round 6).  dec_proj_wide_kernel (decode_fused.hip) gives a workgroup a few output features over the FULL
// K, so every workgroup reads the WHOLE X image — for down_proj 64 x 8960 x 2 B = 1.15 MB beside its 0.43 MB of weights on the 64-CU partition, beside
// 143 KB on the whole chip — and a CU pulls ~50 GB/s whatever the source (profiles/r06_decode_trace_b64.txt): the kernel sat on that line.  Here a workgroup
// owns ONE quarter of K (slices 4q .. 4q + 3 of the same 16 slices, same boundaries) for four times the features: a quarter of the X bytes per workgroup, the
// same weight bytes.  Its 4 waves — one slice each, ALL features of the workgroup: NM MFMAs of 16 weight rows x 4 batch tiles per k-step, one wave per SIMD
// and its 512 registers — reduce their slice sums in order through LDS and store the quarter's fp32 sum to part[q][row][col]; the consumer of the residual
// stream — always a norm kernel at these batch sizes — adds (q0 + q1) + (q2 + q3), the fp8 scale and the residual exactly as proj_epilogue does
// (dec_norm_ximg_kernel<true>).  Every projection kernel adds its 16 slice sums in that order since this round (proj_sum16), so a row's bits do not depend on
// which kernel ran.
constexpr int KS_G = 4;              // k-steps per round: KS_G x (NM weight + 4 activation fragments) requested together
constexpr int KS_WAVES = 16 / DEC_KSPLIT_PARTS;
template <int NM, typename WT>
__global__ __launch_bounds__(KS_WAVES * 64) void dec_proj_ksplit_kernel(const bf16_t* __restrict__ X, const WT* __restrict__ Wd, float* __restrict__ part,
                                                                        int B, int N, int K, int NU) {
    constexpr int TT = S64_TT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* red = reinterpret_cast<f32x4*>(smem);                                  // [KS_WAVES slices][NM][TT][64]
    const int lane = threadIdx.x & 63, wv = wave_id();
    const int m = lane & 15, g = lane >> 4;
    const int quarter = blockIdx.x, n_tiles = (B + 15) >> 4, n_units = N >> 3;
    const int u0 = blockIdx.y * NU;
    const int KS = K / 32, sl = KS_WAVES * quarter + wv;
    const int k0 = (int)((uint32_t)(sl * KS) >> 4), k1 = (int)((uint32_t)((sl + 1) * KS) >> 4);       // the 16-slice boundaries of dec_proj_kernel
    uint32_t woff[NM];
#pragma unroll
    for (int j = 0; j < NM; ++j) {
        const int ua = min(u0 + 2 * j, n_units - 1);
        const int ub = (2 * j + 1 < NU && u0 + 2 * j + 1 < n_units) ? u0 + 2 * j + 1 : ua;
        const int u = (m >> 3) ? ub : ua;
        woff[j] = (uint32_t)(((size_t)(u >> 1) * KS * 64 + lane_slot<WT>(g, (m & 7) + 8 * (u & 1))) * sizeof(WT));
    }
    const char* wbase = reinterpret_cast<const char*>(Wd);
    const char* zbase = reinterpret_cast<const char*>(g_zero_chunk);
    const uint32_t zoff = lane * (uint32_t)sizeof(WT);
    const char* xbase = reinterpret_cast<const char*>(X);
    const uint32_t xoff = lane * 16u;
    uint32_t toff[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) toff[t] = (uint32_t)min(t, n_tiles - 1) * 32u * (uint32_t)K;
    TRACE(0);
#pragma unroll
    for (int par = 0; par < 2; ++par) {                                           // the even chain, then the odd chain (dec_proj_kernel: acc0 / acc1)
        f32x4 acc[NM][TT];
#pragma unroll
        for (int j = 0; j < NM; ++j)
#pragma unroll
            for (int t = 0; t < TT; ++t) acc[j][t] = f32x4{0, 0, 0, 0};
        for (int kb = k0 + par; kb < k1; kb += 2 * KS_G) {                        // wave-uniform trip count
            WT a[NM][KS_G];
            bf16x8 b[TT][KS_G];
#pragma unroll
            for (int jj = 0; jj < KS_G; ++jj) {
                const int k = kb + 2 * jj;
                const bool ok = k < k1;                                           // slots past the slice multiply a chunk of zeros
                const int kc = min(k, KS - 1);
                const char* wk = ok ? wbase + (size_t)k * 64 * sizeof(WT) : zbase;
#pragma unroll
                for (int j = 0; j < NM; ++j) a[j][jj] = __builtin_nontemporal_load(reinterpret_cast<const WT*>(wk + (ok ? woff[j] : zoff)));
#pragma unroll
                for (int t = 0; t < TT; ++t) b[t][jj] = *reinterpret_cast<const bf16x8*>(xbase + (size_t)(toff[t] + (uint32_t)kc * 1024u) + xoff);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < KS_G; ++jj)
#pragma unroll
                for (int j = 0; j < NM; ++j) {
                    const bf16x8 wa = as_a(a[j][jj]);
#pragma unroll
                    for (int t = 0; t < TT; ++t) acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, b[t][jj], acc[j][t], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < NM; ++j)
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                f32x4* rp = red + ((size_t)((wv * NM + j) * TT + t)) * 64 + lane;
                if (par == 0) *rp = acc[j][t];
                else *rp = *rp + acc[j][t];                                       // even + odd
            }
    }
    TRACE(1);
    __syncthreads();
    TRACE(2);
    // accumulator tile e = (MFMA je, batch tile te): wave wv finishes tiles wv, wv + KS_WAVES, ...
#pragma unroll
    for (int i = 0; i < (NM * TT + KS_WAVES - 1) / KS_WAVES; ++i) {
        const int e = wv + KS_WAVES * i;
        if (e >= NM * TT) break;                                                  // wave-uniform
        const int je = e / TT, te = e % TT;
        const int ul = 2 * je + (g >> 1), unit = u0 + ul, row = 16 * te + m;
        if (ul < NU && unit < n_units && row < B) {
            f32x4 p = {0, 0, 0, 0};
#pragma unroll
            for (int s4 = 0; s4 < KS_WAVES; ++s4) p += red[((size_t)((s4 * NM + je) * TT + te)) * 64 + lane];      // this quarter of proj_sum16
            *reinterpret_cast<f32x4*>(part + ((size_t)quarter * MAX_DECODE_ROWS + row) * N + 8 * unit + 4 * (g & 1)) = p;
        }
    }
    TRACE(3);
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
    dma(0, 0);                                                                    // chunk 0 first, then the ring's first D k-steps, then chunk 1: the first barrier
    int it = item_of(0);                                                          // waits for 48 KiB of X, not for 96 (whole chip: it opened 6 us into the kernel)
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
    dma(1, 1);
    // Counted wait in front of every chunk barrier.  Loads return in order, so "at most N operations outstanding" means every load but the N youngest is
    // back: N must not exceed the number of LOADS issued after the last piece of the chunk that is waited for (stores in flight only make the wait longer):
    //   chunk 0 of round 0: issued first; behind it the 2 D ring loads and chunk 1's PW pieces;
    //   chunk 1 of round 0: behind it chunk 2's pieces (issued at the first barrier) and the 2 L refills of the first chunk phase  <- the smallest count;
    //   every later chunk s: issued at the barrier of chunk s - 2; behind it 2 L refills, chunk s + 1's pieces, 2 L refills.
    // One constant for all of them (no branches in the stream): min(2 D, PW + 2 L) — 2 D, i.e. the whole ring stays in flight, for the 8- and 12-wave plans.
    // (Round 6 review: with L = 6 and a 24-deep ring the former vmcnt(2 D) allowed MORE operations in flight than had been issued behind chunk s — it did
    // not wait for the chunk at all; the small-model tests passed on timing.)
    constexpr int WAIT_N = 2 * D < PW + 2 * L ? 2 * D : PW + 2 * L;

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
            // own pieces of chunk seq are in
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_N) : "memory");
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

hipError_t launch_dec_norm_ximg(hipStream_t s, const bf16_t* h_in, const bf16_t* ln_w, bf16_t* xn, int B, int H, float eps, const float* part, const float* pscale) {
    bf16_t* h = const_cast<bf16_t*>(h_in);                    // written only when `part` is given (the pending residual update of a K-half projection)
    if (H % 8 || H > 512 * NC_MAX || B < 1 || B > MAX_DECODE_ROWS || (!xn && !part)) return hipErrorInvalidValue;
    if (part) hipLaunchKernelGGL(dec_norm_ximg_kernel<true>, dim3((B + 3) / 4), dim3(256), 0, s, h, ln_w, xn, part, pscale, B, H, eps);
    else hipLaunchKernelGGL(dec_norm_ximg_kernel<false>, dim3((B + 3) / 4), dim3(256), 0, s, h, ln_w, xn, part, pscale, B, H, eps);
    return hipGetLastError();
}

// the projections above 32 rows as four K quarters: part[4][64][N] <- the quarters' slice sums (h is NOT updated: launch_dec_norm_ximg(.., part, wscale) does that).
bool dec_proj_ksplit_supports(int B, int N, int K) {
    const char* env = getenv("DOTS_OCR_DEC_KSPLIT");                // A/B switch, read per call (tests flip it): 0 = dec_proj_wide_kernel over the full K,
    const int mode = env ? atoi(env) : 1;                           // 1 = long K only (down_proj), 2 = o_proj too
    const int min_ks = mode >= 2 ? 16 : 64;
    return mode > 0 && B > 32 && B <= MAX_DECODE_ROWS && N % 16 == 0 && N <= 512 * NC_MAX && K % 32 == 0 && K / 32 >= min_ks;      // N: what the norm kernel's rows hold
}

hipError_t launch_dec_proj_ksplit(hipStream_t s, const bf16_t* X, const void* Wd, bool fp8, float* part, int B, int N, int K, int cus) {
    if (!dec_proj_ksplit_supports(B, N, K) || !part) return hipErrorInvalidValue;
    if (cus <= 0) cus = device_cus();
    static uint32_t attr[12] = {0};
    const int n_units = N / 8;
    // 4 K quarters x ceil(n_units / nu) workgroups in ONE round on the CUs the stream may use; a workgroup takes up to 12 units (6 MFMAs of 16 weight rows)
    const int nu = std::max(1, std::min(12, (DEC_KSPLIT_PARTS * n_units + cus - 1) / cus)), nm = (nu + 1) / 2;
    const size_t lds = (size_t)KS_WAVES * nm * S64_TT * 64 * sizeof(f32x4);
    const dim3 grid(DEC_KSPLIT_PARTS, (n_units + nu - 1) / nu);
    auto go = [&](auto kern, auto wd, uint32_t* done) -> hipError_t {
        hipError_t e = ensure_lds64(kern, lds, done);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(KS_WAVES * 64), lds, s, X, wd, part, B, N, K, nu);
        return hipGetLastError();
    };
#define KSPLIT(NMV) if (nm == NMV) return fp8 ? go(dec_proj_ksplit_kernel<NMV, u32x2>, (const u32x2*)Wd, &attr[2 * (NMV - 1)]) : go(dec_proj_ksplit_kernel<NMV, bf16x8>, (const bf16x8*)Wd, &attr[2 * (NMV - 1) + 1])
    KSPLIT(1); KSPLIT(2); KSPLIT(3); KSPLIT(4); KSPLIT(5);
    return fp8 ? go(dec_proj_ksplit_kernel<6, u32x2>, (const u32x2*)Wd, &attr[10]) : go(dec_proj_ksplit_kernel<6, bf16x8>, (const bf16x8*)Wd, &attr[11]);
#undef KSPLIT
}

// act = silu(gate) * up of rmsnorm(h), B in (32, 64]: norm kernel -> xn (scratch: 64 x H bf16), then the streaming kernel.  cus: CUs the stream may use.
hipError_t launch_dec_gateup64(hipStream_t s, const bf16_t* h, const bf16_t* ln_w, const void* W13d, const float* wscale, bf16_t* act, bf16_t* xn,
                               int B, int H, int I, float eps, int cus, const float* pend, const float* pend_scale) {
    if (!dec_stream64_supports(B, H) || I % 32 || !xn) return hipErrorInvalidValue;
    HIP_CHECK_RET(launch_dec_norm_ximg(s, h, ln_w, xn, B, H, eps, pend, pend_scale));
    if (cus <= 0) cus = device_cus();
    return wscale ? stream64_launch<S64_GATEUP>(s, xn, (const u32x2*)W13d, wscale, act, B, H, I, I / 16, cus)
                  : stream64_launch<S64_GATEUP>(s, xn, (const bf16x8*)W13d, wscale, act, B, H, I, I / 16, cus);
}

hipError_t launch_dec_lmhead64(hipStream_t s, const bf16_t* h, const bf16_t* ln_w, const void* Wd, const float* wscale, float* logits, bf16_t* xn,
                               int B, int H, int V, float eps, int cus, const float* pend, const float* pend_scale) {
    if (!dec_stream64_supports(B, H) || V % 16 || !xn) return hipErrorInvalidValue;
    HIP_CHECK_RET(launch_dec_norm_ximg(s, h, ln_w, xn, B, H, eps, pend, pend_scale));
    if (cus <= 0) cus = device_cus();
    const int n_items = (V / 16 + 1) / 2;
    return wscale ? stream64_launch<S64_LMHEAD>(s, xn, (const u32x2*)Wd, wscale, logits, B, H, V, n_items, cus)
                  : stream64_launch<S64_LMHEAD>(s, xn, (const bf16x8*)Wd, wscale, logits, B, H, V, n_items, cus);
}
