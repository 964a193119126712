// Fused dense kernels of the decode step: 4 dense launches per layer.
//
//   dec_qkv      rmsnorm(h) prologue -> qkv GEMM -> bias + RoPE + q store + K/V page append epilogue
//   (decode_attn + combine: decode.hip)
//   dec_proj     o_proj / down_proj GEMM -> h += result                       (residual epilogue in place, no norm)
//   dec_gateup   rmsnorm(h) prologue -> gate/up GEMM -> SwiGLU epilogue -> activations in the X image layout
//   dec_lmhead   rmsnorm(h) prologue -> lm_head GEMM -> fp32 logits
//
// What bounds these kernels (tools/bw_probe.hip, tools/decode_bench.hip on MI355X; profiles/r02_*): a dependent kernel
// boundary is 1.6 us; a pure HBM stream of the layer's matrices takes 2.7 / 2.9 / 6-7 / 9.7-10.5 us (o, qkv, down,
// gate|up) including that boundary; ONE workgroup pulls at most ~59 GB/s, 256 of them ~28 GB/s each (7.2 TB/s); a wave's
// loads return IN ORDER and a CU's memory pipeline is shared by its waves; and the prologue's VALU work is paid once per
// workgroup, so its instruction count times the workgroups per CU is real time.  Hence:
//   1. all small operands first (residual rows, norm weights, positions, bias, page ids), unconditionally and in one
//      round trip, pinned (PIN) so that the compiler cannot sink them behind the weight stream;
//   2. then the wave's weight slice, non-temporal, straight into MFMA A-operand registers;
//   3. prologue math (packed fp32 multiplies, v_cvt_pk roundings; only the waves that own a row) while the weights
//      are in flight; normalised rows go to LDS in the X image layout (decode_layout.h);
//   4. MFMAs as the weights land, split-K across the 4-16 waves of ONE workgroup reduced through LDS in a fixed order
//      (deterministic, no atomics, no slabs in HBM), epilogue by one wave whose operands were fetched in step 1.
// N = 1536 / 2048 outputs would be only 96 / 128 sixteen-row tiles, so the projections run one EIGHT-row half tile per
// workgroup (192 / 256 workgroups): the 8 rows of a half are four whole 128-B lines of every 1 KiB fragment chunk, the
// other 8 A-operand rows are fed duplicates and their results ignored (MFMA time is irrelevant here).  No K split across
// workgroups anywhere: every output element is complete inside one workgroup, the residual add happens in place.
// The consumer of a residual-stream row recomputes its RMS statistic itself; numerics are identical to the standalone
// norm kernel (bf16(bf16(x*rstd)*w)).
// Batches above 16 rows: gridDim.x = ceil(B / 16) — workgroup (t, y) handles rows 16t .. 16t+15 (one MFMA column tile) exactly like
// a batch of <= 16 rows.  The tile index is the FAST grid dimension, so the workgroups that read the same weight slice are dispatched
// back to back and run at the same time: the slice comes from HBM once, the other tiles' reads are served by the Infinity Cache
// (lm_head alone is 467 MB: with the tiles a whole grid apart every tile would stream it from HBM again).  decode_layout.h: the X image of a tile sits at element offset t * 16 * K.
// fp8 weights (quant.hip): every kernel is a template over the streamed fragment type WT — bf16x8 (16 B per lane and k-step) or
// u32x2 (8 e4m3 bytes) — which is converted to the bf16 MFMA operand in registers (4 v_cvt_scalef32_pk_bf16_fp8, exact) when
// its MFMA issues; the per-output-channel scale is one more small operand of step 1 and multiplies the reduced accumulator in
// the epilogue.  Half the bytes per step, the same arithmetic as the bf16(q) x scale GEMMs of prefill.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "decode_layout.h"
#include "kernels.h"

namespace {
TRACE_DECL
#include "decode_dev.h"

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_embed_kernel(const int32_t* __restrict__ tokens, const bf16_t* __restrict__ embed,
                                                        bf16_t* __restrict__ h, int dim) {
    const int b = blockIdx.x;
    const bf16_t* src = embed + (size_t)tokens[b] * dim;
    for (int k = threadIdx.x * 8; k < dim; k += 256 * 8)
        *reinterpret_cast<u32x4*>(h + (size_t)b * dim + k) = *reinterpret_cast<const u32x4*>(src + k);
}

// ---- epilogue of one lane of the qkv projection: x = D[row 4g + r][batch row] (split-K reduced) -> scale, bias, bf16 rounding of the qkv
// tensor, RoPE, q store / K, V page append.  f0 = the lane's first feature (q / k heads: rows = features (f0, f0 + 64, f0 + 1, f0 + 65); v
// heads: f0 .. f0 + 3), row = the batch row, pos / page = its position and the page that holds it.  Shared by dec_qkv_kernel and
// dec_qkv_wide_kernel: one definition, the same bits.
template <typename WT>
DEVI void qkv_epilogue(f32x4 x, bool rot, int head, int Hq, int Hkv, int f0, bool has_bias, uint32_t bia0, uint32_t bia1, f32x2 sc0, f32x2 sc1,
                       const float (&rc)[2], const float (&rs)[2], int pos, int page, int row, bf16_t* __restrict__ pool, bf16_t* __restrict__ q_out) {
#pragma clang fp contract(off)      // every fused multiply-add below is written out: the bits must not depend on the kernel this is inlined into
    const int key = pos & 63;
    // bias of rows r = 0..3: rot (lo(bia0), lo(bia1), hi(bia0), hi(bia1)), else (lo(bia0), hi(bia0), lo(bia1), hi(bia1))
    const float b0 = has_bias ? lo_bf(bia0) : 0.f, b1 = has_bias ? (rot ? lo_bf(bia1) : hi_bf(bia0)) : 0.f;
    const float b2 = has_bias ? (rot ? hi_bf(bia0) : lo_bf(bia1)) : 0.f, b3 = has_bias ? hi_bf(bia1) : 0.f;
    if constexpr (is_fp8<WT>::value) {
        x[0] *= sc0[0]; x[1] *= rot ? sc1[0] : sc0[1]; x[2] *= rot ? sc0[1] : sc1[0]; x[3] *= sc1[1];
    }
    const float y[4] = {bf2f(f2bf(x[0] + b0)), bf2f(f2bf(x[1] + b1)), bf2f(f2bf(x[2] + b2)), bf2f(f2bf(x[3] + b3))};   // the qkv output is a bf16 tensor
    if (rot) {
        // pairs (y0, y1) = features (d, d + 64) and (y2, y3) = (d + 1, d + 65)
        const int d = f0;
        const uint32_t lo = pack_bf2(__builtin_fmaf(y[0], rc[0], -(y[1] * rs[0])), __builtin_fmaf(y[2], rc[1], -(y[3] * rs[1])));     // features d, d + 1
        const uint32_t hi = pack_bf2(__builtin_fmaf(y[1], rc[0], y[0] * rs[0]), __builtin_fmaf(y[3], rc[1], y[2] * rs[1]));           // features d + 64, d + 65
        if (head < Hq) {
            bf16_t* qp = q_out + ((size_t)row * Hq + head) * 128;
            *reinterpret_cast<uint32_t*>(qp + d) = lo;
            *reinterpret_cast<uint32_t*>(qp + d + 64) = hi;
        } else {
            bf16_t* kp = pool + ((size_t)(page * Hkv + (head - Hq)) * 2) * PAGE_ELEMS;
            *reinterpret_cast<uint32_t*>(kp + k_chunk(key, d) * 8 + (d & 7)) = lo;
            *reinterpret_cast<uint32_t*>(kp + k_chunk(key, d + 64) * 8 + (d & 7)) = hi;
        }
    } else {
        bf16_t* vp = pool + ((size_t)(page * Hkv + (head - Hq - Hkv)) * 2 + 1) * PAGE_ELEMS;
#pragma unroll
        for (int r = 0; r < 4; ++r) vp[v_off(key, f0 + r)] = f2bf(y[r]);
    }
}

// ------------------------------------------------------------------------------------------------
// grid = (Hq + 2 Hkv) * 16 workgroups (one 8-row HALF tile each: 256 at dots.ocr's 12 + 2 + 2 heads) x 16 waves (K-slices).
// The rows of a q / k head are stored PERMUTED (launch_pack_frag_qkv): tile j of a head holds features 8j .. 8j+7
// interleaved with their RoPE partners, row 2a = feature 8j + a, row 2a + 1 = feature 8j + a + 64, so a lane's 4 accumulator
// rows are two complete rotation pairs and the epilogue needs no exchange.  v heads keep the natural order.  Wave 15 owns the
// epilogue; its operands (position, page id, bias, rotation angles) are fetched / computed while waves < B normalise the rows.
// k-steps per wave = H / 32 / 16 <= NC (the same bound as the row chunks: H <= 512 NC).
// FULL: one whole 16-row tile per workgroup (half as many workgroups: (Hq + 2 Hkv) * 8) — for a stream that is CU-masked to half the chip
// (the decode partition of the pipelined step, engine.hip), where 256 workgroups of 1024 threads would need two rounds.  Same arithmetic per element.
template <int NC, typename WT, bool FULL>
__global__ __launch_bounds__(1024) void dec_qkv_kernel(const bf16_t* __restrict__ h, const bf16_t* __restrict__ ln_w,
                                                       const WT* __restrict__ Wd, const float* __restrict__ wscale, const bf16_t* __restrict__ bias,
                                                       const float* __restrict__ inv_freq, const int32_t* __restrict__ ctx_len,
                                                       const int32_t* __restrict__ block_table, int max_pages,
                                                       bf16_t* __restrict__ pool, bf16_t* __restrict__ q_out,
                                                       int B, int H, int Hq, int Hkv, float eps, int XR) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    {   // this workgroup's 16-row batch tile
        const int t0 = 16 * blockIdx.x;
        h += (size_t)t0 * H; ctx_len += t0; block_table += (size_t)t0 * max_pages; q_out += (size_t)t0 * Hq * 128; B = min(16, B - t0);
    }
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                                 // [H/8][XR][8]
    f32x4* red = reinterpret_cast<f32x4*>(smem + (size_t)XR * H * 2);             // [16 waves][64 lanes]
    const int lane = threadIdx.x & 63, wv = wave_id();
    const int half = FULL ? 0 : (blockIdx.y & 1), tile = FULL ? blockIdx.y : (blockIdx.y >> 1), head = tile >> 3, j = tile & 7;
    const int KS = H / 32;
    const int k0 = wv * KS / 16, k1 = (wv + 1) * KS / 16;
    const int m = lane & 15, g = lane >> 4;
    const bool rot = head < Hq + Hkv;
    const bool epi = wv == 15 && m < B && (FULL || g < 2);         // accumulator rows 4g + r, g < 2: the 8 rows of this half
    TRACE(0);
    // ---- 1. small operands (one round trip)
    Rows<1, NC> R;
    rows_issue<1, NC>(R, h, ln_w, B, H, wv, 16, lane);
    // accumulator rows 4g .. 4g+3 of this lane (g < 2): q / k heads (d, d + 64, d + 1, d + 65) with d = 8j + 4 half + 2g;
    // v heads f .. f + 3 with f = 16j + 8 half + 4g
    const int gg = FULL ? g : (g & 1);
    const int f0 = rot ? 8 * j + 4 * half + 2 * gg : 16 * j + 8 * half + 4 * gg;          // first feature (even)
    const int f1 = rot ? f0 + 64 : f0 + 2;                                                // second bf16 pair
    const int mc = min(m, B - 1);
    const int pos = ctx_len[mc];
    const bf16_t* bp = bias ? bias + head * 128 : ln_w;           // always a valid address; masked below
    const uint32_t bia0 = *reinterpret_cast<const uint32_t*>(bp + f0), bia1 = *reinterpret_cast<const uint32_t*>(bp + f1);
    const float fr0 = inv_freq[f0 & 63], fr1 = inv_freq[(f0 + 1) & 63];
    f32x2 sc0 = {1.f, 1.f}, sc1 = {1.f, 1.f};          // weight scales of rows (f0, f0 + 1), (f1, f1 + 1): the bias pattern
    if constexpr (is_fp8<WT>::value) {
        sc0 = *reinterpret_cast<const f32x2*>(wscale + head * 128 + f0);
        sc1 = *reinterpret_cast<const f32x2*>(wscale + head * 128 + f1);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- 2. weight slice: lane (g, i) reads row (i & 7) + 8 half of the chunk; rows of the other half are duplicates
    const WT* wp = Wd + ((size_t)tile * KS) * 64 + lane_slot<WT>(g, FULL ? m : (m & 7) + 8 * half);
    WT a[NC];
    weights_issue<NC>(a, wp, k0, k1, lane);
    __builtin_amdgcn_sched_barrier(0);
    const int page = block_table[mc * max_pages + (pos >> 6)];    // second (dependent) round trip, behind the weights
    __builtin_amdgcn_sched_barrier(0);
    // ---- 3. prologue math while the weights are in flight
    pin_rows<1, NC>(R);
    PIN(fr0); PIN(fr1); PIN(bia0); PIN(bia1);
    if constexpr (is_fp8<WT>::value) { PIN(sc0); PIN(sc1); }
    TRACE(1);
    float rc[2] = {1.f, 1.f}, rs[2] = {0.f, 0.f};
    if (wv == 15 && rot) {      // precise sincosf (the fast path up to |x| < 2^17); runs beside the other waves' norm
        sincosf((float)pos * fr0, &rs[0], &rc[0]);
        sincosf((float)pos * fr1, &rs[1], &rc[1]);
    }
    rows_norm_to_lds<1, NC>(R, B, H, eps, xs, XR, wv, 16, lane);
    TRACE(2);
    PIN(page);
    __syncthreads();
    TRACE(3);
    // ---- 4. contraction
    red[wv * 64 + lane] = mfma_lds<NC>(a, reinterpret_cast<const bf16x8*>(xs) + g * XR + (m & (XR - 1)), 4 * XR, k0, KS);
    TRACE(4);
    __syncthreads();
    TRACE(5);
    if (!epi) return;
    f32x4 x = {0, 0, 0, 0};
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) x += red[sl * 64 + lane];
    qkv_epilogue<WT>(x, rot, head, Hq, Hkv, f0, bias != nullptr, bia0, bia1, sc0, sc1, rc, rs, pos, page, m, pool, q_out);
    TRACE(6);
}

// ---- split-K reduction of a projection: the 16 slice sums as FOUR quarters, each added in slice order from zero, then (q0 + q1) + (q2 + q3) (round 6: until
// then one chain of 16).  The K-split kernel of decode_b64.hip computes the four quarters in four different workgroups (each needs only its quarter of the X
// image) and the norm kernel that consumes the residual stream adds them — so every projection kernel uses this order: a row's bits must not depend on the
// kernel that ran.
DEVI f32x4 proj_sum16(const f32x4* red, int stride) {
    f32x4 q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        q[i] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int sl = 4 * i; sl < 4 * i + 4; ++sl) q[i] += red[(size_t)sl * stride];
    }
    return (q[0] + q[1]) + (q[2] + q[3]);
}

// ---- residual epilogue of one lane of a projection: h[row][col0 .. col0 + 3] = bf16(residual + sum * scale).  One definition for the
// projection kernels (a row's bits must not depend on the batch it shares, i.e. on which of them ran).
template <typename WT>
DEVI void proj_epilogue(bf16_t* __restrict__ hp, u32x2 res, f32x4 s, f32x4 sc) {
#pragma clang fp contract(off)
    if constexpr (is_fp8<WT>::value) s *= sc;
    const u32x2 o = {pack_bf2(lo_bf(res[0]) + s[0], hi_bf(res[0]) + s[1]), pack_bf2(lo_bf(res[1]) + s[2], hi_bf(res[1]) + s[3])};
    *reinterpret_cast<u32x2*>(hp) = o;
}

// ------------------------------------------------------------------------------------------------
// o_proj / down_proj:  h[m][n] += sum_k X[m][k] W[n][k]   (residual epilogue in place).
// grid N/8 workgroups (one 8-row half tile over the FULL K) x 16 waves (K-slices); X arrives as the X image from the
// previous kernel (L2 / Infinity Cache): its loads and the residual go first.  A slice longer than G k-steps runs in rounds
// of G with the next round's loads issued before this round's MFMAs (down_proj at K = 8960: 17-18 k-steps per wave, G = 6).
#ifndef PROJ_G_FP8
#define PROJ_G_FP8 6
#endif
template <int G, typename WT, bool FULL = false>
__global__ __launch_bounds__(1024) void dec_proj_kernel(const bf16_t* __restrict__ X, const WT* __restrict__ Wd, const float* __restrict__ wscale,
                                                        bf16_t* __restrict__ h, int B, int N, int K, int XR) {
    __shared__ f32x4 red[16 * 64];
    {   // this workgroup's 16-row batch tile
        const int t0 = 16 * blockIdx.x;
        X += (size_t)t0 * K; h += (size_t)t0 * N; B = min(16, B - t0);
    }
    const int lane = threadIdx.x & 63, wv = wave_id();
    const int half = FULL ? 0 : (blockIdx.y & 1), tile = FULL ? blockIdx.y : (blockIdx.y >> 1);
    const int KS = K / 32;
    const int k0 = (int)((uint32_t)(wv * KS) >> 4), k1 = (int)((uint32_t)((wv + 1) * KS) >> 4);
    const int m = lane & 15, g = lane >> 4;
    const bool epi = wv == 15 && m < B && (FULL || g < 2);
    const WT* wp = Wd + ((size_t)tile * KS) * 64 + lane_slot<WT>(g, FULL ? m : (m & 7) + 8 * half);
    const bf16x8* xp = reinterpret_cast<const bf16x8*>(X) + g * XR + (m & (XR - 1));
    const int xstride = 4 * XR;
    const int col0 = tile * 16 + (FULL ? 4 * g : 8 * half + 4 * (g & 1));
    bf16_t* hp = h + (size_t)min(m, B - 1) * N + col0;
    TRACE(0);
    const u32x2 res = *reinterpret_cast<const u32x2*>(hp);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f};
    if constexpr (is_fp8<WT>::value) sc = *reinterpret_cast<const f32x4*>(wscale + col0);
    WT a[G];
    bf16x8 b[G];
#pragma unroll
    for (int jj = 0; jj < G; ++jj) b[jj] = xp[(size_t)min(k0 + jj, KS - 1) * xstride];
    __builtin_amdgcn_sched_barrier(0);
    weights_issue<G>(a, wp, k0, k1, lane);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    int ks = k0;
    for (; ks + G < k1; ks += G) {                        // all rounds but the last
        WT an[G];
        bf16x8 bn[G];
#pragma unroll
        for (int jj = 0; jj < G; ++jj) bn[jj] = xp[(size_t)min(ks + G + jj, KS - 1) * xstride];
        weights_issue<G>(an, wp, ks + G, k1, lane);
#pragma unroll
        for (int jj = 0; jj < G; jj += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_a(a[jj]), b[jj], acc0, 0, 0, 0);
            if (jj + 1 < G) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_a(a[jj + 1]), b[jj + 1], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int jj = 0; jj < G; ++jj) { a[jj] = an[jj]; b[jj] = bn[jj]; }
    }
#pragma unroll
    for (int jj = 0; jj < G; jj += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_a(a[jj]), b[jj], acc0, 0, 0, 0);
        if (jj + 1 < G) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_a(a[jj + 1]), b[jj + 1], acc1, 0, 0, 0);
    }
    TRACE(1);
    red[wv * 64 + lane] = acc0 + acc1;
    PIN(res);
    if constexpr (is_fp8<WT>::value) PIN(sc);
    __syncthreads();
    TRACE(2);
    if (!epi) return;
    const f32x4 s = proj_sum16(red + lane, 64);
    proj_epilogue<WT>(hp, res, s, sc);
    TRACE(3);
}

// ------------------------------------------------------------------------------------------------
// dec_proj for long K at B <= 8 (down_proj: K = 8960, 17-18 k-steps per wave): the register version above runs three dependent
// rounds of 6 k-steps because the X fragments (4 VGPRs per k-step) share the register file with the weights — the in-kernel trace
// shows its waves finishing their K loops after 4.4 / 6.4 / 8.5 us (min / mean / max).  Here the wave's slice of the X image (8-row
// image: 512 B per k-step, <= 10 KiB) is copied global -> LDS by DMA — no registers — and ALL of the wave's weight chunks are
// requested in one round; B fragments come from LDS when their MFMA issues.  A wave reads back only the pieces it copied itself, so
// its own "vmcnt(G)" is all the ordering the copy needs (no barrier).  Measured: 10.1 -> 9.5 us per launch at B = 8 (9.8 -> 9.4 at
// B = 1).  What still holds it at ~9.5 us for 27.5 MB: every one of the 192 workgroups needs the whole 143 KB X image, i.e. as many
// 128-B lines again as the weights (a row-0-only gather for B = 1 was tried: same number of lines, slower).
// LDS: X image 16 K bytes | reduction buffer 16 KiB  (159 744 B at K = 8960: one workgroup per CU; the grid is N / 8 = 192).
constexpr int PROJ_LDS_G = 18;
template <typename WT, bool FULL = false>
__global__ __launch_bounds__(1024) void dec_proj_lds_kernel(const bf16_t* __restrict__ X, const WT* __restrict__ Wd, const float* __restrict__ wscale,
                                                            bf16_t* __restrict__ h, int B, int N, int K) {
    constexpr int G = PROJ_LDS_G, NP = G / 2 + 1, XR = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ximg = smem;                                                            // [K/8][8][8] bf16 = 16 K bytes
    f32x4* red = reinterpret_cast<f32x4*>(smem + (size_t)16 * K);
    const int lane = threadIdx.x & 63, wv = wave_id();
    const int half = FULL ? 0 : (blockIdx.y & 1), tile = FULL ? blockIdx.y : (blockIdx.y >> 1);       // B <= 8: a single batch tile (blockIdx.x == 0)
    const int KS = K / 32;
    const int k0 = (int)((uint32_t)(wv * KS) >> 4), k1 = (int)((uint32_t)((wv + 1) * KS) >> 4);       // k1 - k0 <= G (launcher)
    const int m = lane & 15, g = lane >> 4;
    const bool epi = wv == 15 && m < B && (FULL || g < 2);
    const WT* wp = Wd + ((size_t)tile * KS) * 64 + lane_slot<WT>(g, FULL ? m : (m & 7) + 8 * half);
    const int col0 = tile * 16 + (FULL ? 4 * g : 8 * half + 4 * (g & 1));
    bf16_t* hp = h + (size_t)min(m, B - 1) * N + col0;
    TRACE(0);
    const u32x2 res = *reinterpret_cast<const u32x2*>(hp);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f};
    if constexpr (is_fp8<WT>::value) sc = *reinterpret_cast<const f32x4*>(wscale + col0);
    // this wave's k-steps live in image bytes [512 k0, 512 k1): 1 KiB pieces p0 .. (a piece shared with the neighbour wave is
    // copied by both: same bytes)
    const int p0 = k0 >> 1, plast = (KS - 1) >> 1;
#pragma unroll
    for (int jj = 0; jj < NP; ++jj) {
        const int pc = min(p0 + jj, plast);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + (size_t)pc * 512 + lane * 8),
                                         (__attribute__((address_space(3))) void*)(ximg + pc * 1024), 16, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    WT a[G];
    weights_issue<G>(a, wp, k0, k1, lane);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");          // memory returns in order: residual, scales and the X pieces are in
    const bf16x8* xp = reinterpret_cast<const bf16x8*>(ximg) + g * XR + (m & (XR - 1));
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
    for (int jj = 0; jj < G; jj += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_a(a[jj]), xp[(size_t)min(k0 + jj, KS - 1) * 4 * XR], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_a(a[jj + 1]), xp[(size_t)min(k0 + jj + 1, KS - 1) * 4 * XR], acc1, 0, 0, 0);
    }
    TRACE(1);
    red[wv * 64 + lane] = acc0 + acc1;
    PIN(res);
    if constexpr (is_fp8<WT>::value) PIN(sc);
    __syncthreads();
    TRACE(2);
    if (!epi) return;
    const f32x4 sum = proj_sum16(red + lane, 64);
    proj_epilogue<WT>(hp, res, sum, sc);
    TRACE(3);
}

// ------------------------------------------------------------------------------------------------
// Round 5: WIDE kernels for batches above 16 rows — qkv and the two projections of the 64-row step of the pipelined bench.
// The per-tile kernels above run one workgroup set per 16-row batch tile.  At 64 rows on the 64-CU decode partition that was 512 / 384
// workgroups of 1024 threads (ONE resident per CU: > 64 VGPRs) = 6-8 dispatch rounds, every round paying its norm prologue or its X
// round trips again, and every weight byte crossing a CU four times: dec_qkv 35 us for 6.3 MB, o_proj 16 us for 4.7 MB, down_proj 67 us for
// 27.5 MB (profiles/r05_decode_attn_stream_ab.txt) — 40 % of the step.  Here ONE workgroup holds ALL batch tiles (TT = 2 or 4) of its
// output features: a streamed weight fragment feeds TT MFMAs, the grid is one round (<= the CUs the stream may use), and a workgroup
// takes as many 8-feature units as that needs (NM MFMAs of 16 weight rows = two units each).
//   * waves = the 16 K slices of the per-tile kernels, same boundaries.  Per output element the arithmetic is theirs exactly: an
//     MFMA chain over the slice's even k-steps, one over its odd k-steps, their sum, the 16 slice sums added in order, the shared
//     epilogue (proj_epilogue / qkv_epilogue) — so a row's bits do not depend on the batch it shares (tests:
//     test_wide_kernels_equal_the_one_tile_kernels_bitwise, test_decode_plans_gpu.py).
//   * registers, not LDS, bound the bytes in flight (128 per lane at 1024 threads): the two parities run as two PASSES over the slice
//     with one accumulator set (NM x TT x 4 VGPRs; the even sums wait in the wave's own LDS slots), each pass in rounds of WIDE_G
//     k-steps = WIDE_G x (NM weight + TT activation fragments) requested together.
//   * X fragments come straight from the X image in L2 (a wave reads each of its fragments once: nothing to share through LDS).
//   * dec_qkv_wide has no LDS X image either (64 x 1536 bf16 = 192 KiB would not fit): every wave first computes the RMS statistic of
//     four rows (row_rstd: the per-tile prologue's own function) -> 64 floats in LDS -> barrier; then it loads the raw residual
//     fragments of ITS K slice (TT x <= 3 x 16 B per lane, L2 hits) and normalises them in registers (norm8: the same roundings).
//   * the NM x TT (<= 8) accumulator tiles are reduced and finished by waves 15, 14, ... — one tile each.
// LDS: [16 slices][NM][TT][64 lanes] f32x4 = NM x TT x 16 KiB (+ 64 floats): one workgroup per CU.
constexpr int WIDE_G = 3;

// ONE: the whole slice is at most WIDE_G k-steps (o_proj: K = 1536 -> 3): both parities' fragments are requested in ONE round (the
// in-kernel timeline showed two dependent round trips of ~3 us each on the partition, profiles/r05_decode_trace_b64_partition.txt).
template <int NM, int TT, typename WT, bool ONE>
__global__ __launch_bounds__(1024) void dec_proj_wide_kernel(const bf16_t* __restrict__ X, const WT* __restrict__ Wd, const float* __restrict__ wscale,
                                                             bf16_t* __restrict__ h, int B, int N, int K, int NU) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* red = reinterpret_cast<f32x4*>(smem);                                  // [16][NM][TT][64]
    const int lane = threadIdx.x & 63, wv = wave_id();
    const int m = lane & 15, g = lane >> 4;
    const int n_tiles = (B + 15) >> 4, n_units = N >> 3;
    const int u0 = blockIdx.y * NU;                                               // first 8-feature unit of this workgroup (< n_units: launcher)
    const int KS = K / 32;
    const int k0 = (int)((uint32_t)(wv * KS) >> 4), k1 = (int)((uint32_t)((wv + 1) * KS) >> 4);
    // MFMA j multiplies units u0 + 2j (A rows 0-7) and u0 + 2j + 1 (A rows 8-15; a duplicate of the first when the workgroup has no such unit).
    // Addresses = wave-uniform 64-bit base (scalar registers) + a 32-bit per-lane byte offset: the register file is what bounds this kernel.
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
    // X image of tile t (XR = 16): fragment (k-step, lane) at byte t * 32 K + k-step * 1024 + lane * 16; tiles past the batch re-read the last one
    const char* xbase = reinterpret_cast<const char*>(X);
    const uint32_t xoff = lane * 16u;
    uint32_t toff[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) toff[t] = (uint32_t)min(t, n_tiles - 1) * 32u * (uint32_t)K;
    // epilogue role: wave 15 - e finishes accumulator tile e = (MFMA je, batch tile te)
    const int e = 15 - wv;
    const bool ew = e < NM * TT;                                                  // wave-uniform
    const int je = ew ? e / TT : 0, te = ew ? e % TT : 0;
    const int ul = 2 * je + (g >> 1), unit = u0 + ul, row = 16 * te + m;          // accumulator rows 4g + r: unit ul of the workgroup, features 4 (g & 1) + r
    const bool epi = ew && ul < NU && unit < n_units && row < B;
    const int col0 = 8 * min(unit, n_units - 1) + 4 * (g & 1);
    bf16_t* hp = h + (size_t)min(row, B - 1) * N + col0;
    TRACE(0);
    if constexpr (ONE) {
        WT a[NM][WIDE_G];
        bf16x8 b[TT][WIDE_G];
#pragma unroll
        for (int o = 0; o < WIDE_G; ++o) {
            const int k = k0 + o;
            const bool ok = k < k1;
            const int kc = min(k, KS - 1);
            const char* wk = ok ? wbase + (size_t)k * 64 * sizeof(WT) : zbase;
#pragma unroll
            for (int j = 0; j < NM; ++j) a[j][o] = __builtin_nontemporal_load(reinterpret_cast<const WT*>(wk + (ok ? woff[j] : zoff)));
#pragma unroll
            for (int t = 0; t < TT; ++t) b[t][o] = *reinterpret_cast<const bf16x8*>(xbase + (size_t)(toff[t] + (uint32_t)kc * 1024u) + xoff);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            f32x4 acc[NM][TT];
#pragma unroll
            for (int j = 0; j < NM; ++j)
#pragma unroll
                for (int t = 0; t < TT; ++t) acc[j][t] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int o = par; o < WIDE_G; o += 2)
#pragma unroll
                for (int j = 0; j < NM; ++j) {
                    const bf16x8 wa = as_a(a[j][o]);
#pragma unroll
                    for (int t = 0; t < TT; ++t) acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, b[t][o], acc[j][t], 0, 0, 0);
                }
#pragma unroll
            for (int j = 0; j < NM; ++j)
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    f32x4* r = red + ((size_t)((wv * NM + j) * TT + t)) * 64 + lane;
                    if (par == 0) *r = acc[j][t];
                    else *r = *r + acc[j][t];
                }
        }
    } else
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        f32x4 acc[NM][TT];
#pragma unroll
        for (int j = 0; j < NM; ++j)
#pragma unroll
            for (int t = 0; t < TT; ++t) acc[j][t] = f32x4{0, 0, 0, 0};
        for (int kb = k0 + par; kb < k1; kb += 2 * WIDE_G) {                      // wave-uniform trip count
            WT a[NM][WIDE_G];
            bf16x8 b[TT][WIDE_G];
#pragma unroll
            for (int jj = 0; jj < WIDE_G; ++jj) {
                const int k = kb + 2 * jj;
                const bool ok = k < k1;                                           // slots past the slice multiply a chunk of zeros
                const int kc = min(k, KS - 1);
                const char* wk = ok ? wbase + (size_t)k * 64 * sizeof(WT) : zbase;                 // wave-uniform
#pragma unroll
                for (int j = 0; j < NM; ++j) a[j][jj] = __builtin_nontemporal_load(reinterpret_cast<const WT*>(wk + (ok ? woff[j] : zoff)));
#pragma unroll
                for (int t = 0; t < TT; ++t) b[t][jj] = *reinterpret_cast<const bf16x8*>(xbase + (size_t)(toff[t] + (uint32_t)kc * 1024u) + xoff);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < WIDE_G; ++jj)
#pragma unroll
                for (int j = 0; j < NM; ++j) {
                    const bf16x8 wa = as_a(a[j][jj]);
#pragma unroll
                    for (int t = 0; t < TT; ++t) acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, b[t][jj], acc[j][t], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the wave's own slots: even sums wait here for the odd ones (same lane writes and reads: no barrier)
#pragma unroll
        for (int j = 0; j < NM; ++j)
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                f32x4* r = red + ((size_t)((wv * NM + j) * TT + t)) * 64 + lane;
                if (par == 0) *r = acc[j][t];
                else *r = *r + acc[j][t];
            }
    }
    TRACE(1);
    // the residual (and the fp8 scales) only now: through the K loops every register carries fragments; its round trip (an L2 hit) runs
    // under the barrier and the 16 reduction reads
    const u32x2 res = *reinterpret_cast<const u32x2*>(hp);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f};
    if constexpr (is_fp8<WT>::value) sc = *reinterpret_cast<const f32x4*>(wscale + col0);
    __syncthreads();
    TRACE(2);
    if (!epi) return;
    const f32x4 sum = proj_sum16(red + ((size_t)(je * TT + te)) * 64 + lane, NM * TT * 64);
    proj_epilogue<WT>(hp, res, sum, sc);
    TRACE(3);
}

// grid.y = ceil(n_out / NM) workgroups, n_out = (Hq + 2 Hkv) * 8 whole 16-row tiles of the (permuted, launch_pack_frag_qkv) qkv weight.
// XIMG (round 6, batches above 32 rows): h is the NORMALISED X image of the rows (dec_norm_ximg_kernel, decode_b64.hip: the same row_rstd / norm8,
// so the same bits) — no statistics, no in-register normalisation, no first barrier: a wave's fragments of its K slice are requested with its weights.
template <int NM, int TT, int NC, typename WT, bool XIMG = false>
__global__ __launch_bounds__(1024) void dec_qkv_wide_kernel(const bf16_t* __restrict__ h, const bf16_t* __restrict__ ln_w,
                                                            const WT* __restrict__ Wd, const float* __restrict__ wscale, const bf16_t* __restrict__ bias,
                                                            const float* __restrict__ inv_freq, const int32_t* __restrict__ ctx_len,
                                                            const int32_t* __restrict__ block_table, int max_pages,
                                                            bf16_t* __restrict__ pool, bf16_t* __restrict__ q_out,
                                                            int B, int H, int Hq, int Hkv, float eps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* red = reinterpret_cast<f32x4*>(smem);                                  // [16][NM][TT][64]
    float* rstd_s = reinterpret_cast<float*>(smem + (size_t)16 * NM * TT * 64 * sizeof(f32x4));     // [16 TT]
    const int lane = threadIdx.x & 63, wv = wave_id();
    const int m = lane & 15, g = lane >> 4;
    const int n_out = (Hq + 2 * Hkv) * 8, tl0 = blockIdx.y * NM;
    const int KS = H / 32;
    const int k0 = wv * KS / 16, k1 = (wv + 1) * KS / 16;                         // k1 - k0 <= NC (launcher)
    TRACE(0);
    // ---- 1. one round trip: the rows whose statistic this wave computes (wv, wv + 16, ...), its weight slice, the norm weights of its K
    // slice, the epilogue waves' small operands
    u32x4 v[XIMG ? 1 : TT][XIMG ? 1 : NC];
    u32x4 wf[XIMG ? 1 : NC];
    bf16x8 xf[TT][NC];
    if constexpr (XIMG) {
        // lane (g, m) of k-step ks of tile t: 16 B at t * 32 H + ks * 1024 + lane * 16 (tiles past the batch re-read the last one)
        const char* xb = reinterpret_cast<const char*>(h) + lane * 16;
        const int n_bt = (B + 15) >> 4;
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int o = 0; o < NC; ++o)
                xf[t][o] = *reinterpret_cast<const bf16x8*>(xb + (size_t)min(t, n_bt - 1) * 32 * H + (size_t)min(k0 + o, KS - 1) * 1024);
    } else {
#pragma unroll
        for (int i = 0; i < TT; ++i) {
            const int r = min(wv + 16 * i, B - 1);
#pragma unroll
            for (int c = 0; c < NC; ++c) v[i][c] = *reinterpret_cast<const u32x4*>(h + (size_t)r * H + min(c * 512 + lane * 8, H - 8));
        }
#pragma unroll
        for (int o = 0; o < NC; ++o) wf[o] = *reinterpret_cast<const u32x4*>(ln_w + (size_t)min(k0 + o, KS - 1) * 32 + g * 8);
    }
    const int e = 15 - wv;
    const bool ew = e < NM * TT;                                                  // wave-uniform
    const int je = ew ? e / TT : 0, te = ew ? e % TT : 0;
    const int tile = min(tl0 + je, n_out - 1), head = tile >> 3, j8 = tile & 7;
    const bool rot = head < Hq + Hkv;
    const int row = 16 * te + m, mc = min(row, B - 1);
    const bool epi = ew && tl0 + je < n_out && row < B;
    const int f0 = rot ? 8 * j8 + 2 * g : 16 * j8 + 4 * g;                        // whole tiles: accumulator rows 4g .. 4g + 3 (dec_qkv_kernel, FULL)
    const int f1 = rot ? f0 + 64 : f0 + 2;
    const int pos = ctx_len[mc];
    const bf16_t* bp = bias ? bias + head * 128 : ln_w;
    const uint32_t bia0 = *reinterpret_cast<const uint32_t*>(bp + f0), bia1 = *reinterpret_cast<const uint32_t*>(bp + f1);
    const float fr0 = inv_freq[f0 & 63], fr1 = inv_freq[(f0 + 1) & 63];
    f32x2 sc0 = {1.f, 1.f}, sc1 = {1.f, 1.f};
    if constexpr (is_fp8<WT>::value) {
        sc0 = *reinterpret_cast<const f32x2*>(wscale + head * 128 + f0);
        sc1 = *reinterpret_cast<const f32x2*>(wscale + head * 128 + f1);
    }
    __builtin_amdgcn_sched_barrier(0);
    const WT* zc = reinterpret_cast<const WT*>(g_zero_chunk) + lane;
    WT a[NM][NC];
#pragma unroll
    for (int j = 0; j < NM; ++j) {
        const WT* wp = Wd + (size_t)min(tl0 + j, n_out - 1) * KS * 64 + lane_slot<WT>(g, m);
#pragma unroll
        for (int o = 0; o < NC; ++o) a[j][o] = __builtin_nontemporal_load(k0 + o < k1 ? wp + (size_t)(k0 + o) * 64 : zc);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int page = block_table[(size_t)mc * max_pages + (pos >> 6)];           // second (dependent) round trip, behind the weights
    __builtin_amdgcn_sched_barrier(0);
    float rc[2] = {1.f, 1.f}, rs[2] = {0.f, 0.f};
    if constexpr (XIMG) {
        TRACE(1);
        if (ew && rot) {                  // precise sincosf, as in dec_qkv_kernel (beside the other waves' loads)
            sincosf((float)pos * fr0, &rs[0], &rc[0]);
            sincosf((float)pos * fr1, &rs[1], &rc[1]);
        }
        TRACE(3);
    } else {
    // ---- 2. statistics
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int c = 0; c < NC; ++c) PIN(v[i][c]);
#pragma unroll
    for (int o = 0; o < NC; ++o) PIN(wf[o]);
    TRACE(1);
#pragma unroll
    for (int i = 0; i < TT; ++i) {
        const int r = wv + 16 * i;                                                // wave-uniform
        if (r < B) {
            const float rstd = row_rstd<NC>(v[i], H, eps, lane);
            if (lane == 0) rstd_s[r] = rstd;
        }
    }
    if (ew && rot) {                  // precise sincosf, as in dec_qkv_kernel
        sincosf((float)pos * fr0, &rs[0], &rc[0]);
        sincosf((float)pos * fr1, &rs[1], &rc[1]);
    }
    __syncthreads();
    TRACE(2);
    // ---- 3. this wave's K slice of every row, normalised in registers: lane (g, m) of k-step ks holds X[16 t + m][32 ks + 8 g .. + 7].
    // (Requesting the raw fragments BEFORE the barrier — their addresses do not need the statistics — measured slower: 16.9 vs 15.7 us
    // on the partition, 16.3 vs 14.3 on the whole chip, profiles/r05_decode_wide_second_pass.txt.)
    {
        u32x4 raw[TT][NC];
        float rsd[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const int r = min(16 * t + m, B - 1);
            rsd[t] = rstd_s[r];
#pragma unroll
            for (int o = 0; o < NC; ++o) raw[t][o] = *reinterpret_cast<const u32x4*>(h + (size_t)r * H + (size_t)min(k0 + o, KS - 1) * 32 + g * 8);
        }
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int o = 0; o < NC; ++o) xf[t][o] = __builtin_bit_cast(bf16x8, norm8(raw[t][o], wf[o], rsd[t]));
    }
    TRACE(3);
    }
    // ---- 4. contraction: even k-steps, then odd k-steps (mfma_lds: acc0 / acc1)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        f32x4 acc[NM][TT];
#pragma unroll
        for (int j = 0; j < NM; ++j)
#pragma unroll
            for (int t = 0; t < TT; ++t) acc[j][t] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int o = par; o < NC; o += 2)
#pragma unroll
            for (int j = 0; j < NM; ++j) {
                const bf16x8 wa = as_a(a[j][o]);
#pragma unroll
                for (int t = 0; t < TT; ++t) acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xf[t][o], acc[j][t], 0, 0, 0);
            }
#pragma unroll
        for (int j = 0; j < NM; ++j)
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                f32x4* r = red + ((size_t)((wv * NM + j) * TT + t)) * 64 + lane;
                if (par == 0) *r = acc[j][t];
                else *r = *r + acc[j][t];
            }
    }
    TRACE(4);
    PIN(page); PIN(fr0); PIN(fr1); PIN(bia0); PIN(bia1);
    if constexpr (is_fp8<WT>::value) { PIN(sc0); PIN(sc1); }
    __syncthreads();
    TRACE(5);
    if (!epi) return;
    f32x4 x = {0, 0, 0, 0};
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) x += red[((size_t)((sl * NM + je) * TT + te)) * 64 + lane];
    qkv_epilogue<WT>(x, rot, head, Hq, Hkv, f0, bias != nullptr, bia0, bia1, sc0, sc1, rc, rs, pos, page, row, pool, q_out);
    TRACE(6);
}

// ------------------------------------------------------------------------------------------------
// act = silu(gate) * up with gate/up = rmsnorm(h) @ W13^T.  grid I/16 workgroups x GU_WAVES waves (K-slices);
// workgroup = one (gate tile, up tile) pair of the packed W13 (64-row groups: 32 gate rows | 32 up rows).
constexpr int GU_G = 12;         // k-steps per wave held in registers (H = 1536: 48 k-steps / 4 waves)
constexpr int GU_WAVES = 4;      // 12 (single round) was measured slower: 12 waves x 49 KB LDS leaves 560 workgroups non-resident
#ifndef GU_EARLY_N
#define GU_EARLY_N 4
#endif
constexpr int GU_EARLY = GU_EARLY_N;      // k-steps requested before the norm prologue; the rest right after it, when the row registers
                                 // are free: all I/16 workgroups are resident only at 3 waves per SIMD, i.e. <= 168 VGPRs

// PAIRS (gate tile, up tile) pairs per workgroup: waves [p * GU_WAVES, (p + 1) * GU_WAVES) split the K of pair p; all PAIRS * GU_WAVES
// waves share one norm prologue, so the residual rows are fetched and normalised by half as many workgroups when PAIRS = 2.
// Round 4: the workgroup WALKS the pair groups blockIdx.y, blockIdx.y + gridDim.y, ... (one norm prologue, the X image stays in LDS).  With
// gridDim.y = I / 16 / PAIRS (the whole-chip plan) that is exactly one group per workgroup, as before; on a stream that is CU-masked
// to the decode partition of the pipelined step the launcher caps the grid at what the partition holds at once (4 workgroups per CU),
// so the 560 pairs of dots.ocr run as ONE resident round whose leftover pairs are picked up by workgroups that are already there
// (norm done, streaming) instead of a second dispatch round of cold workgroups.  A group's next weights are requested right after its
// MFMAs, before the split-K reduction and the SwiGLU epilogue.  Same arithmetic per element.
// TT = batch tiles per workgroup (round 4).  TT = 1: one 16-row tile, as above.  TT = 2 (batches above 16 rows): the X images of TWO tiles sit
// in LDS (2 x 48 KiB + the reduction buffers = 128 KiB at H = 1536: one workgroup per CU) and every streamed weight fragment feeds one MFMA
// per tile, so a 64-row batch pulls the 55 MB of W13 through the CUs twice instead of four times.  Per output element the MFMAs, the
// split-K reduction order and the epilogue are those of TT = 1: bit-identical activations whatever the batch a row sits in.
template <int MAXR, int NC, typename WT, int PAIRS, int TT = 1>
__global__ __launch_bounds__(PAIRS * GU_WAVES * 64) void dec_gateup_kernel(const bf16_t* __restrict__ h, const bf16_t* __restrict__ ln_w,
                                                                   const WT* __restrict__ Wd, const float* __restrict__ wscale, bf16_t* __restrict__ act,
                                                                   int B, int H, int I, float eps, int XR) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    {   // this workgroup's TT 16-row batch tiles
        const int t0 = 16 * TT * blockIdx.x;
        h += (size_t)t0 * H; act += (size_t)t0 * I; B = min(16 * TT, B - t0);
    }
    constexpr int NW = PAIRS * GU_WAVES;
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                                 // [TT][H/8][XR][8]
    f32x4* red = reinterpret_cast<f32x4*>(smem + (size_t)TT * XR * H * 2);        // [TT][PAIRS * GU_WAVES][2][64]
    const int lane = threadIdx.x & 63, wv = wave_id();
    const int pw = wv / GU_WAVES, kw = wv % GU_WAVES;                             // pair inside the workgroup, K slice
    const int n_pairs = I / 16;
    int pair = blockIdx.y * PAIRS + pw;                                           // < n_pairs (launcher: gridDim.y * PAIRS <= n_pairs)
    const int KS = H / 32;
    const int k0 = kw * KS / GU_WAVES, k1 = (kw + 1) * KS / GU_WAVES;             // k1 - k0 <= GU_G (launcher)
    const bf16x8* xp = reinterpret_cast<const bf16x8*>(xs) + (lane >> 4) * XR + (lane & (XR - 1));
    const int xstride = 4 * XR;
    const size_t xtile = (size_t)2 * H;                                           // tile 1's image: 16 * H bf16 = 2 H fragments further (TT = 2: XR = 16)
    const int ls = lane_slot<WT>(lane >> 4, lane & 15);
    const int m = lane & 15, g = lane >> 4;
    // wave-uniform chunk bases (scalar registers); the lane's slot `ls` is added per load
    auto gate_ptr = [&](int pr) { return Wd + ((size_t)((pr >> 1) * 4 + (pr & 1)) * KS + k0) * 64; };
    auto up_ptr = [&](int pr) { return Wd + ((size_t)((pr >> 1) * 4 + 2 + (pr & 1)) * KS + k0) * 64; };
    const WT* wg = gate_ptr(pair);
    const WT* wu = up_ptr(pair);
    TRACE(0);
    Rows<MAXR, NC> R;
    rows_issue<MAXR, NC>(R, h, ln_w, B, H, wv, NW, lane);
    // fp8: per-output-channel scales of the packed-W13 rows (G*4 + a)*16 + 4g + r (gate), + 32 (up): small operands, fetched with the rows
    auto scale_g = [&](int pr) { return *reinterpret_cast<const f32x4*>(wscale + ((pr >> 1) * 4 + (pr & 1)) * 16 + 4 * g); };
    auto scale_u = [&](int pr) { return *reinterpret_cast<const f32x4*>(wscale + ((pr >> 1) * 4 + 2 + (pr & 1)) * 16 + 4 * g); };
    f32x4 scg = {1.f, 1.f, 1.f, 1.f}, scu = {1.f, 1.f, 1.f, 1.f};
    if constexpr (is_fp8<WT>::value) { scg = scale_g(pair); scu = scale_u(pair); }
    __builtin_amdgcn_sched_barrier(0);
    WT a_[GU_G], u_[GU_G];
    const WT* zc = reinterpret_cast<const WT*>(g_zero_chunk);          // a slice shorter than GU_G reads zeros: uniform pointer select, no branch
#pragma unroll
    for (int jj = 0; jj < GU_EARLY; ++jj) {
        a_[jj] = __builtin_nontemporal_load((k0 + jj < k1 ? wg + (size_t)jj * 64 : zc) + ls);
        u_[jj] = __builtin_nontemporal_load((k0 + jj < k1 ? wu + (size_t)jj * 64 : zc) + ls);
    }
    __builtin_amdgcn_sched_barrier(0);
    pin_rows<MAXR, NC>(R);
    if constexpr (is_fp8<WT>::value) { PIN(scg); PIN(scu); }
    TRACE(1);
    if constexpr (TT == 1) rows_norm_to_lds<MAXR, NC>(R, B, H, eps, xs, XR, wv, NW, lane);
    else {
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int r = wv + i * NW;                                 // wave-uniform; row r & 15 of tile r >> 4
            if (r < B) row_norm_to_lds<NC>(R.v[i], R.w, r & 15, H, eps, xs + (size_t)(r >> 4) * 16 * H, XR, lane);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jj = GU_EARLY; jj < GU_G; ++jj) {
        a_[jj] = __builtin_nontemporal_load((k0 + jj < k1 ? wg + (size_t)jj * 64 : zc) + ls);
        u_[jj] = __builtin_nontemporal_load((k0 + jj < k1 ? wu + (size_t)jj * 64 : zc) + ls);
    }
    __builtin_amdgcn_sched_barrier(0);
    TRACE(2);
    for (;;) {
        __syncthreads();                                   // the X image is complete / the reduction buffer of the previous group has been read
        TRACE(3);
        f32x4 ag[TT], au[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) { ag[t] = f32x4{0, 0, 0, 0}; au[t] = f32x4{0, 0, 0, 0}; }
#pragma unroll
        for (int jj = 0; jj < GU_G; ++jj) {
            const bf16x8 wa = as_a(a_[jj]), wb = as_a(u_[jj]);
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const bf16x8 b = xp[t * xtile + (size_t)min(k0 + jj, KS - 1) * xstride];
                ag[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, b, ag[t], 0, 0, 0);
                au[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, b, au[t], 0, 0, 0);
            }
        }
        TRACE(4);
        const int cur = pair;
        pair += gridDim.y * PAIRS;
        const bool more = pair < n_pairs;                  // uniform over the workgroup (PAIRS divides n_pairs: launcher)
        f32x4 scg_n = scg, scu_n = scu;
        if (more) {                                        // the next group's weights: requested before this group's reduction and epilogue
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (is_fp8<WT>::value) { scg_n = scale_g(pair); scu_n = scale_u(pair); }
            wg = gate_ptr(pair);
            wu = up_ptr(pair);
#pragma unroll
            for (int jj = 0; jj < GU_G; ++jj) {
                a_[jj] = __builtin_nontemporal_load((k0 + jj < k1 ? wg + (size_t)jj * 64 : zc) + ls);
                u_[jj] = __builtin_nontemporal_load((k0 + jj < k1 ? wu + (size_t)jj * 64 : zc) + ls);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            red[((t * NW + wv) * 2) * 64 + lane] = ag[t];
            red[((t * NW + wv) * 2 + 1) * 64 + lane] = au[t];
        }
        __syncthreads();
        TRACE(5);
        if (kw == 0) {
            const int G = cur >> 1, a = cur & 1;
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                if (m + 16 * t < B) {
                    f32x4 gs = ag[t], us = au[t];
#pragma unroll
                    for (int ww = 1; ww < GU_WAVES; ++ww) {
                        gs += red[((t * NW + pw * GU_WAVES + ww) * 2) * 64 + lane];
                        us += red[((t * NW + pw * GU_WAVES + ww) * 2 + 1) * 64 + lane];
                    }
                    if constexpr (is_fp8<WT>::value) { gs *= scg; us *= scu; }
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = gs[r] / (1.0f + __expf(-gs[r])) * us[r];
                    store_frag4(act + (size_t)t * 16 * I, m, G * 32 + a * 16 + 4 * g, XR, o[0], o[1], o[2], o[3]);
                }
            }
        }
        TRACE(6);
        if (!more) break;
        scg = scg_n; scu = scu_n;
    }
}

// ------------------------------------------------------------------------------------------------
// logits[m][n] = rmsnorm(h[m]) . lm_head[n]   (fp32, never rounded).  grid ceil(V/256) workgroups x 16 waves, wave = one
// 16-row vocabulary tile over the full K in rounds of LM_G k-steps (the next round's loads are issued before this round's
// MFMAs).  16 tiles per workgroup share one norm prologue.
#ifndef LM_G_FP8
#define LM_G_FP8 8
#endif
template <typename WT> struct LmG { static constexpr int value = 8; };
template <> struct LmG<u32x2> { static constexpr int value = LM_G_FP8; };

template <int NC, typename WT>
__global__ __launch_bounds__(1024) void dec_lmhead_kernel(const bf16_t* __restrict__ h, const bf16_t* __restrict__ ln_w,
                                                          const WT* __restrict__ Wd, const float* __restrict__ wscale, float* __restrict__ logits,
                                                          int B, int H, int V, float eps, int XR) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    {   // this workgroup's 16-row batch tile
        const int t0 = 16 * blockIdx.x;
        h += (size_t)t0 * H; logits += (size_t)t0 * V; B = min(16, B - t0);
    }
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
    const int lane = threadIdx.x & 63, wv = wave_id();
    const int n_tile = min((int)blockIdx.y * 16 + wv, V / 16 - 1);     // tail waves recompute the last tile (same values)
    const int KS = H / 32;                                             // a multiple of LM_G (launcher)
    constexpr int LM_G = LmG<WT>::value;
    const WT* wp = Wd + ((size_t)n_tile * KS) * 64 + lane_slot<WT>(lane >> 4, lane & 15);
    Rows<1, NC> R;
    rows_issue<1, NC>(R, h, ln_w, B, H, wv, 16, lane);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f};
    if constexpr (is_fp8<WT>::value) sc = *reinterpret_cast<const f32x4*>(wscale + n_tile * 16 + 4 * (lane >> 4));
    __builtin_amdgcn_sched_barrier(0);
    WT a[LM_G];
    weights_issue<LM_G>(a, wp, 0, KS, lane);
    __builtin_amdgcn_sched_barrier(0);
    pin_rows<1, NC>(R);
    if constexpr (is_fp8<WT>::value) PIN(sc);
    rows_norm_to_lds<1, NC>(R, B, H, eps, xs, XR, wv, 16, lane);
    __syncthreads();
    const bf16x8* xp = reinterpret_cast<const bf16x8*>(xs) + (lane >> 4) * XR + (lane & (XR - 1));
    const int xstride = 4 * XR;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    int ks = 0;
    for (; ks + LM_G < KS; ks += LM_G) {                 // all rounds but the last: next round's loads first
        WT an[LM_G];
        weights_issue<LM_G>(an, wp, ks + LM_G, KS, lane);
#pragma unroll
        for (int jj = 0; jj < LM_G; jj += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_a(a[jj]), xp[(size_t)(ks + jj) * xstride], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_a(a[jj + 1]), xp[(size_t)(ks + jj + 1) * xstride], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int jj = 0; jj < LM_G; ++jj) a[jj] = an[jj];
    }
#pragma unroll
    for (int jj = 0; jj < LM_G; jj += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_a(a[jj]), xp[(size_t)(ks + jj) * xstride], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_a(a[jj + 1]), xp[(size_t)(ks + jj + 1) * xstride], acc1, 0, 0, 0);
    }
    f32x4 acc = acc0 + acc1;
    if constexpr (is_fp8<WT>::value) acc *= sc;
    const int m = lane & 15, g = lane >> 4;
    if (m < B) *reinterpret_cast<f32x4*>(logits + (size_t)m * V + n_tile * 16 + 4 * g) = acc;
}

// Round 4: batches above 16 rows — TWO 16-row batch tiles per workgroup.  The per-tile kernel above runs one workgroup set per tile,
// so at 64 rows every lm_head byte crosses a CU's load path four times (once from HBM, three times from L2 / Infinity Cache): 0.27 ms
// per step at 8 rows on the whole chip became 1 ms on the 64-CU decode partition of the pipelined step.  Here both tiles' X images sit
// in LDS (2 x 16 x H x 2 B = 96 KiB at H = 1536: one workgroup per CU) and every streamed weight fragment feeds one MFMA per tile.
// Same MFMAs in the same order per output element as the per-tile kernel (even / odd k-steps on two accumulators, added at the end):
// bit-identical logits, so a sequence's tokens still do not depend on the batch it shares.
template <int NC, typename WT>
__global__ __launch_bounds__(1024) void dec_lmhead2_kernel(const bf16_t* __restrict__ h, const bf16_t* __restrict__ ln_w,
                                                           const WT* __restrict__ Wd, const float* __restrict__ wscale, float* __restrict__ logits,
                                                           int B, int H, int V, float eps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int XR = 16;
    {   // this workgroup's pair of 16-row batch tiles
        const int t0 = 32 * blockIdx.x;
        h += (size_t)t0 * H; logits += (size_t)t0 * V; B = min(32, B - t0);
    }
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                      // [2][H/8][16][8]
    const int lane = threadIdx.x & 63, wv = wave_id();
    const int n_tile = min((int)blockIdx.y * 16 + wv, V / 16 - 1);
    const int KS = H / 32;
    constexpr int LM_G = LmG<WT>::value;
    const WT* wp = Wd + ((size_t)n_tile * KS) * 64 + lane_slot<WT>(lane >> 4, lane & 15);
    Rows<2, NC> R;                                                     // rows wv and wv + 16: row wv of either tile
    rows_issue<2, NC>(R, h, ln_w, B, H, wv, 16, lane);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f};
    if constexpr (is_fp8<WT>::value) sc = *reinterpret_cast<const f32x4*>(wscale + n_tile * 16 + 4 * (lane >> 4));
    __builtin_amdgcn_sched_barrier(0);
    WT a[LM_G];
    weights_issue<LM_G>(a, wp, 0, KS, lane);
    __builtin_amdgcn_sched_barrier(0);
    pin_rows<2, NC>(R);
    if constexpr (is_fp8<WT>::value) PIN(sc);
#pragma unroll
    for (int t = 0; t < 2; ++t)
        if (wv + 16 * t < B) row_norm_to_lds<NC>(R.v[t], R.w, wv, H, eps, xs + (size_t)t * 16 * H, XR, lane);
    __syncthreads();
    const bf16x8* xp0 = reinterpret_cast<const bf16x8*>(xs) + (lane >> 4) * XR + (lane & 15);
    const bf16x8* xp1 = xp0 + (size_t)2 * H;                           // tile 1: 16 * H bf16 = 2 H fragments further
    const int xstride = 4 * XR;
    f32x4 acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc10 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
    int ks = 0;
    for (; ks + LM_G < KS; ks += LM_G) {
        WT an[LM_G];
        weights_issue<LM_G>(an, wp, ks + LM_G, KS, lane);
#pragma unroll
        for (int jj = 0; jj < LM_G; jj += 2) {
            const bf16x8 w0 = as_a(a[jj]), w1 = as_a(a[jj + 1]);
            acc00 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xp0[(size_t)(ks + jj) * xstride], acc00, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xp1[(size_t)(ks + jj) * xstride], acc10, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, xp0[(size_t)(ks + jj + 1) * xstride], acc01, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, xp1[(size_t)(ks + jj + 1) * xstride], acc11, 0, 0, 0);
        }
#pragma unroll
        for (int jj = 0; jj < LM_G; ++jj) a[jj] = an[jj];
    }
#pragma unroll
    for (int jj = 0; jj < LM_G; jj += 2) {
        const bf16x8 w0 = as_a(a[jj]), w1 = as_a(a[jj + 1]);
        acc00 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xp0[(size_t)(ks + jj) * xstride], acc00, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xp1[(size_t)(ks + jj) * xstride], acc10, 0, 0, 0);
        acc01 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, xp0[(size_t)(ks + jj + 1) * xstride], acc01, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, xp1[(size_t)(ks + jj + 1) * xstride], acc11, 0, 0, 0);
    }
    f32x4 acc0 = acc00 + acc01, acc1 = acc10 + acc11;
    if constexpr (is_fp8<WT>::value) { acc0 *= sc; acc1 *= sc; }
    const int m = lane & 15, g = lane >> 4;
    if (m < B) *reinterpret_cast<f32x4*>(logits + (size_t)m * V + n_tile * 16 + 4 * g) = acc0;
    if (m + 16 < B) *reinterpret_cast<f32x4*>(logits + (size_t)(m + 16) * V + n_tile * 16 + 4 * g) = acc1;
}

// Dynamic-LDS opt-in above 64 KiB, once per (kernel, device): handles of different devices may live in one process.
template <typename Kern>
hipError_t ensure_lds(Kern kern, size_t bytes, uint32_t* done_mask) {
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

}  // namespace

#ifdef DOTS_TRACE
void dots_trace_set_fused(unsigned long long* buf) { (void)hipMemcpyToSymbol(HIP_SYMBOL(dots_trace_buf), &buf, sizeof(buf)); }
#endif

hipError_t launch_dec_embed(hipStream_t s, const int32_t* tokens, const bf16_t* embed, bf16_t* h, int B, int dim) {
    if (dim % 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(dec_embed_kernel, dim3(B), dim3(256), 0, s, tokens, embed, h, dim);
    return hipGetLastError();
}

// Wd / wscale of the launchers: wscale == nullptr -> Wd is the bf16 fragment image (launch_pack_frag*), else Wd is the e4m3
// fragment image (launch_pack_frag_fp8) and wscale its per-row fp32 scales.
// CUs a decode stream may use: the partition it is masked to, or the device.  DOTS_OCR_DEC_WIDE_CUS overrides (tests, A/B runs; read per call).
static int wide_cus(int part_cus) {
    if (const char* e = getenv("DOTS_OCR_DEC_WIDE_CUS")) { const int v = atoi(e); if (v > 0) return v; }
    if (part_cus > 0) return part_cus;
    static int n_cus = 0;
    if (n_cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) { (void)hipGetLastError(); v = 256; }
        n_cus = v;
    }
    return n_cus;
}
// batches above 16 rows run the wide kernels (DOTS_OCR_DEC_WIDE=0: the per-tile kernels, A/B switch; same bits)
static bool wide_on() {
    static const bool on = !(getenv("DOTS_OCR_DEC_WIDE") && atoi(getenv("DOTS_OCR_DEC_WIDE")) == 0);
    return on;
}

// part_cus > 0: the stream is CU-masked to that many CUs (the decode partition of the pipelined step): whole 16-row weight tiles per workgroup
// (half as many workgroups) at B <= 16, as many features per workgroup as one round on those CUs needs above.
hipError_t launch_dec_qkv(hipStream_t s, const bf16_t* h, const bf16_t* ln_w, const void* Wd, const float* wscale, const bf16_t* bias,
                          const float* inv_freq, const int32_t* ctx_len, const int32_t* block_table, int max_pages,
                          bf16_t* pool_layer, bf16_t* q_out, int B, int H, int Hq, int Hkv, float eps, int part_cus, bf16_t* xn,
                          const float* pend, const float* pend_scale) {
    if (H % 32 || H > 512 * NC_MAX || B < 1 || B > MAX_DECODE_ROWS) return hipErrorInvalidValue;
    const int full_tiles = part_cus > 0;
    {   // a pending K-split residual update that the X-image path below will not fuse into its norm launch gets a launch of its own
        static const bool ximg_on0 = !(getenv("DOTS_OCR_QKV_XIMG") && atoi(getenv("DOTS_OCR_QKV_XIMG")) == 0);
        if (pend && !(B > 16 && wide_on() && xn && B > 32 && ximg_on0)) {
            HIP_CHECK_RET(launch_dec_norm_ximg(s, h, nullptr, nullptr, B, H, eps, pend, pend_scale));
            pend = nullptr;
        }
    }
    if (B > 16 && wide_on()) {
        static uint32_t attr_w[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        // round 6, above 32 rows: the rows are normalised once by a small kernel (X image in xn) instead of by every workgroup
        static const bool ximg_on = !(getenv("DOTS_OCR_QKV_XIMG") && atoi(getenv("DOTS_OCR_QKV_XIMG")) == 0);     // A/B switch
        const bool ximg = xn && B > 32 && ximg_on;
        if (ximg) HIP_CHECK_RET(launch_dec_norm_ximg(s, h, ln_w, xn, B, H, eps, pend, pend_scale));
        const int n_out = (Hq + 2 * Hkv) * 8, cus = wide_cus(part_cus);
        const int nm = (n_out + cus - 1) / cus >= 2 ? 2 : 1, tt = B <= 32 ? 2 : 4;
        const size_t lds_w = (size_t)16 * nm * tt * 64 * sizeof(f32x4) + MAX_DECODE_ROWS * sizeof(float);
        const dim3 grid_w(1, (n_out + nm - 1) / nm);
        auto go = [&](auto kern, auto wd, uint32_t* done) -> hipError_t {
            hipError_t e = ensure_lds(kern, lds_w, done);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, grid_w, dim3(1024), lds_w, s, h, ln_w, wd, wscale, bias, inv_freq, ctx_len, block_table, max_pages, pool_layer, q_out, B, H, Hq, Hkv, eps);
            return hipGetLastError();
        };
#define QKV_WIDE(NMV, TTV, IDX)                                                                                                     \
        return wscale ? go(dec_qkv_wide_kernel<NMV, TTV, NC_MAX, u32x2>, (const u32x2*)Wd, &attr_w[IDX])                              \
                      : go(dec_qkv_wide_kernel<NMV, TTV, NC_MAX, bf16x8>, (const bf16x8*)Wd, &attr_w[IDX + 1])
        if (ximg) {          // tt == 4
            const bf16_t* hx = xn;
            auto gox = [&](auto kern, auto wd, uint32_t* done) -> hipError_t {
                hipError_t e = ensure_lds(kern, lds_w, done);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(kern, grid_w, dim3(1024), lds_w, s, hx, ln_w, wd, wscale, bias, inv_freq, ctx_len, block_table, max_pages, pool_layer, q_out, B, H, Hq, Hkv, eps);
                return hipGetLastError();
            };
            if (nm == 1) return wscale ? gox(dec_qkv_wide_kernel<1, 4, NC_MAX, u32x2, true>, (const u32x2*)Wd, &attr_w[8]) : gox(dec_qkv_wide_kernel<1, 4, NC_MAX, bf16x8, true>, (const bf16x8*)Wd, &attr_w[9]);
            return wscale ? gox(dec_qkv_wide_kernel<2, 4, NC_MAX, u32x2, true>, (const u32x2*)Wd, &attr_w[10]) : gox(dec_qkv_wide_kernel<2, 4, NC_MAX, bf16x8, true>, (const bf16x8*)Wd, &attr_w[11]);
        }
        if (nm == 1 && tt == 2) { QKV_WIDE(1, 2, 0); }
        if (nm == 1) { QKV_WIDE(1, 4, 2); }
        if (tt == 2) { QKV_WIDE(2, 2, 4); }
        QKV_WIDE(2, 4, 6);
#undef QKV_WIDE
    }
    static uint32_t attr[4] = {0, 0, 0, 0};
    const int XR = B <= 8 ? 8 : 16;
    const size_t lds = (size_t)XR * H * 2 + 16 * 64 * sizeof(f32x4), lds_max = (size_t)16 * H * 2 + 16 * 64 * sizeof(f32x4);
    const dim3 grid((B + 15) / 16, (Hq + 2 * Hkv) * (full_tiles ? 8 : 16));
    auto go = [&](auto kern, auto wd, uint32_t* done) -> hipError_t {
        hipError_t e = ensure_lds(kern, lds_max, done);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(1024), lds, s, h, ln_w, wd, wscale, bias, inv_freq, ctx_len, block_table, max_pages, pool_layer, q_out, B, H, Hq, Hkv, eps, XR);
        return hipGetLastError();
    };
    if (wscale) return full_tiles ? go(dec_qkv_kernel<NC_MAX, u32x2, true>, (const u32x2*)Wd, &attr[3]) : go(dec_qkv_kernel<NC_MAX, u32x2, false>, (const u32x2*)Wd, &attr[1]);
    return full_tiles ? go(dec_qkv_kernel<NC_MAX, bf16x8, true>, (const bf16x8*)Wd, &attr[2]) : go(dec_qkv_kernel<NC_MAX, bf16x8, false>, (const bf16x8*)Wd, &attr[0]);
}

// part_cus: see launch_dec_qkv
hipError_t launch_dec_proj(hipStream_t s, const bf16_t* X, const void* Wd, const float* wscale, bf16_t* h, int B, int N, int K, int part_cus,
                           float* part, bool* pending) {
    if (N % 16 || K % 32 || K / 32 < 16 || B < 1 || B > MAX_DECODE_ROWS) return hipErrorInvalidValue;
    if (pending) *pending = false;
    if (part && pending && wide_on() && dec_proj_ksplit_supports(B, N, K)) {         // round 6: four K quarters, the residual update is the consumer's (decode_b64.hip)
        *pending = true;
        return launch_dec_proj_ksplit(s, X, Wd, wscale != nullptr, part, B, N, K, wide_cus(part_cus));
    }
    const int full_tiles = part_cus > 0;
    if (B > 16 && wide_on()) {
        static uint32_t attr_w[16] = {0};
        const int n_units = N / 8, cus = wide_cus(part_cus);
        const bool one = (K / 32 + 15) / 16 <= WIDE_G;                            // a slice is at most WIDE_G k-steps: one round
        const int nu = std::max(1, std::min(4, (n_units + cus - 1) / cus)), nm = (nu + 1) / 2, tt = B <= 32 ? 2 : 4;
        const size_t lds_w = (size_t)16 * nm * tt * 64 * sizeof(f32x4);
        const dim3 grid_w(1, (n_units + nu - 1) / nu);
        auto go = [&](auto kern, auto wd, uint32_t* done) -> hipError_t {
            hipError_t e = ensure_lds(kern, lds_w, done);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, grid_w, dim3(1024), lds_w, s, X, wd, wscale, h, B, N, K, nu);
            return hipGetLastError();
        };
#define PROJ_WIDE(NMV, TTV, IDX)                                                                                                  \
        if (one) return wscale ? go(dec_proj_wide_kernel<NMV, TTV, u32x2, true>, (const u32x2*)Wd, &attr_w[8 + IDX])               \
                               : go(dec_proj_wide_kernel<NMV, TTV, bf16x8, true>, (const bf16x8*)Wd, &attr_w[8 + IDX + 1]);         \
        return wscale ? go(dec_proj_wide_kernel<NMV, TTV, u32x2, false>, (const u32x2*)Wd, &attr_w[IDX])                            \
                      : go(dec_proj_wide_kernel<NMV, TTV, bf16x8, false>, (const bf16x8*)Wd, &attr_w[IDX + 1])
        if (nm == 1 && tt == 2) { PROJ_WIDE(1, 2, 0); }
        if (nm == 1) { PROJ_WIDE(1, 4, 2); }
        if (tt == 2) { PROJ_WIDE(2, 2, 4); }
        PROJ_WIDE(2, 4, 6);
#undef PROJ_WIDE
    }
    const int need = (K / 32 + 15) / 16, XR = B <= 8 ? 8 : 16;
    const dim3 grid((B + 15) / 16, full_tiles ? N / 16 : N / 8);
#define PROJ_LAUNCH(G, F)                                                                                                                        \
    do {                                                                                                                                         \
        if (wscale) hipLaunchKernelGGL((dec_proj_kernel<G, u32x2, F>), grid, dim3(1024), 0, s, X, (const u32x2*)Wd, wscale, h, B, N, K, XR);      \
        else hipLaunchKernelGGL((dec_proj_kernel<G, bf16x8, F>), grid, dim3(1024), 0, s, X, (const bf16x8*)Wd, wscale, h, B, N, K, XR);           \
    } while (0)
#define PROJ_CASE(G) do { if (full_tiles) PROJ_LAUNCH(G, true); else PROJ_LAUNCH(G, false); } while (0)
    static const bool no_lds = getenv("DOTS_OCR_PROJ_REG") != nullptr;            // A/B switch: the register-round kernel for long K too
    const size_t lds_x = (size_t)16 * K + 16 * 64 * sizeof(f32x4);
    if (!no_lds && B <= 8 && need > 4 && need <= PROJ_LDS_G && lds_x <= 160 * 1024) {
        static uint32_t attr_l[4] = {0, 0, 0, 0};
        auto go = [&](auto kern, auto wd, uint32_t* done) -> hipError_t {
            hipError_t e = ensure_lds(kern, lds_x, done);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, grid, dim3(1024), lds_x, s, X, wd, wscale, h, B, N, K);
            return hipGetLastError();
        };
        if (wscale) return full_tiles ? go(dec_proj_lds_kernel<u32x2, true>, (const u32x2*)Wd, &attr_l[3]) : go(dec_proj_lds_kernel<u32x2, false>, (const u32x2*)Wd, &attr_l[1]);
        return full_tiles ? go(dec_proj_lds_kernel<bf16x8, true>, (const bf16x8*)Wd, &attr_l[2]) : go(dec_proj_lds_kernel<bf16x8, false>, (const bf16x8*)Wd, &attr_l[0]);
    }
    if (need <= 1) PROJ_CASE(1);
    else if (need <= 2) PROJ_CASE(2);
    else if (need <= 3) PROJ_CASE(3);
    else if (need <= 4) PROJ_CASE(4);
    else if (wscale) { if (full_tiles) PROJ_LAUNCH(PROJ_G_FP8, true); else PROJ_LAUNCH(PROJ_G_FP8, false); }
    else PROJ_CASE(6);
#undef PROJ_CASE
#undef PROJ_LAUNCH
    return hipGetLastError();
}

// part_cus > 0: the stream is CU-masked to that many CUs (the decode partition of the pipelined step): the grid is capped at what those CUs
// hold at once (occupancy query, cached per kernel) and the kernel walks the remaining pair groups.
template <typename Kern>
static int resident_blocks_per_cu(Kern kern, int threads, size_t lds, int* cache) {
    int v = __atomic_load_n(cache, __ATOMIC_ACQUIRE);
    if (v > 0) return v;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kern), threads, lds) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 1; }
    __atomic_store_n(cache, n, __ATOMIC_RELEASE);
    return n;
}

template <typename WT>
static hipError_t gateup_launch(hipStream_t s, const bf16_t* h, const bf16_t* ln_w, const WT* W13d, const float* wscale, bf16_t* act,
                                int B, int H, int I, float eps, int part_cus) {
    static uint32_t attr[2] = {0, 0};
    const int XR = B <= 8 ? 8 : 16;
    const int v = B <= 8 ? 0 : 1;
    static const int cap_env = [] { const char* e = getenv("DOTS_OCR_GATEUP_WG_CAP"); return e ? atoi(e) : 0; }();     // tests / A-B runs: force the walking path
    const int tiles = (B + 15) / 16;
    // 2 pairs per workgroup (8 waves, two rows per wave) for the 16-row image: 14.7 vs 16.4 us at B = 16 (four rows per wave otherwise);
    // at B <= 8 one pair per workgroup is faster (13.3 vs 13.9 us at B = 8, 11.9 vs 13.6 at B = 1).  DOTS_OCR_GATEUP_PAIRS=1/2 forces either.
    static const int pairs_env = [] { const char* e = getenv("DOTS_OCR_GATEUP_PAIRS"); return e ? atoi(e) : 0; }();
    const int pairs = pairs_env ? pairs_env : (B > 8 ? 2 : 1);
    static const bool per_tile = getenv("DOTS_OCR_GATEUP_PER_TILE") != nullptr;     // A/B switch: one workgroup set per 16-row tile at every batch size
    // both instantiations (bf16, e4m3) are held to the one-tile kernels bit for bit by tests/test_decode_kernels_gpu.py (round 5: the e4m3 one
    // too); DOTS_OCR_TWO_TILE_FP8=0 is the A/B switch back to the per-tile kernels for fp8 batches above 16 rows
    static const bool fp8_ok = !(getenv("DOTS_OCR_TWO_TILE_FP8") && atoi(getenv("DOTS_OCR_TWO_TILE_FP8")) == 0);
    if (B > 16 && !per_tile && (!wscale || fp8_ok) && (I / 16) % 2 == 0) {          // two batch tiles per workgroup (TT = 2): every weight byte feeds 32 rows
        static uint32_t attr_t = 0;
        static int occ_t = 0;
        const size_t lds_t = (size_t)2 * 16 * H * 2 + 2 * 4 * GU_WAVES * 64 * sizeof(f32x4);
        if (lds_t <= 160 * 1024) {
            auto kern_t = dec_gateup_kernel<4, NC_MAX, WT, 2, 2>;
            hipError_t et = ensure_lds(kern_t, lds_t, &attr_t);
            if (et != hipSuccess) return et;
            const int tiles2 = (B + 31) / 32;
            int gy = I / 32;
            const int cap = cap_env > 0 ? cap_env : (part_cus > 0 ? part_cus * resident_blocks_per_cu(kern_t, 2 * GU_WAVES * 64, lds_t, &occ_t) : 0);
            if (cap > 0) gy = std::max(1, std::min(gy, cap / tiles2));
            hipLaunchKernelGGL(kern_t, dim3(tiles2, gy), dim3(2 * GU_WAVES * 64), lds_t, s, h, ln_w, W13d, wscale, act, B, H, I, eps, 16);
            return hipGetLastError();
        }
    }
    if (pairs == 2 && (I / 16) % 2 == 0) {
        static uint32_t attr2[2] = {0, 0};
        static int occ2[2] = {0, 0};
        const size_t lds2 = (size_t)XR * H * 2 + 4 * GU_WAVES * 64 * sizeof(f32x4), lds2_max = (size_t)16 * H * 2 + 4 * GU_WAVES * 64 * sizeof(f32x4);
        auto kern2 = v == 0 ? dec_gateup_kernel<1, NC_MAX, WT, 2> : dec_gateup_kernel<2, NC_MAX, WT, 2>;
        hipError_t e2 = ensure_lds(kern2, lds2_max, &attr2[v]);
        if (e2 != hipSuccess) return e2;
        int gy = I / 32;
        const int cap = cap_env > 0 ? cap_env : (part_cus > 0 ? part_cus * resident_blocks_per_cu(kern2, 2 * GU_WAVES * 64, lds2, &occ2[v]) : 0);
        if (cap > 0) gy = std::max(1, std::min(gy, cap / tiles));
        hipLaunchKernelGGL(kern2, dim3(tiles, gy), dim3(2 * GU_WAVES * 64), lds2, s, h, ln_w, W13d, wscale, act, B, H, I, eps, XR);
        return hipGetLastError();
    }
    static int occ1[2] = {0, 0};
    const size_t lds = (size_t)XR * H * 2 + 2 * GU_WAVES * 64 * sizeof(f32x4);
    const size_t lds_max = (size_t)16 * H * 2 + 2 * GU_WAVES * 64 * sizeof(f32x4);
    auto kern = v == 0 ? dec_gateup_kernel<2, NC_MAX, WT, 1> : dec_gateup_kernel<4, NC_MAX, WT, 1>;
    hipError_t e = ensure_lds(kern, lds_max, &attr[v]);
    if (e != hipSuccess) return e;
    int gy = I / 16;
    const int cap = cap_env > 0 ? cap_env : (part_cus > 0 ? part_cus * resident_blocks_per_cu(kern, GU_WAVES * 64, lds, &occ1[v]) : 0);
    if (cap > 0) gy = std::max(1, std::min(gy, cap / tiles));
    hipLaunchKernelGGL(kern, dim3(tiles, gy), dim3(GU_WAVES * 64), lds, s, h, ln_w, W13d, wscale, act, B, H, I, eps, XR);
    return hipGetLastError();
}

hipError_t launch_dec_gateup(hipStream_t s, const bf16_t* h, const bf16_t* ln_w, const void* W13d, const float* wscale, bf16_t* act,
                             int B, int H, int I, float eps, int part_cus, bf16_t* xn, const float* pend, const float* pend_scale) {
    if (I % 32 || H % 128 || H > 512 * NC_MAX || H / 32 < GU_WAVES || H / 32 > GU_G * GU_WAVES || B < 1 || B > MAX_DECODE_ROWS) return hipErrorInvalidValue;
    if (xn && dec_stream64_supports(B, H)) return launch_dec_gateup64(s, h, ln_w, W13d, wscale, act, xn, B, H, I, eps, wide_cus(part_cus), pend, pend_scale);      // round 6: four batch tiles per workgroup
    if (pend) HIP_CHECK_RET(launch_dec_norm_ximg(s, h, nullptr, nullptr, B, H, eps, pend, pend_scale));      // the pending K-split residual update, by a launch of its own
    return wscale ? gateup_launch(s, h, ln_w, (const u32x2*)W13d, wscale, act, B, H, I, eps, part_cus)
                  : gateup_launch(s, h, ln_w, (const bf16x8*)W13d, wscale, act, B, H, I, eps, part_cus);
}

template <typename WT>
static hipError_t lmhead_launch(hipStream_t s, const bf16_t* h, const bf16_t* ln_w, const WT* Wd, const float* wscale, float* logits,
                                int B, int H, int V, float eps) {
    if (H % (32 * LmG<WT>::value)) return hipErrorInvalidValue;
    static uint32_t attr = 0, attr2 = 0;
    static const bool per_tile = getenv("DOTS_OCR_LMHEAD_PER_TILE") != nullptr;       // A/B switch: one workgroup set per 16-row tile at every batch size
    static const bool fp8_ok = !(getenv("DOTS_OCR_TWO_TILE_FP8") && atoi(getenv("DOTS_OCR_TWO_TILE_FP8")) == 0);           // see gateup_launch
    if (B > 16 && !per_tile && (!wscale || fp8_ok) && (size_t)32 * H * 2 <= 160 * 1024) {   // two batch tiles per workgroup: each weight byte feeds 32 rows
        hipError_t e2 = ensure_lds(dec_lmhead2_kernel<NC_MAX, WT>, (size_t)32 * H * 2, &attr2);
        if (e2 != hipSuccess) return e2;
        hipLaunchKernelGGL((dec_lmhead2_kernel<NC_MAX, WT>), dim3((B + 31) / 32, (V / 16 + 15) / 16), dim3(1024), (size_t)32 * H * 2, s, h, ln_w, Wd, wscale, logits, B, H, V, eps);
        return hipGetLastError();
    }
    const int XR = B <= 8 ? 8 : 16;
    const size_t lds = (size_t)XR * H * 2;
    hipError_t e = ensure_lds(dec_lmhead_kernel<NC_MAX, WT>, (size_t)16 * H * 2, &attr);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((dec_lmhead_kernel<NC_MAX, WT>), dim3((B + 15) / 16, (V / 16 + 15) / 16), dim3(1024), lds, s, h, ln_w, Wd, wscale, logits, B, H, V, eps, XR);
    return hipGetLastError();
}

hipError_t launch_dec_lmhead(hipStream_t s, const bf16_t* h, const bf16_t* ln_w, const void* Wd, const float* wscale, float* logits,
                             int B, int H, int V, float eps, int part_cus, bf16_t* xn, const float* pend, const float* pend_scale) {
    if (V % 16 || H % 32 || H > 512 * NC_MAX || B < 1 || B > MAX_DECODE_ROWS) return hipErrorInvalidValue;
    if (xn && dec_stream64_supports(B, H) && H % (32 * 8) == 0) return launch_dec_lmhead64(s, h, ln_w, Wd, wscale, logits, xn, B, H, V, eps, wide_cus(part_cus), pend, pend_scale);
    if (pend) HIP_CHECK_RET(launch_dec_norm_ximg(s, h, nullptr, nullptr, B, H, eps, pend, pend_scale));      // the pending K-split residual update, by a launch of its own
    return wscale ? lmhead_launch(s, h, ln_w, (const u32x2*)Wd, wscale, logits, B, H, V, eps)
                  : lmhead_launch(s, h, ln_w, (const bf16x8*)Wd, wscale, logits, B, H, V, eps);
}
