// Decode-step kernels.  One decode step of B <= 16 sequences is HBM-bound:
// it streams every LM weight once (3.09 GB bf16) plus ctx*28 672 B of KV per sequence.
//
// Paged KV cache, pages of 64 tokens, stored in MFMA-FRAGMENT ORDER so that every wave-level load
// in decode attention is one fully contiguous 1 KiB global_load_dwordx4 that lands directly in the
// v_mfma_f32_16x16x32_bf16 A-operand registers (no LDS, no shuffles):
//   pool[layer][page][kv_head][0 = K | 1 = V][8192 bf16]
//   K (key, d):  chunk ((key>>4)*4 + (d>>5))*64 + ((d>>3)&3)*16 + (key&15),  element d&7
//                -> chunk (kg,kk), lane (g,i) = K[16kg+i][32kk+8g .. +7]       (A of S^T = K.Q^T)
//   V (key, d):  chunk ((key>>5)*8 + (d>>4))*64 + (((key&31)>>2)&3)*16 + (d&15), element 4*((key&31)>>4) + (key&3)
//                -> chunk (slab,dg), lane (g,i) = V^T[16dg+i][keys 32slab + {4g..4g+3, 16+4g..16+4g+3}]
//                   which is exactly the key order in which a lane holds P^T after the S^T MFMAs
//                   (A of O^T = V^T.P^T).
//
// Dense layers at M = B <= 16 rows (decode_fused.hip): skinny GEMM on MFMA with the weight tile as the A operand,
// weights streamed straight to VGPRs (guide: "GEMV / M<=16: neither LDS nor glds"), split-K only across the waves of
// one workgroup, reduced through LDS in a fixed order.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "decode_layout.h"
#include "kernels.h"

namespace {
TRACE_DECL

// ------------------------------------------------------------------------------------------------
// prefill -> pages.  grid (tiles, Hkv, 2); K from the rope'd head-major buffer, V from the qkv buffer.
__global__ __launch_bounds__(256) void kv_to_pages_kernel(const bf16_t* __restrict__ k, const bf16_t* __restrict__ qkv,
                                                          const Tile64* __restrict__ tiles, const int32_t* __restrict__ block_table,
                                                          int max_pages, bf16_t* __restrict__ pool, int64_t T, int Hq, int Hkv) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[64 * 136];
    const Tile64 tl = tiles[blockIdx.x];
    const int h = blockIdx.y, which = blockIdx.z, tid = threadIdx.x;
    const int page = block_table[tl.seq * max_pages + tl.page];
    bf16_t* dst = pool + ((size_t)(page * Hkv + h) * 2 + which) * PAGE_ELEMS;
    if (which == 0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int ci = it * 256 + tid;                       // destination chunk
            const int i = ci & 15, g = (ci >> 4) & 3, kk = (ci >> 6) & 3, kg = ci >> 8;
            const int key = kg * 16 + i;
            u32x4 v = {0, 0, 0, 0};
            if (key < tl.n) v = *reinterpret_cast<const u32x4*>(k + ((size_t)h * T + tl.tok0 + key) * 128 + kk * 32 + g * 8);
            *reinterpret_cast<u32x4*>(dst + (size_t)ci * 8) = v;
        }
    } else {
        const int ld = (Hq + 2 * Hkv) * 128;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int item = it * 256 + tid;
            const int tok = item >> 4, c = item & 15;
            u32x4 v = {0, 0, 0, 0};
            if (tok < tl.n) v = *reinterpret_cast<const u32x4*>(qkv + (size_t)(tl.tok0 + tok) * ld + (Hq + Hkv + h) * 128 + c * 8);
            *reinterpret_cast<u32x4*>(&lds[tok * 136 + c * 8]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int ci = it * 256 + tid;
            const int i = ci & 15, g = (ci >> 4) & 3, dg = (ci >> 6) & 7, slab = ci >> 9;
            const int d = dg * 16 + i;
            u32x4 o;
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                const int e0 = 2 * e2, e1 = 2 * e2 + 1;
                const int key0 = slab * 32 + (e0 >> 2) * 16 + g * 4 + (e0 & 3);
                const int key1 = slab * 32 + (e1 >> 2) * 16 + g * 4 + (e1 & 3);
                o[e2] = (uint32_t)lds[key0 * 136 + d] | ((uint32_t)lds[key1 * 136 + d] << 16);
            }
            *reinterpret_cast<u32x4*>(dst + (size_t)ci * 8) = o;
        }
    }
}

// row-major [rows, K] -> fragment order (weights: rows = N, 16-row tiles; inputs: one 16-row tile).
// The first rot_rows rows (the q and k heads of a fused qkv matrix, 128 rows per head) are additionally PERMUTED inside
// their head so that every 16-row tile holds complete RoPE pairs (decode_fused.hip, dec_qkv): destination row 16j + t of
// a head is source feature 8j + (t >> 1) + 64 (t & 1).
__global__ __launch_bounds__(256) void pack_frag_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int rows, int K, int rot_rows) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;          // destination 16-B chunk
    const int KS = K / 32;
    const int64_t tiles = (rows + 15) / 16;
    if (c >= tiles * KS * 64) return;
    const int lane = (int)(c & 63);
    const int64_t t = c >> 6;
    const int ks = (int)(t % KS);
    const int64_t tile = t / KS;
    int64_t row = tile * 16 + (lane & 15);
    if (row < rot_rows) {
        const int rr = (int)(row & 127), jj = rr >> 4, tt = rr & 15;
        row = (row & ~(int64_t)127) + 8 * jj + (tt >> 1) + 64 * (tt & 1);
    }
    u32x4 v = {0, 0, 0, 0};
    if (row < rows) v = *reinterpret_cast<const u32x4*>(src + row * K + ks * 32 + (lane >> 4) * 8);
    *reinterpret_cast<u32x4*>(dst + c * 8) = v;
}

// row-major [rows, K] <-> X image [K/8][XR][8]   (parity-test entry points)
__global__ __launch_bounds__(256) void convert_x_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int rows, int K, int XR, int to_image) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * K) return;
    const int m = idx / K, k = idx % K;
    if (to_image) dst[ximage_off(m, k, XR, K)] = src[idx];
    else dst[idx] = src[ximage_off(m, k, XR, K)];
}

// ------------------------------------------------------------------------------------------------
// Split-KV decode attention.  grid (n_splits, Hkv, B), NW waves; wave w walks pages split*NW + w,
// += NW*n_splits.  All `group` (<= 16) query heads of one kv head share every K/V load (GQA).
// Output per (b, hkv, split): unnormalised O [group][128] fp32, (m, l) [group].
// n_splits is an engine constant (from max_seq_len), so which pages a split sums — and with it every bit of the
// result — depends on the sequence's own context only, not on what else is in the batch or on the schedule.
// Splits past the context's last page exit at once and are not read by the combine kernel.
// Memory round trips: {context length, first page id, q} in one, then the page itself.
// Round 4 (register diet + conflict-free tail; same arithmetic, bit-identical results):
//  * ONE = every wave owns at most one page (n_splits * NW >= max_pages: every context up to 16 k tokens).  O is then not
//    loop-carried, K (16 chunks) and the first V slab (8 chunks) are requested together, the second V slab is requested into the
//    registers K frees once S^T is computed: 24 KiB in flight per wave instead of 32, ~125 instead of 200 registers, so THREE
//    workgroups per CU instead of two — on the 128-CU decode partition of the pipelined step all 368 working workgroups of the
//    bench's batch are resident at once (two dispatch rounds before: 16.8 us per launch vs 11.7 on the whole chip).
//  * the cross-wave combine buffer holds only the `group` live q columns, rows padded to AT_LD floats: one conflict-free
//    ds_write_b128 per (lane, d group) instead of 32 ds_write_b32 that were 16-way bank conflicts (the in-kernel trace showed
//    1.9 us between "pages done" and the barrier behind these writes).
constexpr int AT_LD = 132;

template <int NW, bool ONE>
__global__ __launch_bounds__(NW * 64, ONE && NW <= 4 ? 3 : 2) void decode_attn_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ pool,
                                                          const int32_t* __restrict__ ctx_len, const int32_t* __restrict__ block_table,
                                                          int max_pages, float* __restrict__ part_o, float* __restrict__ part_ml,
                                                          int Hq, int Hkv, int n_splits, float scale_log2e) {
    extern __shared__ __attribute__((aligned(16))) char at_smem[];
    const int split = blockIdx.x, hkv = blockIdx.y, b = blockIdx.z;
    const int group = Hq / Hkv;
    float* lds_o = reinterpret_cast<float*>(at_smem);                    // [NW][group][AT_LD]
    float* lds_m = lds_o + NW * group * AT_LD;                           // [NW][16]
    float* lds_l = lds_m + NW * 16;                                      // [NW][16]
    const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = l & 15, g = l >> 4;
    const int p0 = split * NW + w;
    TRACE(0);
    const int ctx = ctx_len[b] + 1;                       // includes the token appended this step
    int page = block_table[b * max_pages + min(p0, max_pages - 1)];
    // Q fragments (B operand): lane (j = i, g) holds Q[hkv*group + j][32kk + 8g .. +7]; zero rows j >= group
    u32x4 qraw[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        qraw[kk] = *reinterpret_cast<const u32x4*>(q + ((size_t)b * Hq + hkv * group + min(i, group - 1)) * 128 + kk * 32 + g * 8);
    const int n_pages = (ctx + PAGE - 1) / PAGE;
    if (split * NW >= n_pages) return;
    TRACE(1);
    f32x4 o[8];
#pragma unroll
    for (int dg = 0; dg < 8; ++dg) o[dg] = f32x4{0, 0, 0, 0};
    float m_run = -1e30f, l_run = 0.f;

    // one page: S^T = K.Q^T -> online softmax -> O^T += V^T.P^T.  The load / MFMA order is pinned (sched_barrier): K and V slab 0
    // first, V slab 1 only once the K registers are free.
    auto page_step = [&](int p, int pg, bool first) {
        const bf16_t* kp = pool + ((size_t)(pg * Hkv + hkv) * 2) * PAGE_ELEMS;
        const bf16_t* vp = kp + PAGE_ELEMS;
        bf16x8 kf[16], va[8], vb[8];
#pragma unroll
        for (int c = 0; c < 16; ++c) kf[c] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(kp + (size_t)(c * 64 + l) * 8));
        __builtin_amdgcn_sched_barrier(0);                   // K before V: S^T waits for K only (a wave's loads return in order)
#pragma unroll
        for (int c = 0; c < 8; ++c) va[c] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(vp + (size_t)(c * 64 + l) * 8));
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 qf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const u32x4 z = {0, 0, 0, 0};
            qf[kk] = __builtin_bit_cast(bf16x8, i < group ? qraw[kk] : z);
        }
        // S^T[kg] : rows = keys 16kg + 4g + r, col = q head i
        f32x4 s[4];
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            s[kg] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) s[kg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kg * 4 + kk], qf[kk], s[kg], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 8; ++c) vb[c] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(vp + (size_t)((c + 8) * 64 + l) * 8));
        __builtin_amdgcn_sched_barrier(0);
        const int key0 = p * PAGE;
        float mx = -INFINITY;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = key0 + kg * 16 + 4 * g + r;
                s[kg][r] = key < ctx ? s[kg][r] : -INFINITY;
                mx = fmaxf(mx, s[kg][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * scale_log2e);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
        bf16x8 pf[2];
#pragma unroll
        for (int slab = 0; slab < 2; ++slab) {
            u32x4 pk;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float pv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pv[r] = __builtin_amdgcn_exp2f(fmaf(s[slab * 2 + t][r], scale_log2e, -m_new));
                    psum += pv[r];
                }
                pk[t * 2] = pack_bf2(pv[0], pv[1]);
                pk[t * 2 + 1] = pack_bf2(pv[2], pv[3]);
            }
            pf[slab] = __builtin_bit_cast(bf16x8, pk);
        }
        l_run = l_run * alpha + psum;
        if (!first) {            // O is still zero before a wave's first page: nothing to rescale (0 * alpha = 0 exactly)
#pragma unroll
            for (int dg = 0; dg < 8; ++dg)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[dg][r] *= alpha;
        }
        // V chunks arrive in order (slab 0: dg 0..7, then slab 1): consume them in that order
#pragma unroll
        for (int dg = 0; dg < 8; ++dg) o[dg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va[dg], pf[0], o[dg], 0, 0, 0);
#pragma unroll
        for (int dg = 0; dg < 8; ++dg) o[dg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb[dg], pf[1], o[dg], 0, 0, 0);
    };
    if constexpr (ONE) {
        if (p0 < n_pages) page_step(p0, page, true);
    } else {
        for (int p = p0; p < n_pages; p += NW * n_splits) {
            if (p != p0) page = block_table[b * max_pages + p];
            page_step(p, page, p == p0);
        }
    }
    TRACE(2);
    // wave partial: l over the 4 lane groups that share a q column
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (g == 0) { lds_m[w * 16 + i] = m_run; lds_l[w * 16 + i] = l_run; }
    if (i < group) {
#pragma unroll
        for (int dg = 0; dg < 8; ++dg) *reinterpret_cast<f32x4*>(lds_o + (w * group + i) * AT_LD + dg * 16 + 4 * g) = o[dg];     // O^T[d = 16dg+4g+r][q = i]
    }
    __syncthreads();
    TRACE(3);
    // combine the NW waves: thread -> (q head j, d) pairs
    for (int item = threadIdx.x; item < group * 128; item += NW * 64) {
        const int j = item >> 7, d = item & 127;
        float m = lds_m[j];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) m = fmaxf(m, lds_m[ww * 16 + j]);
        float acc = 0.f, lsum = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
            const float f = __builtin_amdgcn_exp2f(lds_m[ww * 16 + j] - m);
            acc += lds_o[(ww * group + j) * AT_LD + d] * f;
            lsum += lds_l[ww * 16 + j] * f;
        }
        const size_t base = (((size_t)b * Hkv + hkv) * n_splits + split) * group + j;
        part_o[base * 128 + d] = acc;
        if (d == 0) { part_ml[base * 2] = m; part_ml[base * 2 + 1] = lsum; }
    }
    TRACE(4);
}

// ------------------------------------------------------------------------------------------------
// Round 5: STREAMING decode attention for many (row, kv head, split) items per CU — the 64-row step of the pipelined bench on its 64-CU
// partition.  The kernel above is one short-lived workgroup per item: two dependent round trips (context / page id / q, then the page),
// 128 KB moved, an LDS combine and 3 KB of partials — 3200 of them at 64 rows x 25 splits.  On the partition that measured 374 MB per
// layer in 263 us = 22 GB/s per CU, while a plain streaming loop on 64 CUs pulls 44-54 GB/s per workgroup (profiles/r02_bw_probe.txt A/A2):
// the workgroups' life cycles, not the memory system, set the rate.  Here ONE resident workgroup per CU (4 waves, one per SIMD) WALKS the
// items gridDim.x apart and keeps its memory pipe busy across them:
//   * a wave's page (K 16 KiB | V 16 KiB, contiguous in the pool) arrives by LDS-DMA (global_load_lds, 32 x 1 KiB per wave, no registers,
//     non-temporal) in the wave's own 32-KiB LDS buffer; the lane-linear fragment order of the pool makes the read-back one conflict-free
//     ds_read_b128 per chunk into the MFMA A-operand registers;
//   * as soon as a page is in registers the DMA of the NEXT item's page is issued into the same buffer, so it flies under this item's
//     MFMAs, softmax, cross-wave combine and partial store; the next item's page id and q rows are fetched one item ahead as well;
//   * items whose split lies past the row's last page are skipped by the (wave-uniform) walk: nothing is launched for them.
// Arithmetic per item is that of decode_attn_kernel<4, true> statement for statement — same MFMA order, same softmax, same fixed-order
// combine of the four waves, same partial layout — so the combine kernel and every bit downstream are unchanged (tests:
// test_decode_kernels_gpu.py::test_decode_attention_stream_equals_per_split_bitwise, test_decode_plans_gpu.py).  Needs ONE (every wave
// owns at most one page) and NW = 4; the launcher falls back to the kernel above otherwise.
// LDS: 4 x 32 KiB page buffers | combine buffer [4][group][AT_LD] | m, l [4][16] each | pages, keys per row [64] each  (~142 KiB: one workgroup per CU)
#ifndef ATTN_DMA_AUX
#define ATTN_DMA_AUX 2      // cache policy bits of the DMA requests: 2 = nt (each KV byte is read once per step)
#endif
constexpr int ST_NW = 4, ST_PAGE_BYTES = 2 * PAGE_ELEMS * 2;

__global__ __launch_bounds__(ST_NW * 64, 1) void decode_attn_stream_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ pool,
                                                                          const int32_t* __restrict__ ctx_len, const int32_t* __restrict__ block_table,
                                                                          int max_pages, float* __restrict__ part_o, float* __restrict__ part_ml,
                                                                          int B, int Hq, int Hkv, int n_splits, float scale_log2e) {
    extern __shared__ __attribute__((aligned(16))) char at_smem[];
    const int group = Hq / Hkv;
    char* pagebuf = at_smem;                                                                  // [ST_NW][ST_PAGE_BYTES]
    float* lds_o = reinterpret_cast<float*>(at_smem + ST_NW * ST_PAGE_BYTES);                 // [ST_NW][group][AT_LD]
    float* lds_m = lds_o + ST_NW * group * AT_LD;                                             // [ST_NW][16]
    float* lds_l = lds_m + ST_NW * 16;                                                        // [ST_NW][16]
    int* lds_np = reinterpret_cast<int*>(lds_l + ST_NW * 16);                                 // [MAX_DECODE_ROWS] pages per row
    int* lds_ctx = lds_np + MAX_DECODE_ROWS;                                                  // [MAX_DECODE_ROWS] keys per row
    const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = l & 15, g = l >> 4;
    if ((int)threadIdx.x < B) {
        const int c = ctx_len[threadIdx.x] + 1;                      // includes the token appended this step
        lds_ctx[threadIdx.x] = c;
        lds_np[threadIdx.x] = (c + PAGE - 1) / PAGE;
    }
    __syncthreads();
    const int total = B * Hkv * n_splits, stride = gridDim.x, per_row = Hkv * n_splits;
    // item -> (row, kv head, split); everything about an item is wave-uniform and kept in scalar registers (readfirstlane on what comes from LDS)
    auto pages_of = [&](int row) { return __builtin_amdgcn_readfirstlane(lds_np[row]); };
    auto next_active = [&](int it) {                                 // first item >= it on this workgroup's walk whose split owns a page
        while (it < total && (it % n_splits) * ST_NW >= pages_of(it / per_row)) it += stride;
        return it;
    };
    char* mybuf = pagebuf + w * ST_PAGE_BYTES;
    // this wave's page of an item: page index split * 4 + w; its DMA is skipped (wave-uniform branch) when the row has no such page
    auto load_page_id = [&](int it) {
        const int row = it / per_row, pi = (it % n_splits) * ST_NW + w;
        return __builtin_amdgcn_readfirstlane(block_table[(size_t)row * max_pages + min(pi, max_pages - 1)]);
    };
    // chunks [c0, c1) of the wave's page: 0-15 = K, 16-31 = V
    auto issue_dma = [&](int it, int pg, int c0 = 0, int c1 = 32) {
        const int row = it / per_row, pi = (it % n_splits) * ST_NW + w;
        if (pi >= pages_of(row)) return;
        const int hkv = (it / n_splits) % Hkv;
        const bf16_t* src = pool + ((size_t)pg * Hkv + hkv) * 2 * PAGE_ELEMS + l * 8;
#pragma unroll
        for (int c = 0; c < 32; ++c)
            if (c >= c0 && c < c1)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c * 512),
                                                 (__attribute__((address_space(3))) void*)(mybuf + c * 1024), 16, 0, ATTN_DMA_AUX);
    };
    auto load_q = [&](int it, u32x4 (&qr)[4]) {
        const int row = it / per_row, hkv = (it / n_splits) % Hkv;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            qr[kk] = *reinterpret_cast<const u32x4*>(q + ((size_t)row * Hq + hkv * group + min(i, group - 1)) * 128 + kk * 32 + g * 8);
    };

    int it = next_active(blockIdx.x);
    if (it >= total) return;
    u32x4 qraw[4];
    load_q(it, qraw);
    issue_dma(it, load_page_id(it));
    while (it < total) {
        const int nxt = next_active(it + stride);
        const int b = it / per_row, hkv = (it / n_splits) % Hkv, split = it % n_splits;
        const int p = split * ST_NW + w;
        const bool has_page = p < pages_of(b);
        // the next item's small operands, one item ahead (clamped to a valid item when the walk is over: loaded, never used)
        const int nx = nxt < total ? nxt : it;
        u32x4 qnext[4];
        const int pg_next = load_page_id(nx);
        load_q(nx, qnext);
        const int ctx = __builtin_amdgcn_readfirstlane(lds_ctx[b]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's page is in its LDS buffer (and everything requested above)
        f32x4 o[8];
#pragma unroll
        for (int dg = 0; dg < 8; ++dg) o[dg] = f32x4{0, 0, 0, 0};
        float m_run = -1e30f, l_run = 0.f;
        if (has_page) {
            bf16x8 kf[16], vf[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) kf[c] = *reinterpret_cast<const bf16x8*>(mybuf + c * 1024 + l * 16);
#ifdef ATTN_SPLIT_ISSUE      // experiment: the K half of the next page is requested as soon as this page's K half is in registers
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (nxt < total) issue_dma(nxt, pg_next, 0, 16);
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int c = 0; c < 16; ++c) vf[c] = *reinterpret_cast<const bf16x8*>(mybuf + (16 + c) * 1024 + l * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the buffer is free: every fragment sits in registers
            __builtin_amdgcn_sched_barrier(0);
#ifdef ATTN_SPLIT_ISSUE
            if (nxt < total) issue_dma(nxt, pg_next, 16, 32);
#else
            if (nxt < total) issue_dma(nxt, pg_next);
#endif
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 qf[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const u32x4 z = {0, 0, 0, 0};
                qf[kk] = __builtin_bit_cast(bf16x8, i < group ? qraw[kk] : z);
            }
            // ---- the arithmetic of decode_attn_kernel<4, true>::page_step(first = true), statement for statement
            f32x4 s[4];
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                s[kg] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) s[kg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kg * 4 + kk], qf[kk], s[kg], 0, 0, 0);
            }
            const int key0 = p * PAGE;
            float mx = -INFINITY;
#pragma unroll
            for (int kg = 0; kg < 4; ++kg)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = key0 + kg * 16 + 4 * g + r;
                    s[kg][r] = key < ctx ? s[kg][r] : -INFINITY;
                    mx = fmaxf(mx, s[kg][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx * scale_log2e);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            float psum = 0.f;
            bf16x8 pf[2];
#pragma unroll
            for (int slab = 0; slab < 2; ++slab) {
                u32x4 pk;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float pv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pv[r] = __builtin_amdgcn_exp2f(fmaf(s[slab * 2 + t][r], scale_log2e, -m_new));
                        psum += pv[r];
                    }
                    pk[t * 2] = pack_bf2(pv[0], pv[1]);
                    pk[t * 2 + 1] = pack_bf2(pv[2], pv[3]);
                }
                pf[slab] = __builtin_bit_cast(bf16x8, pk);
            }
            l_run = l_run * alpha + psum;
#pragma unroll
            for (int dg = 0; dg < 8; ++dg) o[dg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[dg], pf[0], o[dg], 0, 0, 0);
#pragma unroll
            for (int dg = 0; dg < 8; ++dg) o[dg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[8 + dg], pf[1], o[dg], 0, 0, 0);
        } else if (nxt < total) {
            issue_dma(nxt, pg_next);
        }
        // ---- wave partial -> LDS, fixed-order combine of the four waves, partial store (as in the per-split kernel)
        l_run += __shfl_xor(l_run, 16, 64);
        l_run += __shfl_xor(l_run, 32, 64);
        if (g == 0) { lds_m[w * 16 + i] = m_run; lds_l[w * 16 + i] = l_run; }
        if (i < group) {
#pragma unroll
            for (int dg = 0; dg < 8; ++dg) *reinterpret_cast<f32x4*>(lds_o + (w * group + i) * AT_LD + dg * 16 + 4 * g) = o[dg];
        }
        __syncthreads();
        for (int item = threadIdx.x; item < group * 128; item += ST_NW * 64) {
            const int j = item >> 7, d = item & 127;
            float m = lds_m[j];
#pragma unroll
            for (int ww = 1; ww < ST_NW; ++ww) m = fmaxf(m, lds_m[ww * 16 + j]);
            float acc = 0.f, lsum = 0.f;
#pragma unroll
            for (int ww = 0; ww < ST_NW; ++ww) {
                const float f = __builtin_amdgcn_exp2f(lds_m[ww * 16 + j] - m);
                acc += lds_o[(ww * group + j) * AT_LD + d] * f;
                lsum += lds_l[ww * 16 + j] * f;
            }
            const size_t base = (((size_t)b * Hkv + hkv) * n_splits + split) * group + j;
            part_o[base * 128 + d] = acc;
            if (d == 0) { part_ml[base * 2] = m; part_ml[base * 2 + 1] = lsum; }
        }
        __syncthreads();                                             // the combine buffer is rewritten by the next item
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qraw[kk] = qnext[kk];
        it = nxt;
    }
}

// out[b][head*128 + d] = sum_s w_s O_s / sum_s w_s l_s   (grid (Hq, B), block 128).
// ONE memory round trip: the context length, the (m, l) pairs and the first CMB_BATCH partial rows are all requested
// up front, unconditionally (slots of splits without pages hold stale data and are masked with a select, never
// multiplied); the split weights are computed by the first wave (lane = split) while the partial rows are in flight.
constexpr int CMB_BATCH = 32;

__global__ __launch_bounds__(128) void decode_attn_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                                  const int32_t* __restrict__ ctx_len, bf16_t* __restrict__ out, int Hq,
                                                                  int Hkv, int n_splits, int XR, int NW) {
    __shared__ float wts[64];
    __shared__ float inv_l;
    const int head = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    const int group = Hq / Hkv, hkv = head / group, j = head % group;
    const size_t base0 = (((size_t)b * Hkv + hkv) * n_splits) * group + j;     // + s*group
    const int ctx = ctx_len[b];
    const float2 ml = *reinterpret_cast<const float2*>(part_ml + (base0 + (size_t)min(d, n_splits - 1) * group) * 2);
    const float* po = part_o + base0 * 128 + d;
    const size_t stride = (size_t)group * 128;
    float pv[CMB_BATCH];
#pragma unroll
    for (int s = 0; s < CMB_BATCH; ++s) pv[s] = po[(size_t)min(s, n_splits - 1) * stride];
    const int n_used = min(n_splits, ((ctx + 1 + PAGE - 1) / PAGE + NW - 1) / NW);       // splits that own at least one page
    if (d < 64) {
        const float m = d < n_used ? ml.x : -1e30f, l = d < n_used ? ml.y : 0.f;
        const float mg = wave_max(m);
        const float f = d < n_used ? __builtin_amdgcn_exp2f(m - mg) : 0.f;
        wts[d] = f;
        const float lsum = wave_sum(l * f);
        if (d == 0) inv_l = 1.0f / lsum;
    }
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < CMB_BATCH; ++s) acc += s < n_used ? pv[s] * wts[s] : 0.f;      // fixed order: deterministic
    for (int s = CMB_BATCH; s < n_used; ++s) acc += po[(size_t)s * stride] * wts[s];
    out[ximage_off(b, head * 128 + d, XR, Hq * 128)] = f2bf(acc * inv_l);     // X image: the input of the o projection
}

// ------------------------------------------------------------------------------------------------
// greedy step glue: argmax (first index wins ties, like torch.argmax) in two stages, then EOS / length bookkeeping.
constexpr int ARGMAX_CHUNKS = 64;

// Commit one selected token of row (slot) b: append to its output, EOS / length bookkeeping, next-step input.
// A finished row keeps decoding (fixed-shape graph) but its context is frozen, so it rewrites the same KV position for ever
// and can idle in its slot until the host refills it (continuous batching).
DEVI void commit_token(const StepState& st, int b, int tok) {
    const bool done = st.finished[b] != 0;
    if (st.advance_ctx && !done) st.ctx_len[b] += 1;
    if (!done) {
        const int n = st.out_lens[b];
        const int cap = st.max_len ? st.max_len[b] : st.cap;
        st.out_ids[(size_t)b * st.out_stride + n] = tok;
        st.out_lens[b] = n + 1;
        bool eos = false;
        for (int k = 0; k < st.n_eos; ++k) eos = eos || (tok == st.eos_ids[k]);
        if (eos || n + 1 >= cap) st.finished[b] = 1;
    }
    st.cur_tokens[b] = tok;
}

DEVI void argmax_merge(float& best, int& bi, float ov, int oi) {
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
}

// grid (ARGMAX_CHUNKS, B): partial (value, index) of one vocabulary chunk
__global__ __launch_bounds__(256) void argmax_partial_kernel(const float* __restrict__ logits, int V, int ld,
                                                             float* __restrict__ pval, int32_t* __restrict__ pidx) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int c = blockIdx.x, b = blockIdx.y;
    const int per = (V + ARGMAX_CHUNKS - 1) / ARGMAX_CHUNKS;
    const int lo = c * per, hi = min(V, lo + per);
    const float* row = logits + (size_t)b * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = lo + threadIdx.x; i < hi; i += 256) argmax_merge(best, bi, row[i], i);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_merge(best, bi, __shfl_xor(best, o, 64), __shfl_xor(bi, o, 64));
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) argmax_merge(best, bi, sv[k], si[k]);
        pval[b * ARGMAX_CHUNKS + c] = best;
        pidx[b * ARGMAX_CHUNKS + c] = bi;
    }
}

// grid B, one wave: final merge + bookkeeping
__global__ __launch_bounds__(64) void argmax_step_kernel(const float* __restrict__ pval, const int32_t* __restrict__ pidx, StepState st) {
    const int b = blockIdx.x;
    if (st.sel && !st.sel[b]) return;
    float best = pval[b * ARGMAX_CHUNKS + threadIdx.x];
    int bi = pidx[b * ARGMAX_CHUNKS + threadIdx.x];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_merge(best, bi, __shfl_xor(best, o, 64), __shfl_xor(bi, o, 64));
    if (threadIdx.x == 0) commit_token(st, b, bi);
}

// ------------------------------------------------------------------------------------------------
// Temperature / nucleus (top-p) sampling step (the reference's vLLM path samples at T = 0.1, top_p = 1.0,
// parser.py:27-28; SVG at T = 0.9, demo_vllm_svg.py:35-36).  One workgroup per sequence, no sort:
//   e_i = exp((l_i - max) / T);  nucleus = largest threshold tau with  sum_{e_i >= tau} e_i >= top_p * sum_i e_i
//   (28 bisection steps on tau);  token = inverse CDF of u * mass over the kept e_i in index order.
// u comes from a counter-based hash of (seed, sequence slot, position), so a run is reproducible from its seed.
constexpr int SAMPLE_THREADS = 1024;

DEVI float block_reduce_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < SAMPLE_THREADS / 64; ++i) t += red[i];
    return t;
}
DEVI float block_reduce_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int i = 1; i < SAMPLE_THREADS / 64; ++i) t = fmaxf(t, red[i]);
    return t;
}
DEVI uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(SAMPLE_THREADS) void sample_step_kernel(const float* __restrict__ logits, int V, int ld, float inv_temp,
                                                                     float top_p, uint64_t seed, StepState st) {
    __shared__ float red[SAMPLE_THREADS / 64];
    __shared__ float scan[SAMPLE_THREADS];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (st.sel && !st.sel[b]) return;                                // uniform per workgroup
    const float* row = logits + (size_t)b * ld;
    const int C = (V + SAMPLE_THREADS - 1) / SAMPLE_THREADS;         // contiguous chunk per thread: index order is preserved
    const int lo = tid * C, hi = min(V, lo + C);
    float mx = -INFINITY;
    for (int i = lo; i < hi; ++i) mx = fmaxf(mx, row[i]);
    mx = block_reduce_max(mx, red);
    float z = 0.f;
    for (int i = lo; i < hi; ++i) z += __expf((row[i] - mx) * inv_temp);
    z = block_reduce_sum(z, red);
    float tau = 0.f;
    if (top_p < 1.0f) {
        float lo_t = 0.f, hi_t = 1.0f;                               // mass(lo_t) >= target always, mass(hi_t = 1) may not be
        const float target = top_p * z;
        for (int it = 0; it < 28; ++it) {
            const float mid = 0.5f * (lo_t + hi_t);
            float mass = 0.f;
            for (int i = lo; i < hi; ++i) { const float e = __expf((row[i] - mx) * inv_temp); mass += e >= mid ? e : 0.f; }
            mass = block_reduce_sum(mass, red);
            if (mass >= target) lo_t = mid; else hi_t = mid;
        }
        tau = lo_t;
    }
    float part = 0.f;
    for (int i = lo; i < hi; ++i) { const float e = __expf((row[i] - mx) * inv_temp); part += e >= tau ? e : 0.f; }
    scan[tid] = part;
    __syncthreads();
    if (tid == 0) {                                                   // serial scan of 1024 partials: ~1 us, once per step
        const int pos = st.out_lens[b];
        const uint64_t h = splitmix64(seed ^ splitmix64(((uint64_t)b << 32) | (uint32_t)pos));
        const float u = (float)(h >> 40) * (1.0f / 16777216.0f);     // [0, 1)
        float total = 0.f;
        for (int t = 0; t < SAMPLE_THREADS; ++t) total += scan[t];
        const float x = u * total;
        float acc = 0.f;
        int t = 0;
        for (; t < SAMPLE_THREADS - 1; ++t) { if (acc + scan[t] > x) break; acc += scan[t]; }
        int tok = -1;
        const int l2 = t * C, h2 = min(V, l2 + C);
        for (int i = l2; i < h2; ++i) {
            const float e = __expf((row[i] - mx) * inv_temp);
            if (e >= tau) { tok = i; acc += e; if (acc > x) break; }
        }
        if (tok < 0) {                                                // numerical corner: fall back to the arg max
            float best = -INFINITY;
            for (int i = 0; i < V; ++i) if (row[i] > best) { best = row[i]; tok = i; }
        }
        commit_token(st, b, tok);
    }
}

}  // namespace

#ifdef DOTS_TRACE
void dots_trace_set_decode(unsigned long long* buf) { (void)hipMemcpyToSymbol(HIP_SYMBOL(dots_trace_buf), &buf, sizeof(buf)); }
#endif

hipError_t launch_kv_to_pages(hipStream_t s, const bf16_t* k, const bf16_t* qkv, const Tile64* tiles, int n_tiles,
                              const int32_t* block_table, int max_pages, bf16_t* pool_layer, int64_t T, int Hq, int Hkv) {
    if (n_tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(kv_to_pages_kernel, dim3(n_tiles, Hkv, 2), dim3(256), 0, s, k, qkv, tiles, block_table, max_pages,
                       pool_layer, T, Hq, Hkv);
    return hipGetLastError();
}

hipError_t launch_pack_frag(hipStream_t s, const bf16_t* src, bf16_t* dst, int64_t rows, int K) {
    if (K % 32 != 0) return hipErrorInvalidValue;
    const int64_t chunks = (rows + 15) / 16 * (K / 32) * 64;
    hipLaunchKernelGGL(pack_frag_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, src, dst, (int)rows, K, 0);
    return hipGetLastError();
}

hipError_t launch_pack_frag_qkv(hipStream_t s, const bf16_t* src, bf16_t* dst, int Hq, int Hkv, int K) {
    if (K % 32 != 0) return hipErrorInvalidValue;
    const int rows = (Hq + 2 * Hkv) * 128;
    const int64_t chunks = (int64_t)(rows / 16) * (K / 32) * 64;
    hipLaunchKernelGGL(pack_frag_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, src, dst, rows, K, (Hq + Hkv) * 128);
    return hipGetLastError();
}

hipError_t launch_pack_x(hipStream_t s, const bf16_t* src, bf16_t* x, int rows, int K) {
    if (K % 8 != 0 || rows < 1 || rows > MAX_DECODE_ROWS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(convert_x_kernel, dim3((rows * K + 255) / 256), dim3(256), 0, s, src, x, rows, K, rows <= 8 ? 8 : 16, 1);
    return hipGetLastError();
}

hipError_t launch_unpack_x(hipStream_t s, const bf16_t* x, bf16_t* dst, int rows, int K) {
    if (K % 8 != 0 || rows < 1 || rows > MAX_DECODE_ROWS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(convert_x_kernel, dim3((rows * K + 255) / 256), dim3(256), 0, s, x, dst, rows, K, rows <= 8 ? 8 : 16, 0);
    return hipGetLastError();
}

// Waves (= pages in flight) per decode-attention workgroup and with it the KV split: an ENGINE constant, never a function of the
// batch (a sequence's partial sums must depend on its own context only).  DOTS_OCR_ATTN_WAVES overrides it for experiments.
// Measured round 6 (profiles/r06_decode_attn_waves_ab.txt): 1 and 2 waves are slower everywhere; 8 waves gain 1.5 % of the step at 8 rows, lose 6 % at one
// row (svg) and 1.4 % at 64 rows on the 64-CU partition: 4 stays.
int decode_attn_waves() {
    static const int nw = [] {
        const char* e = getenv("DOTS_OCR_ATTN_WAVES");
        const int v = e ? atoi(e) : 4;
        return (v == 1 || v == 2 || v == 8) ? v : 4;
    }();
    return nw;
}
int decode_attn_splits(int max_seq_len) {
    const int pages = (max_seq_len + PAGE - 1) / PAGE, nw = decode_attn_waves();
    return std::max(1, std::min((pages + nw - 1) / nw, 64));
}

// Which decode-attention kernel runs (process-wide; both produce the same bits): the per-split kernel (default), or the streaming kernel
// with one workgroup per CU.  MEASURED (profiles/r05_decode_attn_stream_ab.txt): the streaming kernel is the slower one everywhere — 237 vs
// 120 us per launch at 64 rows on the 64-CU partition, 85 vs 63 us on the whole chip, 15.3 vs 11.2 us at 8 rows — one 32-KiB page in flight
// per wave is less memory parallelism than the 3-4 co-resident short-lived workgroups of the per-split kernel, whatever their life cycle
// costs; requesting the K half early or allocating DMA changed nothing.  It therefore never runs by default: opt-in only.
// part_cus > 0: the stream is CU-masked to that many CUs.  stream_mode: 1 = the streaming kernel wherever it is legal, 0 = never, -1 = the
// process default (DOTS_OCR_ATTN_STREAM=1: wherever legal; DOTS_OCR_ATTN_STREAM_MIN=n: from n items per CU; unset: never).
int decode_attn_stream_wgs(int B, int Hkv, int n_splits, int max_pages, int part_cus, int stream_mode) {
    static const int env_mode = [] { const char* e = getenv("DOTS_OCR_ATTN_STREAM"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
    static const int min_items = [] { const char* e = getenv("DOTS_OCR_ATTN_STREAM_MIN"); return e ? std::max(1, atoi(e)) : 0; }();
    const int mode = stream_mode >= 0 ? stream_mode : (env_mode >= 0 ? env_mode : (min_items > 0 ? -1 : 0));
    if (mode == 0 || decode_attn_waves() != ST_NW || (int64_t)n_splits * ST_NW < max_pages || B > MAX_DECODE_ROWS) return 0;
    static int n_cus = 0;
    if (n_cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) { (void)hipGetLastError(); v = 256; }
        n_cus = v;
    }
    const int cus = part_cus > 0 ? std::min(part_cus, n_cus) : n_cus, items = B * Hkv * n_splits;
    if (mode == 1) return std::min(cus, items);
    return items >= min_items * cus ? cus : 0;
}

hipError_t launch_decode_attn(hipStream_t s, const bf16_t* q, const bf16_t* pool_layer, const int32_t* ctx_len,
                              const int32_t* block_table, int max_pages, float* part_o, float* part_ml,
                              int B, int Hq, int Hkv, int n_splits, float scale, int part_cus, int stream_mode) {
    if (Hq % Hkv != 0 || Hq / Hkv > 16) return hipErrorInvalidValue;
    const float sl = scale * 1.44269504088896340736f;
    const int nw = decode_attn_waves(), group = Hq / Hkv;
    if (const int wgs = decode_attn_stream_wgs(B, Hkv, n_splits, max_pages, part_cus, stream_mode)) {
        static uint32_t attr = 0;
        const size_t lds_s = (size_t)ST_NW * ST_PAGE_BYTES + ((size_t)ST_NW * group * AT_LD + 2 * ST_NW * 16) * sizeof(float) + 2 * MAX_DECODE_ROWS * sizeof(int);
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const uint32_t bit = 1u << (dev & 31);
        if (!(__atomic_load_n(&attr, __ATOMIC_ACQUIRE) & bit)) {            // dynamic LDS above 64 KiB: opted into once per device
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_attn_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s);
            if (e != hipSuccess) return e;
            __atomic_fetch_or(&attr, bit, __ATOMIC_RELEASE);
        }
        hipLaunchKernelGGL(decode_attn_stream_kernel, dim3(wgs), dim3(ST_NW * 64), lds_s, s, q, pool_layer, ctx_len, block_table, max_pages, part_o, part_ml,
                           B, Hq, Hkv, n_splits, sl);
        return hipGetLastError();
    }
    const dim3 grid(n_splits, Hkv, B);
    const size_t lds = ((size_t)nw * group * AT_LD + 2 * nw * 16) * sizeof(float);
    const bool one = (int64_t)n_splits * nw >= max_pages;        // every wave owns at most one page: the light-weight instantiation
#define ATTN_GO(NWV, ONEV) hipLaunchKernelGGL((decode_attn_kernel<NWV, ONEV>), grid, dim3(NWV * 64), lds, s, q, pool_layer, ctx_len, block_table, max_pages, part_o, part_ml, Hq, Hkv, n_splits, sl)
    switch (nw) {
        case 1: if (one) ATTN_GO(1, true); else ATTN_GO(1, false); break;
        case 2: if (one) ATTN_GO(2, true); else ATTN_GO(2, false); break;
        case 8: if (one) ATTN_GO(8, true); else ATTN_GO(8, false); break;
        default: if (one) ATTN_GO(4, true); else ATTN_GO(4, false); break;
    }
#undef ATTN_GO
    return hipGetLastError();
}

hipError_t launch_decode_attn_combine(hipStream_t s, const float* part_o, const float* part_ml, const int32_t* ctx_len, bf16_t* out,
                                      int B, int Hq, int Hkv, int n_splits) {
    hipLaunchKernelGGL(decode_attn_combine_kernel, dim3(Hq, B), dim3(128), 0, s, part_o, part_ml, ctx_len, out, Hq, Hkv, n_splits, B <= 8 ? 8 : 16,
                       decode_attn_waves());
    return hipGetLastError();
}

hipError_t launch_argmax_step(hipStream_t s, const float* logits, int V, int ld, int B, float* pval, int32_t* pidx, const StepState& st) {
    hipLaunchKernelGGL(argmax_partial_kernel, dim3(ARGMAX_CHUNKS, B), dim3(256), 0, s, logits, V, ld, pval, pidx);
    hipLaunchKernelGGL(argmax_step_kernel, dim3(B), dim3(64), 0, s, pval, pidx, st);
    return hipGetLastError();
}

hipError_t launch_sample_step(hipStream_t s, const float* logits, int V, int ld, int B, float temperature, float top_p, uint64_t seed,
                              const StepState& st) {
    if (temperature <= 0.f || top_p <= 0.f) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sample_step_kernel, dim3(B), dim3(SAMPLE_THREADS), 0, s, logits, V, ld, 1.0f / temperature, fminf(top_p, 1.0f), seed, st);
    return hipGetLastError();
}
