// Flash attention (prefill) for gfx950: bidirectional var-len (ViT, SURVEY §2.3 V5 — the single
// biggest cost of a page) and causal GQA (LM prefill, L5).  head_dim = 128, bf16 in/out, fp32
// softmax and accumulation.  MFMA-bound: 4*n^2*128 flops per (sequence, head).
//
// One workgroup = 8 waves = 256 query rows of one (sequence, head) (or 4 waves = 128 rows); each wave owns 32 rows.
// Everything is computed TRANSPOSED so that all per-row softmax state is lane-local:
//     S^T[key][q] = K . Q^T      A = K tile (LDS),   B = Q   (registers, loaded once)
//     O^T[d][q]   = V^T . P^T    A = V^T tile (LDS), B = P^T (registers, straight from S^T)
// v_mfma_f32_32x32x16_bf16's C layout gives lane (q = l&31, hi = l>>5) the keys
// {(r&3) + 8(r>>2) + 4hi} of each 32-key tile.  The PV contraction index is free to enumerate keys
// in any order as long as A and B agree, so instead of shuffling P into "8 consecutive keys per
// lane" (permlane/bpermute) the V^T buffer is WRITTEN with keys permuted inside each 16-group
// (0-3, 8-11, 4-7, 12-15; done by qkv_rope_split / elementwise.hip): P^T registers feed the MFMA
// untouched and V^T fragments are one ds_read_b128 each.
//
// LDS: K tile [64 keys][256 B], 16-B slots XOR (key & 15); V^T tile [128 d][128 B], slots XOR
// ((d >> 1) & 7): both fragment gathers are bank-conflict free.  Double buffered; the next tile is
// fetched into registers before this tile's MFMAs and written to LDS after them (guide T14),
// one barrier per tile.  The loop is software-pipelined by one stage (QK of tile j and PV of tile j-1
// form one 32-MFMA block, then the softmax of tile j); s_setprio around the MFMA blocks measured null.
//
// MODE 1 (8-wave workgroups, default): the kernel needs > 200 VGPRs, so a CU holds ONE workgroup = 2 waves per SIMD, and
// with one workgroup-wide barrier per tile both waves of a SIMD enter their MFMA block together and their softmax
// together (SQ counters: MFMA pipe 52 % busy, waves 35 % parked).  Here waves 4-7 run half a tile period behind waves
// 0-3 (one extra barrier at the start, barriers after the MFMA block and after the softmax): while one wave of a SIMD
// issues its 32 MFMAs the other does its softmax on the VALU.  K/V tiles are then alive for 1.5 periods: three LDS
// buffers (96 KB), tile t staged by the early half during its iteration t-1 and by the late half during t-2.  Together
// with the explicit LDS look-ahead of the MFMA block: 1099 -> 1163 TFLOP/s (same-box A/B, profiles/r01_final2_*);
// either change alone gains nothing.  Bit-identical results (same accumulation order).
#include <algorithm>
#include <cstdlib>
#include <utility>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int KT_BYTES = 64 * 256;    // K tile
constexpr int VT_BYTES = 128 * 128;   // V^T tile
constexpr int BUF_BYTES = KT_BYTES + VT_BYTES;
constexpr float RESCALE_THR = 6.0f;   // log2 units: P <= 64; THR = 0 reproduces the textbook rescale-every-tile

template <bool CAUSAL, int NW, int MODE>
__global__ __launch_bounds__(NW * 64, 2) void flash_attn_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ VT,
    bf16_t* __restrict__ O, const QBlock* __restrict__ blocks, int n_items, int64_t T, int64_t Tpad, int Hq, int group,
    float scale_log2e) {
    // MODE 0: one barrier per tile, hipcc's own LDS-read placement (the round-1 schedule, kept for A/B runs and for the
    // 4-wave variant).  MODE 1: ping-pong halves + LDS fragments requested one MFMA group ahead.
    constexpr bool PINGPONG = MODE == 1;
    static_assert(!PINGPONG || NW == 8, "ping-pong pairs wave w with wave w + 4 on the same SIMD");
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 2 (3 with PINGPONG) x BUF_BYTES

    const QBlock qb = blocks[xcd_remap(blockIdx.x, n_items)];
    const int h = qb.head;
    const int hkv = h / group;
    const int tid = threadIdx.x, l = tid & 63, l31 = l & 31, hi = l >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = qb.n;

    const int qrow = qb.q0 + w * 32 + l31;               // row inside the sequence
    const int qrow_c = min(qrow, n - 1);

    // ---- Q fragments: B operand, lane (q, hi) holds Q[q][16*ks + 8*hi .. +7] ----
    bf16x8 qf[8];
    {
        const bf16_t* qp = Q + ((size_t)h * T + qb.tok0 + qrow_c) * 128 + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
    }

    const bf16_t* Kbase = K + ((size_t)hkv * T + qb.tok0) * 128;
    const bf16_t* Vbase = VT + (size_t)hkv * 128 * Tpad + qb.pad0;

    int n_tiles = (n + 63) >> 6;
    if (CAUSAL) n_tiles = min(n_tiles, ((qb.q0 + NW * 32 - 1) >> 6) + 1);

    // ---- staging (registers): 4 K chunks + 4 V^T chunks of 16 B per thread.  V lags K by one tile (see the loop). ----
    constexpr int NT = NW * 64, IT = 1024 / NT;        // 16-B chunks per thread per tile
    u32x4 kst[IT], vst[IT];
    auto load_k = [&](int j) {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int item = it * NT + tid;
            const int row = min(j * 64 + (item >> 4), n - 1);
            kst[it] = *reinterpret_cast<const u32x4*>(Kbase + (size_t)row * 128 + (item & 15) * 8);
        }
    };
    auto load_v = [&](int j) {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int item = it * NT + tid;
            vst[it] = *reinterpret_cast<const u32x4*>(Vbase + (size_t)(item >> 3) * Tpad + j * 64 + (item & 7) * 8);
        }
    };
    auto write_k = [&](int buf) {
        char* kb = smem + buf * BUF_BYTES;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int item = it * NT + tid;
            const int row = item >> 4, c = item & 15;
            *reinterpret_cast<u32x4*>(kb + row * 256 + ((c ^ (row & 15)) << 4)) = kst[it];
        }
    };
    auto write_v = [&](int buf) {
        char* vb = smem + buf * BUF_BYTES + KT_BYTES;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int item = it * NT + tid;
            const int d = item >> 3, cv = item & 7;
            *reinterpret_cast<u32x4*>(vb + d * 128 + ((cv ^ ((d >> 1) & 7)) << 4)) = vst[it];
        }
    };

    f32x16 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    bf16x8 pf[4];                                    // P^T of the previous tile, consumed one iteration later

    // fragment gather offsets (constant per lane)
    const int k_row_off = l31 * 256;                 // + kt*32*256
    const int k_sw = l31 & 15;
    const int v_row_off = l31 * 128;                 // + dt*32*128
    const int v_sw = (l31 >> 1) & 7;

    auto k_frag = [&](const char* kb, int t, int ks) {
        return *reinterpret_cast<const bf16x8*>(kb + t * 32 * 256 + k_row_off + (((ks * 2 + hi) ^ k_sw) << 4));
    };
    auto v_frag = [&](const char* vb, int dt, int sl) {
        return *reinterpret_cast<const bf16x8*>(vb + dt * 32 * 128 + v_row_off + (((sl * 2 + hi) ^ v_sw) << 4));
    };

    // O^T += V^T(vb) . pf : 4 d tiles x 4 key slabs, hipcc's own placement of the LDS reads
    auto pv = [&](const char* vb) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_frag(vb, dt, sl), pf[sl], o[dt], 0, 0, 0);
    };

    // ONE block of 32 MFMAs with explicit LDS look-ahead: S^T(j) = K(j).Q^T (16) then O^T += V^T(j-1).P^T(j-1) (16), as
    // 8 groups of 4.  The 4 fragments of group g+1 are requested before the MFMAs of group g issue (left alone, hipcc
    // puts each ds_read right in front of its MFMA and a lone wave waits out the LDS latency 16 times per tile); the
    // sched_group_barriers pin that order.  Two groups of look-ahead measured slower (1077 vs 1102 TFLOP/s).  The
    // accumulation order per accumulator is the same as in the plain loops (ks, then key slab).
    auto mfma_block = [&](const char* kb, const char* vb, bool has_pv, f32x16 (&s)[2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
        bf16x8 fr[2][4];
        auto fetch = [&](int g, bf16x8 (&f)[4]) {        // g < 4: K (t, ks) = (i & 1, 2g + (i >> 1));  g >= 4: V^T (dt, sl) = (i, g - 4)
#pragma unroll
            for (int i = 0; i < 4; ++i) f[i] = g < 4 ? k_frag(kb, i & 1, 2 * g + (i >> 1)) : v_frag(vb, i, g - 4);
        };
        const int n_groups = has_pv ? 8 : 4;             // no PV before the first tile
        fetch(0, fr[0]);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g < n_groups) {
                if (g + 1 < n_groups) fetch(g + 1, fr[(g + 1) & 1]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (g < 4) s[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[g & 1][i], qf[2 * g + (i >> 1)], s[i & 1], 0, 0, 0);
                    else o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[g & 1][i], pf[g - 4], o[i], 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // 4 DS reads (the next group) ...
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // ... then this group's 4 MFMAs
            }
        }
    };

    // mask + online softmax of tile j: S^T (fp32) -> P^T (bf16, `pf`), running max / sum, deferred rescale of O
    auto softmax_tile = [&](int j, f32x16 (&s)[2]) {
        // ---- mask (only the ragged last tile / the causal diagonal) ----
        const int key0 = j * 64;
        bool need_mask = (key0 + 64 > n);
        if (CAUSAL) need_mask = need_mask || (key0 + 63 > qb.q0);      // the block's first row decides
        if (need_mask) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    bool ok = key < n;
                    if (CAUSAL) ok = ok && (key <= qrow);
                    s[t][r] = ok ? s[t][r] : -INFINITY;
                }
        }
        // ---- online softmax (per q = lane&31; the two half-waves hold disjoint key subsets) ----
        float mx = s[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // Deferred rescale (guide T13): keep the old running max while it is exceeded by at most 2^RESCALE_THR
        // for every row of the wave; P is then bounded by 2^RESCALE_THR instead of 1 (fp32 sum and bf16's 8-bit
        // exponent have the headroom).  When the branch fires, O (complete up to tile j-1: its PV was issued before)
        // and l — everything still expressed against the old max — are scaled exactly once, before this tile's P is
        // formed against the new max.
        const float m_cand = mx * scale_log2e;
        if (!__all(m_cand - m_run <= RESCALE_THR)) {
            const float m_new = fmaxf(m_run, m_cand);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                u32x4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float p0 = __builtin_amdgcn_exp2f(fmaf(s[t][8 * u + 2 * e], scale_log2e, -m_run));
                    float p1 = __builtin_amdgcn_exp2f(fmaf(s[t][8 * u + 2 * e + 1], scale_log2e, -m_run));
                    psum += p0 + p1;
                    pk[e] = pack_bf2(p0, p1);
                }
                pf[t * 2 + u] = __builtin_bit_cast(bf16x8, pk);
            }
        l_run += psum;
    };

    if constexpr (PINGPONG) {
        // Waves 4-7 ("late") run half a tile period behind waves 0-3: one extra barrier at the start, then a barrier after
        // the MFMA block and one after the softmax.  Tile t (K and V^T together) lives in buffer t % 3; the early half
        // stages its share during its iteration t-1, the late half during its iteration t-2 (tile 1: in the prologue), so
        // every share is in LDS one barrier before the first MFMA block that reads it, and a buffer is overwritten only
        // after both halves' PV of the tile it held.
        const bool late = w >= NW / 2;                   // wave-uniform
        // Static priority for the later-dispatched half (guide "Two waves per SIMD" item 4): waves 4-7 lose VALU arbitration
        // to their older SIMD partners on every segment; one s_setprio before the loop, no per-segment flips.
        // DOTS_FLASH_PRIO (compile-time, tools/build_variant.sh) switches it for A/B runs.
#ifndef DOTS_FLASH_PRIO
#define DOTS_FLASH_PRIO 1
#endif
        if (DOTS_FLASH_PRIO && late) __builtin_amdgcn_s_setprio(1);
        load_k(0);
        load_v(0);
        write_k(0);
        write_v(0);
        if (late && n_tiles > 1) {
            load_k(1);
            load_v(1);
            write_k(1);
            write_v(1);
        }
        __syncthreads();
        if (late) __syncthreads();
        int b_cur = 0, b_prev = 2, b_stage = late ? 2 : 1;          // j % 3, (j - 1) % 3, (j + 1 or 2) % 3
        for (int j = 0; j < n_tiles; ++j) {
            const int t_stage = j + (late ? 2 : 1);
            const bool has_stage = t_stage < n_tiles;
            if (has_stage) {
                load_k(t_stage);
                load_v(t_stage);
            }
            f32x16 s[2];
            mfma_block(smem + b_cur * BUF_BYTES, smem + b_prev * BUF_BYTES + KT_BYTES, j > 0, s);
            __syncthreads();                             // the other half of the workgroup starts its MFMA block
            softmax_tile(j, s);
            if (has_stage) {
                write_k(b_stage);
                write_v(b_stage);
            }
            __syncthreads();
            b_prev = b_cur;
            b_cur = b_cur == 2 ? 0 : b_cur + 1;
            b_stage = b_stage == 2 ? 0 : b_stage + 1;
        }
        pv(smem + b_prev * BUF_BYTES + KT_BYTES);
        if (!late) __syncthreads();                      // pairs with the late half's last barrier
    } else {
        // Software-pipelined by one stage: iteration j issues ONE block of 32 MFMAs — S^T(j) = K(j).Q^T and
        // O^T += V^T(j-1).P^T(j-1) — then runs the softmax of tile j on the VALU.  V(j) is staged during iteration j
        // (it is first read in iteration j+1), K(j+1) too, so two LDS buffers and one barrier per tile suffice:
        //   K buffer (j+1)&1 was last read by QK(j-1), V buffer j&1 by PV(j-2) — both before the previous barrier.
        load_k(0);
        write_k(0);
        __syncthreads();
        for (int j = 0; j < n_tiles; ++j) {
            const bool has_next = (j + 1 < n_tiles);
            if (has_next) load_k(j + 1);
            load_v(j);
            const char* kb = smem + (j & 1) * BUF_BYTES;
            f32x16 s[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k_frag(kb, t, ks), qf[ks], s[t], 0, 0, 0);
            }
            if (j > 0) pv(smem + ((j - 1) & 1) * BUF_BYTES + KT_BYTES);
            softmax_tile(j, s);
            if (has_next) write_k((j + 1) & 1);
            write_v(j & 1);
            __syncthreads();
        }
        pv(smem + ((n_tiles - 1) & 1) * BUF_BYTES + KT_BYTES);
    }

    // ---- epilogue: O = O^T / l ; lane owns row q, d = dt*32 + 8*rq + 4*hi + 0..3 ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qrow < n) {
        bf16_t* op = O + ((size_t)(qb.tok0 + qrow) * Hq + h) * 128;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                u32x2 pk = {pack_bf2(o[dt][4 * rq] * inv, o[dt][4 * rq + 1] * inv),
                            pack_bf2(o[dt][4 * rq + 2] * inv, o[dt][4 * rq + 3] * inv)};
                *reinterpret_cast<u32x2*>(op + dt * 32 + 8 * rq + 4 * hi) = pk;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Round 4: the bidirectional (ViT) kernel as 4 waves x 64 query rows — ONE wave per SIMD owning the whole 512-register file.
// Why: with two 32-row waves per SIMD every MFMA needs its own LDS fragment (32 ds_read_b128 per 32 MFMAs), the K/V
// tiles are staged through registers (8 ds_write_b128 per thread and tile) and a wave's 32-MFMA block took 1 860 cycles instead of
// 1 024.  Here a wave holds TWO 32-row query blocks (A, B): every K / V^T fragment feeds two MFMAs, the tiles arrive by LDS-DMA
// (buffer_load ... lds: no staging registers, no ds_write), ONE barrier per tile, and the softmax is spread instruction by
// instruction over the gaps of a 64-MFMA stream that does not depend on it.  tools/mfma_filler_probe.hip measured what a gap holds
// with one wave per SIMD: 4 VALU instructions behind an MFMA are free (32-33 cycles per MFMA), the 5th costs 4 cycles, 7 cost 14 —
// whatever the register files of the MFMA's operands; and EVERY instruction (SALU, waits, nops, LDS, DMA) takes a slot.
//   tile j:   gaps  0- 7   S(j+1) = K(j+1).Q^T, k steps 0-1            || row maxima of S(j): 4 v_max3 per gap
//             [rare, out of line: mask of a ragged tile; rescale of O and l — O is complete through tile j-1, nothing is pending]
//             gaps  8-31   S(j+1), k steps 2-7                         || P(j) = exp2(S(j) c - m), row sums: 14 instructions per 3 gaps,
//             gaps 32-55   O += V^T(j).P(j), key slabs 0-2             ||   key slab by key slab, each one ready before its product starts
//             gaps 56-63   O += V^T(j).P(j), key slab 3
// Register plan (hipcc left alone keeps every MFMA result in the accumulator file and copies all 64 scores out before the first MFMA
// of the next phase; it also hoists / sinks plain arithmetic over scheduling barriers): the MFMAs and their fillers are `asm
// volatile`, one statement per gap.  S(j), S(j+1) live in arch VGPRs (VGPR-form MFMA), where the softmax reads them in place; Q is
// the AGPR B operand of the score MFMAs and never occupies an arch VGPR; O is an AGPR accumulator.  AGPRs: O 128 + Q 64; VGPRs:
// S 2 x 64, P 32, fragments 16, ~35 others.
// Hazards (nothing inside an asm string is padded by the compiler): a transcendental's result is used two instructions later at the
// earliest; a score read by the softmax was written >= 32 MFMAs earlier; a P word is read by an MFMA >= 1 MFMA after the
// v_cvt_pk that wrote it; MFMAs on the same accumulator are >= 3 MFMAs apart; the A operand of an MFMA comes from LDS (the compiler
// places the lgkmcnt wait in front of the statement that uses it).
// LDS: ONE ring of 4 stages, stage t % 4 = {K(t) 16 KiB | V^T(t) 16 KiB}, swizzles as in the 8-wave kernel but applied on the DMA source
// side (LDS-DMA writes lane-linear).  The barrier of a tile sits in front of its LAST 8 gaps (product MFMAs of key slab 3, no softmax
// work left): behind it the wave requests K(j+4) and V^T(j+3) — one DMA piece per gap, in gaps that have nothing else to carry — and
// reads the first K fragments of tile j+1, so no tile starts with an exposed LDS round trip.  Two tile periods of cover for every
// request.  K rows past a sequence's end are read from the 64 spare rows the K buffer carries (kernels.h) and masked.
constexpr int RING = 4;
constexpr int RING_BYTES = RING * 2 * KT_BYTES;               // KT_BYTES == VT_BYTES == 16 KiB

#define F64_MFMA_S0 "v_mfma_f32_32x32x16_bf16 %[d], %[a], %[b], 0\n\t"
#define F64_MFMA_S "v_mfma_f32_32x32x16_bf16 %[d], %[a], %[b], %[d]\n\t"
#define F64_MAX5 "v_max3_f32 %[mx], %[mx], %[x0], %[x1]\n\tv_max3_f32 %[mx], %[mx], %[x2], %[x3]\n\tv_max3_f32 %[mx], %[mx], %[x4], %[x5]\n\tv_max3_f32 %[mx], %[mx], %[x6], %[x7]\n\tv_max3_f32 %[mx], %[mx], %[x8], %[x9]"
#define F64_MAX6 F64_MAX5 "\n\tv_max3_f32 %[mx], %[mx], %[xa], %[xb]"
// decision chain, part 1 (gap 6): the two half-waves hold disjoint key subsets of the same rows -> exchange by v_permlane32_swap (a
// ds_bpermute would park the wave for an LDS round trip); mx <- c = scaled row maximum of S(j).  The s_nop is the VALU-write -> permlane
// read hazard.  Part 2 (gap 7): need = any row of the wave exceeds its running maximum by more than the threshold (wave-uniform mask);
// m <- need ? max(m, c) : m, branch-free — O and l are rescaled later, in front of the first product MFMA, out of line.
#define F64_CHAIN1 "v_mov_b32 %[t0], %[mx0]\n\tv_mov_b32 %[t1], %[mx1]\n\ts_nop 1\n\tv_permlane32_swap_b32 %[t0], %[mx0]\n\tv_permlane32_swap_b32 %[t1], %[mx1]\n\t" \
                   "v_max_f32 %[mx0], %[t0], %[mx0]\n\tv_max_f32 %[mx1], %[t1], %[mx1]\n\tv_mul_f32 %[mx0], %[sc], %[mx0]\n\tv_mul_f32 %[mx1], %[sc], %[mx1]"
#define F64_CHAIN2 "v_sub_f32 %[t0], %[mx0], %[m0]\n\tv_sub_f32 %[t1], %[mx1], %[m1]\n\tv_cmp_lt_f32 %[k0], %[thr], %[t0]\n\tv_cmp_lt_f32 %[k1], %[thr], %[t1]\n\t" \
                   "v_max_f32 %[t0], %[m0], %[mx0]\n\tv_max_f32 %[t1], %[m1], %[mx1]\n\ts_or_b64 %[k0], %[k0], %[k1]\n\ts_cmp_lg_u64 %[k0], 0\n\ts_cselect_b64 %[k0], -1, 0\n\t" \
                   "v_cndmask_b32 %[m0], %[m0], %[t0], %[k0]\n\tv_cndmask_b32 %[m1], %[m1], %[t1], %[k0]"
// exp stream of two pairs X, Y over three gaps (5 + 5 + 4 instructions): p = exp2(s c - m) -> packed bf16x2 word of P, row sum
#ifdef F64_NO_EXP
#define F64_EXP_A ""
#define F64_EXP_B ""
#define F64_EXP_C ""
#define F64_C_EXTRA
#elif defined(F64_DOT2)
// -DF64_DOT2 (EXPERIMENT): the row sums are taken from the ROUNDED P words, one v_dot2c_f32_bf16 (ps += p.lo * 1 + p.hi * 1) per pair instead of two
// v_add_f32: 12 fillers per three gaps = the measured 4 free slots per MFMA.  l is then the sum of exactly the bf16 values the product MFMAs see.
#define F64_EXP_A "v_fma_f32 %[t0], %[x0], %[sc], -%[m]\n\tv_fma_f32 %[t1], %[x1], %[sc], -%[m]\n\tv_exp_f32_e32 %[t0], %[t0]\n\tv_exp_f32_e32 %[t1], %[t1]"
#define F64_EXP_B "v_cvt_pk_bf16_f32 %[wx], %[t0], %[t1]\n\tv_fma_f32 %[u0], %[y0], %[sc], -%[m]\n\tv_fma_f32 %[u1], %[y1], %[sc], -%[m]\n\tv_exp_f32_e32 %[u0], %[u0]"
#define F64_C_EXTRA , [wxi] "v"(pw[xb][xsl][xe]), [one2] "s"(0x3F803F80u)
#define F64_EXP_C "v_exp_f32_e32 %[u1], %[u1]\n\tv_dot2c_f32_bf16_e32 %[ps], %[one2], %[wxi]\n\tv_cvt_pk_bf16_f32 %[wy], %[u0], %[u1]\n\tv_dot2c_f32_bf16_e32 %[ps], %[one2], %[wy]"
#else
#define F64_EXP_A "v_fma_f32 %[t0], %[x0], %[sc], -%[m]\n\tv_fma_f32 %[t1], %[x1], %[sc], -%[m]\n\tv_exp_f32_e32 %[t0], %[t0]\n\tv_exp_f32_e32 %[t1], %[t1]\n\tv_add_f32_e32 %[ps], %[ps], %[t0]"
#define F64_EXP_B "v_add_f32_e32 %[ps], %[ps], %[t1]\n\tv_cvt_pk_bf16_f32 %[wx], %[t0], %[t1]\n\tv_fma_f32 %[u0], %[y0], %[sc], -%[m]\n\tv_fma_f32 %[u1], %[y1], %[sc], -%[m]\n\tv_exp_f32_e32 %[u0], %[u0]"
#define F64_C_EXTRA
#define F64_EXP_C "v_exp_f32_e32 %[u1], %[u1]\n\tv_add_f32_e32 %[ps], %[ps], %[u0]\n\tv_add_f32_e32 %[ps], %[ps], %[u1]\n\tv_cvt_pk_bf16_f32 %[wy], %[u0], %[u1]"
#endif

// A packed-fp32 form of the exp stream (one v_pk_fma_f32 per score pair: 12 instead of 14 fillers per three gaps) was built in round 4 and
// measured in round 5: 16.6 vs 15.6 ms per launch (profiles/r05_flash_f64pk_ab.txt) — packed fp32 VALU beside MFMAs is an anti-lever on
// this part (MI355X_MICROARCH.md, filler price list).  Removed; the source is kept under experiments/flash_f64_pk/.
// compile-time loop: the gap index must be a constant expression (operand selection by `if constexpr`, never by run-time selects)
template <int... I, class F> DEVI void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> DEVI void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// CAUSAL (round 6: the LM prefill, until now on the 8-wave kernel of rounds 1-3): query row i sees keys 0 .. i of its sequence.  A workgroup's 256 query
// rows need the KV tiles up to their own last row only (the tile count is cut, so the DMA never fetches past it), and a wave masks the tiles that
// reach past ITS first row — per element, in the block that already masks a ragged last tile (keys above a row's own position become -inf before the
// maxima see them; a tile wholly above a wave's rows costs that wave one masked tile: the four waves share the ring and its barriers).  Nothing else in
// the schedule changes: same gaps, same fillers, same bits for the bidirectional instantiation.
template <bool CAUSAL>
__global__ __launch_bounds__(256) void flash_attn64_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ VT,
    bf16_t* __restrict__ O, const QBlock* __restrict__ blocks, int n_items, int64_t T, int64_t Tpad, int Hq, int group,
    float scale_log2e, XcdPlan plan) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // K ring | V^T ring
    // this XCD's chunk of the work list, cut by cost (kernels.h: XcdPlan); workgroups past the chunk have nothing to do
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    if (idx >= plan.cnt[xcd]) return;
    const QBlock qb = blocks[plan.base[xcd] + idx];
    const int h = qb.head, hkv = h / group, n = qb.n;
    const int tid = threadIdx.x, l = tid & 63, l31 = l & 31, hi = l >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_tiles = CAUSAL ? min((n + 63) >> 6, (qb.q0 + 256 + 63) >> 6) : (n + 63) >> 6;      // causal: no key beyond the workgroup's last row
    const int t_last = n_tiles - 1;

    // ---- Q fragments of the two blocks: B operand, lane (q, hi) holds Q[q][16 ks + 8 hi .. +7]
    bf16x8 qf[2][8];
#pragma unroll
    for (int bk = 0; bk < 2; ++bk) {
        const int qrow = qb.q0 + w * 64 + bk * 32 + l31;
        const bf16_t* qp = Q + ((size_t)h * T + qb.tok0 + min(qrow, n - 1)) * 128 + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[bk][ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
    }
    // ---- LDS-DMA (buffer_load ... lds): wave w copies K pieces 4i + w (keys 4p .. 4p+3, 256 B each) and V^T pieces 4i + w (d rows
    // 8p .. 8p+7, 128 B each).  Keys / d rows advance by 16 / 32 per i, which leaves the swizzle terms unchanged: ONE per-lane source
    // offset per operand; tile and piece enter through the scalar offset of the instruction, so a piece costs no VALU.
    const int key0w = 4 * w + (l >> 4), d0w = 8 * w + (l >> 3);
    const int k_src = key0w * 256 + (((l & 15) ^ (key0w & 15)) << 4);
    const int v_src = (int)((size_t)d0w * Tpad * 2) + (((l & 7) ^ ((d0w >> 1) & 7)) << 4);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(K + ((size_t)hkv * T + qb.tok0) * 128), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(VT + (size_t)hkv * 128 * Tpad + qb.pad0), 0, 0x7ffffff0, 0x00020000);
    const int v_step = (int)(64 * Tpad);                          // bytes between d rows 32 apart
    auto dma_k = [&](int t, int i) {                            // piece i (0-3) of K(t) -> stage t % 4; t is clamped to the last tile
        const int tc = min(t, t_last);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (__attribute__((address_space(3))) void*)(smem + (t & 3) * 2 * KT_BYTES + (4 * i + w) * 1024), 16, k_src,
                                                 (tc * 64 + 16 * i) * 256, 0, 0);
    };
    auto dma_v = [&](int t, int i) {                            // piece i (0-3) of V^T(t) -> stage t % 4
        const int tc = min(t, t_last);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void*)(smem + ((t & 3) * 2 + 1) * KT_BYTES + (4 * i + w) * 1024), 16,
                                                 v_src, tc * 128 + i * v_step, 0, 0);
    };
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dma_k(t, i);
            if (t < 3) dma_v(t, i);
        }

    f32x16 o[2][4];
#pragma unroll
    for (int bk = 0; bk < 2; ++bk)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[bk][dt][r] = 0.f;
    float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};

    // fragment gather offsets: ((2 ks + hi) ^ (key & 15)) << 4 splits bitwise into a per-lane constant and ks << 5 — ONE register per
    // operand; a fragment address is (lane constant + ring stage) ^ (ks << 5) (+ 8 KiB / 4 KiB steps as immediates)
    const int k_lane = l31 * 256 + ((((l31 & 15) >> 1) << 5) | ((hi ^ (l31 & 1)) << 4));
    const int vsw = (l31 >> 1) & 7;
    const int v_lane = l31 * 128 + (((vsw >> 1) << 5) | ((hi ^ (vsw & 1)) << 4));
    // (base includes the LDS address of the dynamic segment, a multiple of 16 KiB in this kernel — it has no static LDS — so the XOR with
    // ks << 5 still only touches bits 5-7; addressing by integer avoids one "+ symbol" VALU add per fragment)
    const int lds0 = (int)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto frag = [&](int base, int x, int imm) {
        return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>((uintptr_t)(uint32_t)((base ^ (x << 5)) + imm));
    };


    // ---- prologue: S(0)
    f32x16 sA[2][2], sB[2][2];                                   // S(j) / S(j+1): two VGPR sets, swapped every tile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // once per workgroup: Q, K(0..3), V^T(0..2)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const bf16x8 f0 = frag(lds0 + k_lane, ks, 0), f1 = frag(lds0 + k_lane, ks, 8192);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            if (ks == 0) asm volatile(F64_MFMA_S0 : [d] "=&v"(sA[q4 & 1][q4 >> 1]) : [a] "v"(q4 >> 1 ? f1 : f0), [b] "a"(qf[q4 & 1][0]));
            else asm volatile(F64_MFMA_S : [d] "+v"(sA[q4 & 1][q4 >> 1]) : [a] "v"(q4 >> 1 ? f1 : f0), [b] "a"(qf[q4 & 1][ks]));
        }
    }

#ifdef F64_PROF
    unsigned long long prof[6] = {0, 0, 0, 0, 0, 0};
#define F64_STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); prof[i] += now_ - last_; last_ = now_; } while (0)
#else
#define F64_STAMP(i)
#endif
    bf16x8 fa[2][2];                                              // fragment pairs, one step ahead (carried from tile to tile)
    fa[0][0] = frag(lds0 + k_lane + 2 * KT_BYTES, 0, 0);         // K(1), k step 0
    fa[0][1] = frag(lds0 + k_lane + 2 * KT_BYTES, 0, 8192);
    // one tile: sc = S(j) (complete), sn <- S(j+1)
    auto tile_body = [&](int j, f32x16 (&sc)[2][2], f32x16 (&sn)[2][2]) {
#ifdef F64_PROF
        unsigned long long last_ = __builtin_amdgcn_s_memtime();
#endif
        const int kl = lds0 + k_lane + ((j + 1) & 3) * 2 * KT_BYTES, vl = lds0 + v_lane + ((j & 3) * 2 + 1) * KT_BYTES;
        // fa[0] = the first K fragments of this tile: read behind the previous tile's barrier
        const int key0 = j * 64;
        if (key0 + 64 > n || (CAUSAL && key0 + 63 > qb.q0 + w * 64)) {   // ragged last tile / the padding tile of an odd count (rare): keys past the end
#pragma unroll                                                    // hold whatever the spare K rows held -> -inf before the maxima see them;
            for (int bk = 0; bk < 2; ++bk) {                      // causal: the tiles on and above this wave's diagonal (wave-uniform condition)
                const int qrow = qb.q0 + w * 64 + bk * 32 + l31;
#pragma unroll
                for (int tq = 0; tq < 2; ++tq)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = key0 + tq * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        sc[bk][tq][r] = (key < n && (!CAUSAL || key <= qrow)) ? sc[bk][tq][r] : -INFINITY;
                    }
            }
        }
        // ---- gaps 0-5: score MFMAs || row maxima of S(j) (6 + 5 + 5 v_max3 per block); gaps 6-7: the decision chain
        float mx[2] = {-INFINITY, -INFINITY};
        const float m_old[2] = {m_run[0], m_run[1]};
        unsigned long long need = 0;
        static_for<8>([&sn, &sc, &mx, &qf, &fa, &frag, &m_run, &need, kl, scale_log2e](auto gc) {   // explicit captures: clang does not capture a variable that is used by asm operands only
            constexpr int g = decltype(gc)::value;
            constexpr int ks = g >> 2, q4 = g & 3, bk = q4 & 1, tt = q4 >> 1, cur = ks & 1;
            if constexpr (q4 == 0) { fa[cur ^ 1][0] = frag(kl, ks + 1, 0); fa[cur ^ 1][1] = frag(kl, ks + 1, 8192); }
            if constexpr (g < 6) {
                constexpr int mb = g / 3, v0 = (g % 3 == 0) ? 0 : (g % 3 == 1 ? 12 : 22);       // scores v0 .. of block mb, flattened [t][r]
                const f32x16 (&xs)[2] = sc[mb];
#define XV(i) xs[(v0 + (i)) >> 4][(v0 + (i)) & 15]
                if constexpr (g % 3 == 0) {
                    if constexpr (ks == 0)
                        asm volatile(F64_MFMA_S0 F64_MAX6 : [d] "=&v"(sn[bk][tt]), [mx] "+v"(mx[mb])
                                     : [a] "v"(fa[cur][tt]), [b] "a"(qf[bk][0]), [x0] "v"(XV(0)), [x1] "v"(XV(1)), [x2] "v"(XV(2)), [x3] "v"(XV(3)), [x4] "v"(XV(4)), [x5] "v"(XV(5)),
                                       [x6] "v"(XV(6)), [x7] "v"(XV(7)), [x8] "v"(XV(8)), [x9] "v"(XV(9)), [xa] "v"(XV(10)), [xb] "v"(XV(11)));
                    else
                        asm volatile(F64_MFMA_S F64_MAX6 : [d] "+v"(sn[bk][tt]), [mx] "+v"(mx[mb])
                                     : [a] "v"(fa[cur][tt]), [b] "a"(qf[bk][ks]), [x0] "v"(XV(0)), [x1] "v"(XV(1)), [x2] "v"(XV(2)), [x3] "v"(XV(3)), [x4] "v"(XV(4)), [x5] "v"(XV(5)),
                                       [x6] "v"(XV(6)), [x7] "v"(XV(7)), [x8] "v"(XV(8)), [x9] "v"(XV(9)), [xa] "v"(XV(10)), [xb] "v"(XV(11)));
                } else {
                    if constexpr (ks == 0)
                        asm volatile(F64_MFMA_S0 F64_MAX5 : [d] "=&v"(sn[bk][tt]), [mx] "+v"(mx[mb])
                                     : [a] "v"(fa[cur][tt]), [b] "a"(qf[bk][0]), [x0] "v"(XV(0)), [x1] "v"(XV(1)), [x2] "v"(XV(2)), [x3] "v"(XV(3)), [x4] "v"(XV(4)), [x5] "v"(XV(5)),
                                       [x6] "v"(XV(6)), [x7] "v"(XV(7)), [x8] "v"(XV(8)), [x9] "v"(XV(9)));
                    else
                        asm volatile(F64_MFMA_S F64_MAX5 : [d] "+v"(sn[bk][tt]), [mx] "+v"(mx[mb])
                                     : [a] "v"(fa[cur][tt]), [b] "a"(qf[bk][ks]), [x0] "v"(XV(0)), [x1] "v"(XV(1)), [x2] "v"(XV(2)), [x3] "v"(XV(3)), [x4] "v"(XV(4)), [x5] "v"(XV(5)),
                                       [x6] "v"(XV(6)), [x7] "v"(XV(7)), [x8] "v"(XV(8)), [x9] "v"(XV(9)));
                }
#undef XV
            } else if constexpr (g == 6) {
                float c0, c1;
                asm volatile(F64_MFMA_S F64_CHAIN1 : [d] "+v"(sn[bk][tt]), [mx0] "+v"(mx[0]), [mx1] "+v"(mx[1]), [t0] "=&v"(c0), [t1] "=&v"(c1)
                             : [a] "v"(fa[cur][tt]), [b] "a"(qf[bk][ks]), [sc] "s"(scale_log2e));
            } else {
                float c0, c1;
                unsigned long long k1;
                asm volatile(F64_MFMA_S F64_CHAIN2 : [d] "+v"(sn[bk][tt]), [m0] "+v"(m_run[0]), [m1] "+v"(m_run[1]), [t0] "=&v"(c0), [t1] "=&v"(c1), [k0] "=&s"(need), [k1] "=&s"(k1)
                             : [a] "v"(fa[cur][tt]), [b] "a"(qf[bk][ks]), [mx0] "v"(mx[0]), [mx1] "v"(mx[1]), [thr] "s"(RESCALE_THR) : "scc");
            }
        });
        F64_STAMP(1);                                             // gaps 0-7
        // ---- gaps 8-55: P(j) spread over the score MFMAs of k steps 2-7 and the product MFMAs of key slabs 0-2; gaps 56-63: slab 3.
        // In front of the first product MFMA (gap 32), out of line and rare: the rescale of O and l (O is complete through tile j-1).
        float ps[2] = {0.f, 0.f};
        uint32_t pw[2][4][4];                                     // P(j): [block][key slab][word]
        float t0 = 0.f, t1 = 0.f, u0 = 0.f, u1 = 0.f;             // the two pairs in flight
        static_for<56>([&sn, &sc, &qf, &fa, &o, &pw, &ps, &m_run, &m_old, &l_run, &need, &t0, &t1, &u0, &u1, &frag, &dma_k, &dma_v, kl, vl, j, lds0, k_lane, scale_log2e](auto gc) {
            constexpr int g = decltype(gc)::value + 8;
            if constexpr (g == 32) {
                if (need) {                                       // wave-uniform
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) {
                        const float alpha = __builtin_amdgcn_exp2f(m_old[rb] - m_run[rb]);
                        l_run[rb] *= alpha;
#pragma unroll
                        for (int rd = 0; rd < 4; ++rd) {
                            // the empty volatile statement "redefines" the accumulator inside this block: without it hipcc copies all 128 O
                            // registers out of the accumulator file ABOVE the branch, on every tile
                            asm volatile("" : "+a"(o[rb][rd]));
#pragma unroll
                            for (int r = 0; r < 16; ++r) o[rb][rd][r] *= alpha;
                        }
                    }
                }
            }
            constexpr bool is_qk = g < 32;
            constexpr int q4 = g & 3, bk = q4 & 1, hf = q4 >> 1;                           // block, key half (scores) / d tile parity (product)
            constexpr int ks = is_qk ? (g >> 2) : 0;                                        // score MFMA: k step
            constexpr int st = is_qk ? 0 : ((g - 32) >> 2), sl = st >> 1, dt = 2 * (st & 1) + hf;   // product MFMA: step, key slab, d tile
            constexpr int step = g >> 2, cur = step & 1;                                    // 16 steps of 4 MFMAs share 2 fragments
            if constexpr (g == 56) {
                // The tile's barrier, in front of its last 8 gaps.  Before it: this wave's pieces of K(j+2) and V^T(j+1) — requested two tiles
                // ago — have landed (the 8 pieces of the previous tile's tail may still fly).  Behind it every wave is done with K(j+1) and
                // with V^T(j-1), K(j): the stages the requests of this tail overwrite; V^T(j) (key slab 3 is still to come) is not touched.
#ifndef F64_NO_BAR
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#endif
            }
            if constexpr (q4 == 0) {                                                        // next step's fragments
                constexpr int ns = step + 1;
                if constexpr (ns < 8) { fa[cur ^ 1][0] = frag(kl, ns, 0); fa[cur ^ 1][1] = frag(kl, ns, 8192); }
                else if constexpr (ns < 16) { constexpr int s2 = ns - 8; fa[cur ^ 1][0] = frag(vl, s2 >> 1, (2 * (s2 & 1)) * 4096); fa[cur ^ 1][1] = frag(vl, s2 >> 1, (2 * (s2 & 1) + 1) * 4096); }
                else {                                                                      // g == 60: the first K fragments of the NEXT tile (K(j+2), visible since the barrier)
                    const int kn = lds0 + k_lane + ((j + 2) & 3) * 2 * KT_BYTES;
                    fa[cur ^ 1][0] = frag(kn, 0, 0);
                    fa[cur ^ 1][1] = frag(kn, 0, 8192);
                }
            }
#ifndef F64_NO_DMA
            if constexpr (g >= 56 && g < 60) dma_k(j + 4, g - 56);
            if constexpr (g >= 60) dma_v(j + 3, g - 60);
#endif
            if constexpr (g >= 56) {
                const bf16x8 pf = __builtin_bit_cast(bf16x8, u32x4{pw[bk][sl][0], pw[bk][sl][1], pw[bk][sl][2], pw[bk][sl][3]});
                asm volatile("v_mfma_f32_32x32x16_bf16 %[d], %[a], %[b], %[d]" : [d] "+a"(o[bk][dt]) : [a] "v"(fa[cur][hf]), [b] "v"(pf));
            } else {
                // the exp stream: gaps 8-55 in triples, two pairs per triple; pair order: key slab, block, word
                constexpr int tr = (g - 8) / 3, ph = (g - 8) % 3;
                constexpr int qx = 2 * tr, qy = 2 * tr + 1;                                 // pairs 0..31
                constexpr int xb = (qx >> 2) & 1, xsl = qx >> 3, xe = qx & 3, yb = (qy >> 2) & 1, ysl = qy >> 3, ye = qy & 3;
                // scores of pair (block b, slab s, word e): S[b][s >> 1][8 (s & 1) + 2 e], + 1
                const float xs0 = sc[xb][xsl >> 1][8 * (xsl & 1) + 2 * xe], xs1 = sc[xb][xsl >> 1][8 * (xsl & 1) + 2 * xe + 1];
                const float ys0 = sc[yb][ysl >> 1][8 * (ysl & 1) + 2 * ye], ys1 = sc[yb][ysl >> 1][8 * (ysl & 1) + 2 * ye + 1];
                if constexpr (is_qk) {
                    if constexpr (ph == 0)
                        asm volatile(F64_MFMA_S F64_EXP_A : [d] "+v"(sn[bk][hf]), [ps] "+v"(ps[xb]), [t0] "=&v"(t0), [t1] "=&v"(t1)
                                     : [a] "v"(fa[cur][hf]), [b] "a"(qf[bk][ks]), [x0] "v"(xs0), [x1] "v"(xs1), [sc] "s"(scale_log2e), [m] "v"(m_run[xb]));
                    else if constexpr (ph == 1)
                        asm volatile(F64_MFMA_S F64_EXP_B : [d] "+v"(sn[bk][hf]), [ps] "+v"(ps[xb]), [wx] "=&v"(pw[xb][xsl][xe]), [u0] "=&v"(u0), [u1] "=&v"(u1)
                                     : [a] "v"(fa[cur][hf]), [b] "a"(qf[bk][ks]), [t0] "v"(t0), [t1] "v"(t1), [y0] "v"(ys0), [y1] "v"(ys1), [sc] "s"(scale_log2e), [m] "v"(m_run[yb]));
                    else
                        asm volatile(F64_MFMA_S F64_EXP_C : [d] "+v"(sn[bk][hf]), [ps] "+v"(ps[yb]), [wy] "=&v"(pw[yb][ysl][ye]), [u1] "+v"(u1)
                                     : [a] "v"(fa[cur][hf]), [b] "a"(qf[bk][ks]), [u0] "v"(u0) F64_C_EXTRA);
                } else {
                    const bf16x8 pf = __builtin_bit_cast(bf16x8, u32x4{pw[bk][sl][0], pw[bk][sl][1], pw[bk][sl][2], pw[bk][sl][3]});
                    if constexpr (ph == 0)
                        asm volatile(F64_MFMA_S F64_EXP_A : [d] "+a"(o[bk][dt]), [ps] "+v"(ps[xb]), [t0] "=&v"(t0), [t1] "=&v"(t1)
                                     : [a] "v"(fa[cur][hf]), [b] "v"(pf), [x0] "v"(xs0), [x1] "v"(xs1), [sc] "s"(scale_log2e), [m] "v"(m_run[xb]));
                    else if constexpr (ph == 1)
                        asm volatile(F64_MFMA_S F64_EXP_B : [d] "+a"(o[bk][dt]), [ps] "+v"(ps[xb]), [wx] "=&v"(pw[xb][xsl][xe]), [u0] "=&v"(u0), [u1] "=&v"(u1)
                                     : [a] "v"(fa[cur][hf]), [b] "v"(pf), [t0] "v"(t0), [t1] "v"(t1), [y0] "v"(ys0), [y1] "v"(ys1), [sc] "s"(scale_log2e), [m] "v"(m_run[yb]));
                    else
                        asm volatile(F64_MFMA_S F64_EXP_C : [d] "+a"(o[bk][dt]), [ps] "+v"(ps[yb]), [wy] "=&v"(pw[yb][ysl][ye]), [u1] "+v"(u1)
                                     : [a] "v"(fa[cur][hf]), [b] "v"(pf), [u0] "v"(u0) F64_C_EXTRA);
                }
            }
        });
        F64_STAMP(3);                                             // gaps 8-63
        l_run[0] += ps[0];
        l_run[1] += ps[1];
    };
    // tiles in pairs (the two S sets swap roles); the extra tile of an odd count is fully masked (P = 0)
    for (int j = 0; j < n_tiles; j += 2) {
        tile_body(j, sA, sB);
        tile_body(j + 1, sB, sA);
    }

#ifdef F64_PROF
    if (blockIdx.x == 1000 && tid == 0)
        printf("F64_PROF tiles %d: gaps0-7 %llu, gaps8-63 %llu cycles per tile (s_memtime units)\n", n_tiles, prof[1] / n_tiles, prof[3] / n_tiles);
#endif
    // ---- epilogue: O = O^T / l ; lane owns row q, d = dt*32 + 8*rq + 4*hi + 0..3
#pragma unroll
    for (int bk = 0; bk < 2; ++bk) {
        const float l_tot = l_run[bk] + __shfl_xor(l_run[bk], 32, 64);
        const float inv = 1.0f / l_tot;
        const int qrow = qb.q0 + w * 64 + bk * 32 + l31;
        if (qrow < n) {
            bf16_t* op = O + ((size_t)(qb.tok0 + qrow) * Hq + h) * 128;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    u32x2 pk = {pack_bf2(o[bk][dt][4 * rq] * inv, o[bk][dt][4 * rq + 1] * inv),
                                pack_bf2(o[bk][dt][4 * rq + 2] * inv, o[bk][dt][4 * rq + 3] * inv)};
                    *reinterpret_cast<u32x2*>(op + dt * 32 + 8 * rq + 4 * hi) = pk;
                }
        }
    }
}
#undef F64_QK_OPS
#undef F64_PV_OPS

}  // namespace

XcdPlan make_xcd_plan(const QBlock* b, int n_blocks) {
    XcdPlan p;
    bool uniform = true;
    for (int i = 1; i < n_blocks && uniform; ++i) uniform = b[i].n == b[0].n;
    if (uniform || n_blocks < 8) {                               // equal costs: exactly xcd_remap's chunks
        const int q = n_blocks / 8, r = n_blocks % 8;
        for (int x = 0, pos = 0; x < 8; ++x) { p.base[x] = pos; p.cnt[x] = q + (x < r ? 1 : 0); pos += p.cnt[x]; }
        return p;
    }
    auto cost = [&](int i) { return (int64_t)(((b[i].n + 63) / 64 + 1) & ~1); };      // KV tiles of the block's sequence (walked in pairs)
    int64_t total = 0;
    for (int i = 0; i < n_blocks; ++i) total += cost(i);
    int pos = 0;
    int64_t done = 0;
    for (int x = 0; x < 8; ++x) {
        const int64_t target = total * (x + 1) / 8;              // prefix-sum split: chunk x ends where the running cost passes (x + 1) / 8
        p.base[x] = pos;
        while (pos < n_blocks && (x == 7 || done + cost(pos) / 2 < target)) done += cost(pos++);
        p.cnt[x] = pos - p.base[x];
    }
    return p;
}

hipError_t launch_flash_attn(hipStream_t s, const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* out,
                             const QBlock* blocks, int n_blocks, int64_t T, int64_t Tpad, int Hq, int Hkv,
                             int causal, float scale, const XcdPlan* plan) {
    if (n_blocks <= 0) return hipSuccess;
    if (Hq % Hkv != 0) return hipErrorInvalidValue;
    const float c = scale * 1.44269504088896340736f;
    dim3 grid(n_blocks);                      // n_blocks = work items (seq x head x query block)
    // DOTS_OCR_ATTN_MODE: 2 (default) = bidirectional: 4 waves x 64 rows, LDS-DMA tiles, softmax interleaved with the MFMAs (round 4);
    // causal: as 1.  1 = 8 waves, ping-pong halves + LDS fragment look-ahead (rounds 1-3), 0 = the round-1 schedule
    static const int mode = getenv("DOTS_OCR_ATTN_MODE") ? atoi(getenv("DOTS_OCR_ATTN_MODE")) : 2;
    auto go = [&](auto kern, int threads, int lds) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, grid, dim3(threads), lds, s, q, k, vt, out, blocks, n_blocks, T, Tpad, Hq, Hq / Hkv, c);
        return hipGetLastError();
    };
    if (flash_rows_per_block() == 256) {
        // causal (LM prefill) on the 64-row kernel since round 6; DOTS_OCR_PREFILL_F64=0: the 8-wave kernel of rounds 1-3 (A/B switch)
        static const bool causal64 = !(getenv("DOTS_OCR_PREFILL_F64") && atoi(getenv("DOTS_OCR_PREFILL_F64")) == 0);
        if (mode == 2 && (!causal || causal64)) {
            // dynamic LDS > 64 KB is opted into once per DEVICE (engines on several GPUs may live in one process): bit d of the mask
            static uint32_t attr_done[2] = {0, 0};
            int dev = 0;
            hipError_t e = hipGetDevice(&dev);
            if (e != hipSuccess) return e;
            const uint32_t bit = 1u << (dev & 31);
            if (!(__atomic_load_n(&attr_done[causal ? 1 : 0], __ATOMIC_ACQUIRE) & bit)) {
                e = causal ? hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn64_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, RING_BYTES)
                           : hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn64_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, RING_BYTES);
                if (e != hipSuccess) return e;
                __atomic_fetch_or(&attr_done[causal ? 1 : 0], bit, __ATOMIC_RELEASE);
            }
            XcdPlan pl;
            if (plan) pl = *plan;
            else {                                               // no plan from the caller: xcd_remap's equal-count chunks
                const int qn = n_blocks / 8, rn = n_blocks % 8;
                for (int x = 0, pos = 0; x < 8; ++x) { pl.base[x] = pos; pl.cnt[x] = qn + (x < rn ? 1 : 0); pos += pl.cnt[x]; }
            }
            int mx = 0;
            for (int x = 0; x < 8; ++x) mx = std::max(mx, (int)pl.cnt[x]);
            if (causal) hipLaunchKernelGGL(flash_attn64_kernel<true>, dim3(8 * mx), dim3(256), RING_BYTES, s, q, k, vt, out, blocks, n_blocks, T, Tpad, Hq, Hq / Hkv, c, pl);
            else hipLaunchKernelGGL(flash_attn64_kernel<false>, dim3(8 * mx), dim3(256), RING_BYTES, s, q, k, vt, out, blocks, n_blocks, T, Tpad, Hq, Hq / Hkv, c, pl);
            return hipGetLastError();
        }
        if (mode >= 1) return causal ? go(flash_attn_kernel<true, 8, 1>, 512, 3 * BUF_BYTES) : go(flash_attn_kernel<false, 8, 1>, 512, 3 * BUF_BYTES);
        return causal ? go(flash_attn_kernel<true, 8, 0>, 512, 2 * BUF_BYTES) : go(flash_attn_kernel<false, 8, 0>, 512, 2 * BUF_BYTES);
    }
    return causal ? go(flash_attn_kernel<true, 4, 0>, 256, 2 * BUF_BYTES) : go(flash_attn_kernel<false, 4, 0>, 256, 2 * BUF_BYTES);
}

// query rows per work item: 256 (8 waves sharing each K/V tile, 1 workgroup per CU; default: half the global->LDS staging
// work per wave, measured +2-3 % in interleaved A/B) or 128 (4 waves, 2 independent workgroups per CU; DOTS_OCR_ATTN_ROWS128=1)
int flash_rows_per_block() {
    static const int rows = getenv("DOTS_OCR_ATTN_ROWS128") ? 128 : 256;
    return rows;
}
