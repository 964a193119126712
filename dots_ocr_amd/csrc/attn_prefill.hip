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
#include <cstdlib>

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

}  // namespace

hipError_t launch_flash_attn(hipStream_t s, const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* out,
                             const QBlock* blocks, int n_blocks, int64_t T, int64_t Tpad, int Hq, int Hkv,
                             int causal, float scale) {
    if (n_blocks <= 0) return hipSuccess;
    if (Hq % Hkv != 0) return hipErrorInvalidValue;
    const float c = scale * 1.44269504088896340736f;
    dim3 grid(n_blocks);                      // n_blocks = work items (seq x head x query block)
    // DOTS_OCR_ATTN_MODE: 1 (default) = ping-pong halves + LDS fragment look-ahead, 0 = the round-1 schedule
    static const int mode = getenv("DOTS_OCR_ATTN_MODE") ? atoi(getenv("DOTS_OCR_ATTN_MODE")) : 1;
    auto go = [&](auto kern, int threads, int lds) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, grid, dim3(threads), lds, s, q, k, vt, out, blocks, n_blocks, T, Tpad, Hq, Hq / Hkv, c);
        return hipGetLastError();
    };
    if (flash_rows_per_block() == 256) {
        if (mode == 1) return causal ? go(flash_attn_kernel<true, 8, 1>, 512, 3 * BUF_BYTES) : go(flash_attn_kernel<false, 8, 1>, 512, 3 * BUF_BYTES);
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
