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
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int KT_BYTES = 64 * 256;    // K tile
constexpr int VT_BYTES = 128 * 128;   // V^T tile
constexpr int BUF_BYTES = KT_BYTES + VT_BYTES;
constexpr float RESCALE_THR = 6.0f;   // log2 units: P <= 64; THR = 0 reproduces the textbook rescale-every-tile

template <bool CAUSAL, int NW>
__global__ __launch_bounds__(NW * 64, 2) void flash_attn_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ VT,
    bf16_t* __restrict__ O, const QBlock* __restrict__ blocks, int n_items, int64_t T, int64_t Tpad, int Hq, int group,
    float scale_log2e) {
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES];

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

    auto pv = [&](int buf) {                         // O^T += V^T(buf) . pf : 4 d tiles x 4 key slabs
        const char* vb = smem + buf * BUF_BYTES + KT_BYTES;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                bf16x8 vf = *reinterpret_cast<const bf16x8*>(vb + dt * 32 * 128 + v_row_off + (((sl * 2 + hi) ^ v_sw) << 4));
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[sl], o[dt], 0, 0, 0);
            }
    };

    load_k(0);
    write_k(0);
    __syncthreads();

    // Software-pipelined by one stage: iteration j issues ONE block of 32 MFMAs — S^T(j) = K(j).Q^T and
    // O^T += V^T(j-1).P^T(j-1) — then runs the softmax of tile j on the VALU.  The two waves of a SIMD fall into
    // antiphase (one in its MFMA block while the other is in its VALU block), which a strict QK -> softmax -> PV
    // order per tile cannot do.  V(j) is staged during iteration j (it is first read in iteration j+1), K(j+1) too,
    // so two LDS buffers and one barrier per tile still suffice:
    //   K buffer (j+1)&1 was last read by QK(j-1), V buffer j&1 by PV(j-2) — both before the previous barrier.
    for (int j = 0; j < n_tiles; ++j) {
        const bool has_next = (j + 1 < n_tiles);
        if (has_next) load_k(j + 1);
        load_v(j);
        const char* kb = smem + (j & 1) * BUF_BYTES;

        // ---- S^T = K . Q^T : 2 key tiles x 8 k-steps ----
        f32x16 s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb + t * 32 * 256 + k_row_off + (((ks * 2 + hi) ^ k_sw) << 4));
                s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[t], 0, 0, 0);
            }
        }
        if (j > 0) pv((j - 1) & 1);

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
        // exponent have the headroom).  When the branch fires, O (complete up to tile j-1: its PV was issued above)
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

        if (has_next) write_k((j + 1) & 1);
        write_v(j & 1);
        __syncthreads();
    }
    pv((n_tiles - 1) & 1);

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
    if (flash_rows_per_block() == 256) {
        if (causal)
            hipLaunchKernelGGL((flash_attn_kernel<true, 8>), grid, dim3(512), 0, s, q, k, vt, out, blocks, n_blocks, T, Tpad, Hq, Hq / Hkv, c);
        else
            hipLaunchKernelGGL((flash_attn_kernel<false, 8>), grid, dim3(512), 0, s, q, k, vt, out, blocks, n_blocks, T, Tpad, Hq, Hq / Hkv, c);
    } else {
        if (causal)
            hipLaunchKernelGGL((flash_attn_kernel<true, 4>), grid, dim3(256), 0, s, q, k, vt, out, blocks, n_blocks, T, Tpad, Hq, Hq / Hkv, c);
        else
            hipLaunchKernelGGL((flash_attn_kernel<false, 4>), grid, dim3(256), 0, s, q, k, vt, out, blocks, n_blocks, T, Tpad, Hq, Hq / Hkv, c);
    }
    return hipGetLastError();
}

// query rows per work item: 256 (8 waves sharing each K/V tile, 1 workgroup per CU; default: half the global->LDS staging
// work per wave, measured +2-3 % in interleaved A/B) or 128 (4 waves, 2 independent workgroups per CU; DOTS_OCR_ATTN_ROWS128=1)
int flash_rows_per_block() {
    static const int rows = getenv("DOTS_OCR_ATTN_ROWS128") ? 128 : 256;
    return rows;
}
