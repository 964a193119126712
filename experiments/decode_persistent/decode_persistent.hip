// One decode step as ONE persistent kernel (DESIGN.md "decode loop").
//
// The launch-per-phase step (decode_fused.hip + decode.hip: 6 kernels per layer) is bound by dependent-launch latency:
// every kernel pays ~4 us of dispatch, cold instruction cache and pipeline ramp before its first weight byte arrives,
// 170 times per step, against 0.55 ms of pure HBM streaming.  Here one workgroup per CU (8 waves, so each wave keeps
// the 256-register budget the attention and streaming loops want) stays resident for the whole step and walks the same phases
//
//     QKV -> ATTN -> COMBINE -> O_PROJ -> GATE_UP -> DOWN      (x layers)  -> LM_HEAD
//
// separated by device-wide barriers (~2 us: hierarchical arrival counters, one per XCD plus a top level, bounded spin;
// csrc/probe_sync.hip holds the measurements this design rests on).  Each phase is the arithmetic of the corresponding
// stand-alone kernel — same tiles, same K-slices, same reduction order — so the two paths produce bit-identical
// logits (tests/test_model_gpu.py compares them), and the stand-alone kernels remain the reference implementation.
//
// Memory consistency inside the kernel (each XCD has its own L2, and they are not coherent with each other):
//   * every buffer written in one phase and read in a later one exists once PER LAYER (written once, read afterwards),
//     and every 128-B line of it is written by exactly ONE workgroup (producer-owned layouts: q [head][t][row][32],
//     h1 and slabs [column tile][row][16], att / act in MFMA fragment order with a workgroup owning whole row-halves):
//     a reader either misses its L2 and gets the written-through data from memory, or shares the writer's L2.  A line
//     assembled from partial write-throughs of two XCDs would leave each writer's L2 with a stale half (seen as a
//     one-in-a-few-steps logit difference in the first version).  Plain 16-B loads are therefore safe;
//   * all such writes are agent-scope write-through stores (st_wt_*), complete (s_waitcnt vmcnt(0)) before the
//     workgroup arrives at the barrier;
//   * split-KV partials, the only buffer reused across layers, are read with agent-scope loads;
//   * weights, norm weights, biases and the step's inputs (tokens, context lengths, block table) are read-only here.
// A barrier that does not complete within its spin budget sets *err and every later barrier falls through: the kernel
// always terminates, the host reports the failure.
#include "common.h"
#include "decode_layout.h"
#include "decode_device.h"
#include "kernels.h"

namespace {

// The phases are inlined into one body.  Left alone, the optimiser hoists every phase's loop-invariant values (lane offsets,
// operand pointers, DecStep fields) above the layer loop, where they all stay live across all phases: 250 spilled VGPRs.
// Each phase therefore launders the thread index and the DecStep pointer through an empty asm first, which pins its
// address arithmetic inside the phase (a real call per phase would instead save ~120 callee-saved VGPRs to scratch).
#define PHASE __device__ __forceinline__
#define PHASE_BEGIN(pp_in)                                       \
    const DecStep* pp_ = (pp_in);                                \
    asm volatile("" : "+s"(pp_));                                \
    const DecStep& p = *pp_;                                     \
    int tid_ = threadIdx.x;                                      \
    asm volatile("" : "+v"(tid_));

constexpr int BAR_STRIDE = 8 * 32;          // ints per barrier: 8 arrival counters (one per XCD: workgroup b runs on XCD b % 8), 128 B apart
constexpr int SPIN_LIMIT = 1 << 18;

struct GridBar {
    int* base;
    int* err;
    int* dead;      // LDS flag
    int idx;
};

// Arrive on this XCD's counter, then the first 8 lanes poll all 8 counters: the critical path is one atomic round trip plus
// one poll round trip (a top-level counter would add a third), and at most 32 arrivals serialise on one address.
DEVI void grid_sync(GridBar& gb) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's write-through stores have landed
    __syncthreads();
    if (threadIdx.x < 64 && !*gb.dead) {
        int* c = gb.base + (size_t)gb.idx * BAR_STRIDE;
        const int n_wg = gridDim.x, lane = threadIdx.x;
        if (lane == 0) __hip_atomic_fetch_add(c + 32 * (blockIdx.x & 7), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int want = lane < 8 ? (n_wg - lane + 7) >> 3 : 0;      // workgroups arriving on counter `lane`
        int spins = 0;
        while (true) {
            const int got = lane < 8 ? __hip_atomic_load(c + 32 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            if (__all(got >= want)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) {
                if (lane == 0) {
                    *gb.dead = 1;
                    __hip_atomic_store(gb.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                break;
            }
        }
    }
    __syncthreads();
    ++gb.idx;
}

// ------------------------------------------------------------------------------------------------ QKV
// dec_qkv_kernel's workgroup `vb` = (head, t): two weight tiles x 8 K-slices on 16 waves, RoPE + KV append epilogue.
PHASE void phase_qkv(const DecStep* pp_in, int L, char* smem) {
    PHASE_BEGIN(pp_in)
    const DecLayer& W = p.layers[L];
    const int H = p.H, Hq = p.Hq, Hkv = p.Hkv, B = p.B, XR = p.XR;
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
    f32x4* red = reinterpret_cast<f32x4*>(smem + (size_t)XR * H * 2);
    const int lane = tid_ & 63, wv = tid_ >> 6;
    const int n_vb = (Hq + 2 * Hkv) * 4;
    const size_t hs = (size_t)16 * H;
    // residual input: layer 0 gathers embedding rows, later layers fold the previous layer's down-projection slabs
    const bf16_t* h_in = L == 0 ? p.embed : p.h1 + hs * (L - 1);
    const int32_t* rows = L == 0 ? p.cur_tokens : nullptr;
    const float* slabs = L == 0 ? nullptr : p.slabs + (size_t)4 * hs * (L - 1);
    const int n_slabs = L == 0 ? 0 : p.down_split;
    bf16_t* h_out = p.h0 + hs * L;
    bf16_t* pool = p.pool + p.pool_layer_elems * L;
    bf16_t* q_out = p.q + (size_t)16 * Hq * 128 * L;
    for (int vb = blockIdx.x; vb < n_vb; vb += gridDim.x) {
        const int head = vb >> 2, t = vb & 3;
        const int KS = H / 32;
        // unit u = wave + 8e  <->  wave u of the 16-wave stand-alone workgroup: (tile select, K-slice) = (u & 1, u >> 1)
        const bf16x8* wp[2];
        int k0[2], k1[2];
        bf16x8 a0[2][8];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int u = wv + 8 * e, sel = u & 1, slice = u >> 1;
            const int n_tile = head * 8 + t + 4 * sel;
            wp[e] = reinterpret_cast<const bf16x8*>(W.qkv_wd) + ((size_t)n_tile * KS) * 64 + lane;
            k0[e] = slice * KS / 8;
            k1[e] = (slice + 1) * KS / 8;
            preload_group(wp[e], k0[e], k1[e], a0[e]);
        }
        norm_rows_to_lds(h_in, rows, slabs, n_slabs, h_out, vb == 0, true, W.ln1, B, H, p.eps, xs, XR, wv, 8, lane, L > 0);
        __syncthreads();
        const bf16x8* xp = reinterpret_cast<const bf16x8*>(xs) + (lane >> 4) * XR + (lane & (XR - 1));
#pragma unroll
        for (int e = 0; e < 2; ++e) red[(wv + 8 * e) * 64 + lane] = stream_tile_lean(wp[e], xp, 4 * XR, k0[e], k1[e], a0[e]);
        // RoPE angles of the epilogue lanes (same values as the stand-alone kernel, which computes them under its weight loads)
        float rc[4] = {1.f, 1.f, 1.f, 1.f}, rs[4] = {0.f, 0.f, 0.f, 0.f};
        if (wv == 0 && (lane & 15) < B && head < Hq + Hkv) {
            const int pos0 = p.ctx_len[lane & 15];
#pragma unroll
            for (int r = 0; r < 4; ++r) sincosf((float)pos0 * p.inv_freq[16 * t + 4 * (lane >> 4) + r], &rs[r], &rc[r]);
        }
        __syncthreads();
        const int m = lane & 15, g = lane >> 4;
        if (wv == 0 && m < B) {
            f32x4 x1 = {0, 0, 0, 0}, x2 = {0, 0, 0, 0};
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) { x1 += red[(2 * sl) * 64 + lane]; x2 += red[(2 * sl + 1) * 64 + lane]; }
            const int d0 = 16 * t + 4 * g;
            const int pos = p.ctx_len[m];
            const int page = p.block_table[m * p.max_pages + (pos >> 6)];
            const int key = pos & 63;
            bf16_t o1[4], o2[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = d0 + r;
                float a = x1[r], b = x2[r];
                if (W.qkv_b) { a += bf2f(W.qkv_b[head * 128 + d]); b += bf2f(W.qkv_b[head * 128 + d + 64]); }
                a = bf2f(f2bf(a)); b = bf2f(f2bf(b));
                if (head < Hq + Hkv) {
                    const float sn = rs[r], cs = rc[r];
                    o1[r] = f2bf(a * cs - b * sn);
                    o2[r] = f2bf(b * cs + a * sn);
                } else {
                    o1[r] = f2bf(a);
                    o2[r] = f2bf(b);
                }
            }
            const u32x2 w1 = {(uint32_t)o1[0] | ((uint32_t)o1[1] << 16), (uint32_t)o1[2] | ((uint32_t)o1[3] << 16)};
            const u32x2 w2 = {(uint32_t)o2[0] | ((uint32_t)o2[1] << 16), (uint32_t)o2[2] | ((uint32_t)o2[3] << 16)};
            if (head < Hq) {
                bf16_t* qp = q_out + ((size_t)(head * 4 + t) * 16 + m) * 32 + 4 * g;       // [head][t][row][d & 15 | 16 + (d & 15)]
                st_wt_u64(qp, w1);
                st_wt_u64(qp + 16, w2);
            } else if (head < Hq + Hkv) {
                bf16_t* kp = pool + ((size_t)(page * Hkv + (head - Hq)) * 2) * PAGE_ELEMS;
                st_wt_u64(kp + k_chunk(key, d0) * 8 + (d0 & 7), w1);                 // 4 consecutive d of one 16-B chunk
                st_wt_u64(kp + k_chunk(key, d0 + 64) * 8 + (d0 & 7), w2);
            } else {
                bf16_t* vp = pool + ((size_t)(page * Hkv + (head - Hq - Hkv)) * 2 + 1) * PAGE_ELEMS;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st_wt_u16(vp + v_off(key, d0 + r), o1[r]);
                    st_wt_u16(vp + v_off(key, d0 + r + 64), o2[r]);
                }
            }
        }
        __syncthreads();                                   // LDS is reused by the next trip / phase
    }
}

// ------------------------------------------------------------------------------------------------ ATTN
// decode_attn_kernel's workgroup = one 4-wave group here; the 8-wave workgroup runs 2 of them side by side.
PHASE void phase_attn(const DecStep* pp_in, int L, char* smem) {
    PHASE_BEGIN(pp_in)
    const int Hq = p.Hq, Hkv = p.Hkv, B = p.B, n_splits = p.n_splits;
    const int group = Hq / Hkv;
    const int grp = tid_ >> 8, tid = tid_ & 255;
    const int l = tid & 63, w = tid >> 6, i = l & 15, g = l >> 4;
    float* lds_o = reinterpret_cast<float*>(smem) + (size_t)grp * (4 * 16 * 128 + 128);       // [4][16*128]
    float* lds_m = lds_o + 4 * 16 * 128;                                                        // [4][16]
    float* lds_l = lds_m + 64;                                                                  // [4][16]
    const bf16_t* pool = p.pool + p.pool_layer_elems * L;
    const bf16_t* q = p.q + (size_t)16 * Hq * 128 * L;
    const int n_vb = n_splits * Hkv * B;
    for (int v0 = 0; v0 < n_vb; v0 += 2 * gridDim.x) {
        const int vb = v0 + grp * gridDim.x + blockIdx.x;
        const int split = vb % n_splits, hkv = (vb / n_splits) % Hkv, b = vb / (n_splits * Hkv);
        bool active = vb < n_vb;
        int ctx = 0, n_pages = 0;
        if (active) {
            ctx = p.ctx_len[b] + 1;
            n_pages = (ctx + PAGE - 1) / PAGE;
            active = split * 4 < n_pages;
        }
        f32x4 o[8];
#pragma unroll
        for (int dg = 0; dg < 8; ++dg) o[dg] = f32x4{0, 0, 0, 0};
        float m_run = -1e30f, l_run = 0.f;
        if (active) {
            bf16x8 qf[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                u32x4 z = {0, 0, 0, 0};
                const int d = kk * 32 + g * 8;                     // 8 consecutive d of one (t, upper/lower half) block
                if (i < group)
                    z = *reinterpret_cast<const u32x4*>(q + ((size_t)((hkv * group + i) * 4 + ((d & 63) >> 4)) * 16 + b) * 32 + (d >> 6) * 16 + (d & 15));
                qf[kk] = __builtin_bit_cast(bf16x8, z);
            }
            for (int pg = split * 4 + w; pg < n_pages; pg += 4 * n_splits) {
                const int page = p.block_table[b * p.max_pages + pg];
                const bf16_t* kp = pool + ((size_t)(page * Hkv + hkv) * 2) * PAGE_ELEMS;
                const bf16_t* vp = kp + PAGE_ELEMS;
                bf16x8 kf[16], vf[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) kf[c] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(kp + (size_t)(c * 64 + l) * 8));
#pragma unroll
                for (int c = 0; c < 16; ++c) vf[c] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(vp + (size_t)(c * 64 + l) * 8));
                f32x4 s[4];
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) {
                    s[kg] = f32x4{0, 0, 0, 0};
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) s[kg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kg * 4 + kk], qf[kk], s[kg], 0, 0, 0);
                }
                const int key0 = pg * PAGE;
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
                const float m_new = fmaxf(m_run, mx * p.scale_log2e);
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
                            pv[r] = __builtin_amdgcn_exp2f(fmaf(s[slab * 2 + t][r], p.scale_log2e, -m_new));
                            psum += pv[r];
                        }
                        pk[t * 2] = pack_bf2(pv[0], pv[1]);
                        pk[t * 2 + 1] = pack_bf2(pv[2], pv[3]);
                    }
                    pf[slab] = __builtin_bit_cast(bf16x8, pk);
                }
                l_run = l_run * alpha + psum;
#pragma unroll
                for (int dg = 0; dg < 8; ++dg) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[dg][r] *= alpha;
#pragma unroll
                    for (int slab = 0; slab < 2; ++slab)
                        o[dg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[slab * 8 + dg], pf[slab], o[dg], 0, 0, 0);
                }
            }
            l_run += __shfl_xor(l_run, 16, 64);
            l_run += __shfl_xor(l_run, 32, 64);
            if (g == 0) { lds_m[w * 16 + i] = m_run; lds_l[w * 16 + i] = l_run; }
#pragma unroll
            for (int dg = 0; dg < 8; ++dg)
#pragma unroll
                for (int r = 0; r < 4; ++r) lds_o[w * 2048 + i * 128 + dg * 16 + 4 * g + r] = o[dg][r];
        }
        __syncthreads();
        if (active) {
            for (int item = tid; item < group * 128; item += 256) {
                const int j = item >> 7, d = item & 127;
                const float m = fmaxf(fmaxf(lds_m[j], lds_m[16 + j]), fmaxf(lds_m[32 + j], lds_m[48 + j]));
                float acc = 0.f, lsum = 0.f;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    const float f = __builtin_amdgcn_exp2f(lds_m[ww * 16 + j] - m);
                    acc += lds_o[ww * 2048 + j * 128 + d] * f;
                    lsum += lds_l[ww * 16 + j] * f;
                }
                const size_t base = (((size_t)b * Hkv + hkv) * n_splits + split) * group + j;
                st_wt_f32(p.part_o + base * 128 + d, acc);
                if (d == 0) { st_wt_f32(p.part_ml + base * 2, m); st_wt_f32(p.part_ml + base * 2 + 1, lsum); }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ COMBINE
// decode_attn_combine_kernel per (row, head, d); a workgroup owns one (head, half of the 16 rows): in the fragment-order
// output a 128-B line holds 8 rows x 8 features, so all writers of a line sit in one workgroup.  Quarter `sub` of the
// workgroup takes rows 2*sub and 2*sub + 1 of the half, its two waves compute the split weights of one row each.
PHASE void phase_combine(const DecStep* pp_in, int L, char* smem) {
    PHASE_BEGIN(pp_in)
    const int Hq = p.Hq, Hkv = p.Hkv, B = p.B, n_splits = p.n_splits;
    const int group = Hq / Hkv;
    const int sub = tid_ >> 7, d = tid_ & 127, wsel = (tid_ >> 6) & 1, lane = tid_ & 63;
    float* wts_all = reinterpret_cast<float*>(smem);              // [4 subs][2 rows][64 weights + 1/l]
    bf16_t* out = p.att + (size_t)16 * Hq * 128 * L;
    const int n_items = Hq * ((B + 7) / 8);
    for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
        const int head = it % Hq, mh = it / Hq;
        const int hkv = head / group, j = head % group;
        {
            const int b = 8 * mh + 2 * sub + wsel;
            if (b < B) {
                const int n_used = min(n_splits, ((p.ctx_len[b] + 1 + PAGE - 1) / PAGE + 3) / 4);
                const size_t base0 = (((size_t)b * Hkv + hkv) * n_splits) * group + j;
                float m = -1e30f, l = 0.f;
                if (lane < n_used) { m = ld_agent_f32(p.part_ml + (base0 + (size_t)lane * group) * 2); l = ld_agent_f32(p.part_ml + (base0 + (size_t)lane * group) * 2 + 1); }
                const float mg = wave_max(m);
                const float f = lane < n_used ? __builtin_amdgcn_exp2f(m - mg) : 0.f;
                float* wts = wts_all + (sub * 2 + wsel) * 68;
                wts[lane] = f;
                const float lsum = wave_sum(l * f);
                if (lane == 0) wts[64] = 1.0f / lsum;
            }
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int b = 8 * mh + 2 * sub + rr;
            if (b < B) {
                const int n_used = min(n_splits, ((p.ctx_len[b] + 1 + PAGE - 1) / PAGE + 3) / 4);
                const size_t base0 = (((size_t)b * Hkv + hkv) * n_splits) * group + j;
                const float* wts = wts_all + (sub * 2 + rr) * 68;
                const float* po = p.part_o + base0 * 128 + d;
                const size_t stride = (size_t)group * 128;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                int s = 0;
                for (; s + 4 <= n_used; s += 4) {
                    a0 += ld_agent_f32(po + (size_t)s * stride) * wts[s];
                    a1 += ld_agent_f32(po + (size_t)(s + 1) * stride) * wts[s + 1];
                    a2 += ld_agent_f32(po + (size_t)(s + 2) * stride) * wts[s + 2];
                    a3 += ld_agent_f32(po + (size_t)(s + 3) * stride) * wts[s + 3];
                }
                for (; s < n_used; ++s) a0 += ld_agent_f32(po + (size_t)s * stride) * wts[s];
                st_wt_u16(out + frag_off(b, head * 128 + d), f2bf(((a0 + a1) + (a2 + a3)) * wts[64]));
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ O_PROJ / DOWN
// dec_proj_kernel's workgroup (n_tile, part): 16 K-slices (two per wave), reduced through LDS in slice order.
//   resid != nullptr : out_h[m][n] = bf16(resid[m][n] + sum)          (o_proj, parts == 1)
//   resid == nullptr : out_slabs[part][m][n] = sum (fp32)             (down_proj; the next consumer folds the slabs)
PHASE void phase_proj(const DecStep* pp_in, int L, bool down, char* smem) {
    PHASE_BEGIN(pp_in)
    const size_t hs = (size_t)16 * p.H;
    const bf16_t* Xf = down ? p.act + (size_t)16 * p.I * L : p.att + (size_t)16 * p.Hq * 128 * L;
    const bf16_t* Wd = down ? p.layers[L].down_wd : p.layers[L].o_wd;
    const bf16_t* resid = down ? nullptr : p.h0 + hs * L;
    bf16_t* out_h = p.h1 + hs * L;
    float* out_slabs = p.slabs + (size_t)4 * hs * L;
    const int parts = down ? p.down_split : 1, N = p.H, K = down ? p.I : p.Hq * 128;
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    const int lane = tid_ & 63, wv = tid_ >> 6;
    const int n_tiles = N / 16, KS = K / 32;
    for (int vb = blockIdx.x; vb < n_tiles * parts; vb += gridDim.x) {
        const int n_tile = vb % n_tiles, part = vb / n_tiles;
        const int slices = parts * 16;
        const bf16x8* wp = reinterpret_cast<const bf16x8*>(Wd) + ((size_t)n_tile * KS) * 64 + lane;
        // K-slice u = wave + 8e  <->  wave u of the 16-wave stand-alone workgroup; both first groups are in flight at once
        int k0[2], k1[2];
        bf16x8 a0[2][8];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int slice = part * 16 + wv + 8 * e;
            k0[e] = (int)((int64_t)slice * KS / slices);
            k1[e] = (int)((int64_t)(slice + 1) * KS / slices);
            preload_group(wp, k0[e], k1[e], a0[e]);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e)
            red[(wv + 8 * e) * 64 + lane] = stream_tile(wp, reinterpret_cast<const bf16x8*>(Xf) + lane, 64, k0[e], k1[e], a0[e]);
        __syncthreads();
        const int m = lane & 15, g = lane >> 4;
        if (wv == 0 && m < p.B) {
            f32x4 a = {0, 0, 0, 0};
#pragma unroll
            for (int sl = 0; sl < 16; ++sl) a += red[sl * 64 + lane];
            if (!resid) {
                st_wt_f32x4(out_slabs + (((size_t)part * n_tiles + n_tile) * 16 + m) * 16 + 4 * g, a);        // [part][tile][row][16]
            } else {
                const u32x2 x = *reinterpret_cast<const u32x2*>(resid + (size_t)m * N + n_tile * 16 + 4 * g);
                st_wt_u64(out_h + ((size_t)n_tile * 16 + m) * 16 + 4 * g,                                     // [tile][row][16]
                          u32x2{pack_bf2(lo_bf(x[0]) + a[0], hi_bf(x[0]) + a[1]), pack_bf2(lo_bf(x[1]) + a[2], hi_bf(x[1]) + a[3])});
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ GATE_UP
// dec_gateup_kernel's 4-wave workgroup (one gate/up tile pair, K over 4 waves) = one 4-wave group here; the normalised
// rows are built once per workgroup and shared by its two groups.  Accumulation order per (pair, wave) is the stand-alone
// kernel's.
constexpr int PGU = 12;          // k-steps of a wave's K-slice held in registers (H = 1536: 48 / 4)

DEVI void gu_load(const bf16x8* wg, const bf16x8* wu, int k0, int k1, bf16x8 (&a)[PGU], bf16x8 (&u)[PGU]) {
#pragma unroll
    for (int j = 0; j < PGU; ++j)
        if (k0 + j < k1) {
            a[j] = __builtin_nontemporal_load(wg + (size_t)(k0 + j) * 64);
            u[j] = __builtin_nontemporal_load(wu + (size_t)(k0 + j) * 64);
        }
}

PHASE void phase_gateup(const DecStep* pp_in, int L, char* smem) {
    PHASE_BEGIN(pp_in)
    const DecLayer& W = p.layers[L];
    const int H = p.H, I = p.I, B = p.B, XR = p.XR;
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
    f32x4* red = reinterpret_cast<f32x4*>(smem + (size_t)XR * H * 2);                 // [2 groups][4 waves][gate|up][64]
    const int lane = tid_ & 63, wv = tid_ >> 6;
    const int grp = wv >> 2, gw = wv & 3;
    const int KS = H / 32;
    const int k0 = gw * KS / 4, k1 = (gw + 1) * KS / 4;
    const bf16_t* h = p.h1 + (size_t)16 * H * L;
    bf16_t* act = p.act + (size_t)16 * I * L;
    const bf16x8* xp = reinterpret_cast<const bf16x8*>(xs) + (lane >> 4) * XR + (lane & (XR - 1));
    const int xstride = 4 * XR;
    const int n_vb = I / 16;
    const bf16x8* w13 = reinterpret_cast<const bf16x8*>(W.w13_wd) + lane;
    bool have_x = false;
    // I/16 pairs over (2 groups x n_wg) slots: 560 / 512 at I = 8960.  The second, partial trip only adds bytes to a phase
    // that is bandwidth-bound as a whole (its loads are issued while the other workgroups are still streaming).
    for (int v0 = 0; v0 < n_vb; v0 += 2 * gridDim.x) {
        const int pair = v0 + grp * gridDim.x + blockIdx.x;
        const bool on = pair < n_vb;
        const int G = pair >> 1, a = pair & 1;
        const bf16x8* wg = w13 + ((size_t)(G * 4 + a) * KS) * 64;
        const bf16x8* wu = w13 + ((size_t)(G * 4 + 2 + a) * KS) * 64;
        bf16x8 a_[PGU], u_[PGU];
        if (on) gu_load(wg, wu, k0, k1, a_, u_);
        if (!have_x) {                                     // the weights above are in flight under the prologue
            norm_rows_to_lds(h, nullptr, nullptr, 0, nullptr, false, false, W.ln2, B, H, p.eps, xs, XR, wv, 8, lane, true);
            have_x = true;
            __syncthreads();
        }
        f32x4 ag = {0, 0, 0, 0}, au = {0, 0, 0, 0};
        if (on) {
            for (int ks = k0; ks < k1; ks += PGU) {          // one trip when the slice fits (H <= 1536)
#pragma unroll
                for (int j = 0; j < PGU; ++j)
                    if (ks + j < k1) {
                        const bf16x8 b = xp[(size_t)(ks + j) * xstride];
                        ag = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_[j], b, ag, 0, 0, 0);
                        au = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u_[j], b, au, 0, 0, 0);
                    }
                if (ks + PGU < k1) gu_load(wg, wu, ks + PGU, k1, a_, u_);
            }
            red[((grp * 4 + gw) * 2) * 64 + lane] = ag;
            red[((grp * 4 + gw) * 2 + 1) * 64 + lane] = au;
        }
        __syncthreads();
        const int m = lane & 15, g = lane >> 4;
        if (on && gw == 0 && m < B) {
            f32x4 gs = ag, us = au;
#pragma unroll
            for (int ww = 1; ww < 4; ++ww) { gs += red[((grp * 4 + ww) * 2) * 64 + lane]; us += red[((grp * 4 + ww) * 2 + 1) * 64 + lane]; }
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = gs[r] / (1.0f + __expf(-gs[r])) * us[r];
            st_wt_u64(act + frag_off(m, G * 32 + a * 16 + 4 * g), u32x2{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])});
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ LM_HEAD
// dec_lmhead_kernel's wave = one 16-row vocabulary tile over the full K.
PHASE void phase_lmhead(const DecStep* pp_in, char* smem) {
    PHASE_BEGIN(pp_in)
    const int H = p.H, V = p.V, B = p.B, XR = p.XR, L = p.n_layers - 1;
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
    const int lane = tid_ & 63, wv = tid_ >> 6;
    const int KS = H / 32, n_tiles = V / 16;
    const size_t hs = (size_t)16 * H;
    int n_tile = blockIdx.x * 8 + wv;
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(p.lm_head_d) + ((size_t)min(n_tile, n_tiles - 1) * KS) * 64 + lane;
    bf16x8 a0[8];
    preload_group(wp, 0, KS, a0);
    norm_rows_to_lds(p.h1 + hs * L, nullptr, p.slabs + (size_t)4 * hs * L, p.down_split, nullptr, false, false, p.final_norm, B, H, p.eps,
                     xs, XR, wv, 8, lane, true);
    __syncthreads();
    const bf16x8* xp = reinterpret_cast<const bf16x8*>(xs) + (lane >> 4) * XR + (lane & (XR - 1));
    const int m = lane & 15, g = lane >> 4;
    for (; n_tile < n_tiles; n_tile += gridDim.x * 8) {
        const f32x4 acc = stream_tile(wp, xp, 4 * XR, 0, KS, a0);
        if (m < B) *reinterpret_cast<f32x4*>(p.logits + (size_t)m * V + n_tile * 16 + 4 * g) = acc;
        const int next = n_tile + gridDim.x * 8;
        if (next < n_tiles) {
            wp = reinterpret_cast<const bf16x8*>(p.lm_head_d) + ((size_t)next * KS) * 64 + lane;
            preload_group(wp, 0, KS, a0);
        }
    }
}

__global__ __launch_bounds__(512) void dec_step_kernel(const DecStep* __restrict__ pp) {
    const DecStep& p = *pp;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_dead;
    if (threadIdx.x == 0) s_dead = 0;
    __syncthreads();
    GridBar gb{p.bar, p.err, &s_dead, 0};
    const int n_layers = p.n_layers;
    for (int L = 0; L < n_layers; ++L) {
        phase_qkv(pp, L, smem);
        grid_sync(gb);
        phase_attn(pp, L, smem);
        grid_sync(gb);
        phase_combine(pp, L, smem);
        grid_sync(gb);
        phase_proj(pp, L, false, smem);
        grid_sync(gb);
        phase_gateup(pp, L, smem);
        grid_sync(gb);
        phase_proj(pp, L, true, smem);
        grid_sync(gb);
    }
    phase_lmhead(pp, smem);
}

}  // namespace

size_t dec_step_lds_bytes(int H, int XR) {
    const size_t attn = (size_t)2 * (4 * 16 * 128 + 128) * sizeof(float);
    const size_t dense = (size_t)XR * H * 2 + 32 * 1024;
    const size_t need = attn > dense ? attn : dense;
    return need > 82 * 1024 ? need : 82 * 1024;             // > half of the 160 KB LDS: exactly one workgroup per CU
}

int dec_step_barriers(int n_layers) { return 6 * n_layers; }
size_t dec_step_barrier_bytes(int n_layers) { return (size_t)dec_step_barriers(n_layers) * BAR_STRIDE * sizeof(int); }

hipError_t launch_dec_step(hipStream_t s, const DecStep& p, const DecStep* p_dev, int n_wg) {
    if (p.H % 256 || p.H > 2048 || p.I % 32 || p.V % 16 || p.B < 1 || p.B > 16 || p.down_split < 1 || p.down_split > 4 ||
        (p.Hq * 128) / 32 < 16 || p.I / 32 < 16 * p.down_split || n_wg < 1 || p.n_splits < 1 || p.n_splits > 64)
        return hipErrorInvalidValue;
    static bool attr = false;
    const size_t lds = dec_step_lds_bytes(p.H, p.XR);
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)dec_step_lds_bytes(2048, 16));
        if (e != hipSuccess) return e;
        attr = true;
    }
    HIP_CHECK_RET(hipMemsetAsync(p.bar, 0, dec_step_barrier_bytes(p.n_layers), s));
    hipLaunchKernelGGL(dec_step_kernel, dim3(n_wg), dim3(512), lds, s, p_dev);
    return hipGetLastError();
}
