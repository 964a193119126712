// Dataflow decode layer (round 3): the launch-per-phase decode layer
//     qkv | attention | combine | o_proj | gate|up | down           (reference call site dots_ocr/parser.py:110)
// with the two pairs whose second kernel streams far more bytes than its first fused into ONE launch each:
//     [qkv -> attention] | combine | [o_proj -> gate|up] | down
// The workgroups of the second ROLE of a launch (selected by block id) are dispatched while the first role still runs: they request
// their KV pages / weight slices at once (those bytes depend on nothing the step computes), ONE wave per workgroup polls the READY
// flag of the first role, and only then fetches the few KB of fresh activations.  A kernel boundary (1.6 us) plus the ramp of a cold
// stream behind it becomes an in-launch hand-off that overlaps the stream.  Measured first on synthetic roles with this layer's
// byte counts (tools/dataflow_probe.hip, profiles/r03_dataflow_probe.txt).
//
// Hand-off protocol (guide §6 G16 form R1, shaped by the probe):
//   producer  payload with write-through (sc1) stores -> every storing wave `s_waitcnt vmcnt(0)` -> one lane ARRIVES: a returning
//             add on one of 16 shard counters (128 B apart); the last arriver of a shard adds to the top counter; the last of
//             those stores the READY word into 16 replicas.  (All consumers polling the counter the producers add to was
//             measured 2x SLOWER than separate launches: ~12 ns per operation on one word, thousands of them.)
//   consumer  wave 0 polls ONE replica (relaxed agent load, s_sleep) — a line that is written once —, then reads the payload with
//             sc1 (L1-bypassing) loads.  Every spin is bounded; a timeout sets *err (the engine fails the step).
//   ordering  a wave's loads return IN ORDER, so the polling wave must not have a prefetch of its own in flight: wave 0 of a
//             consumer workgroup polls FIRST, fetches the activations, and requests its slice of the stream last; waves 1-3 request
//             theirs at once (after a short delay that lets the loads of the producer role enter the memory queues first:
//             in the first version the producers' 6 MB sat behind the consumers' 47-55 MB and the chain got slower, not faster).
// Deadlock freedom: a role only ever waits for a role with LOWER block ids, i.e. workgroups the dispatcher has started earlier
// (block ids are dispatched in order), and the first role of a launch never waits.  The sync words are zeroed by a memset node at
// the head of every step.
//
// Numerics: every role runs the arithmetic of its launch-per-phase twin (decode_fused.hip / decode.hip) in the same order — the
// 16-way split-K of the 1024-thread kernels is kept as 16 partial sums per output tile, four per wave, reduced in the same
// sequence — so logits and tokens are bit-identical to the round-2 path (tests/test_flow_gpu.py).
// The token appended this step never makes a round trip through its KV page: the qkv role also hands the new K / V row over in
// the q buffer and the attention wave that owns the last page patches it into its registers (the page store is for later steps).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "decode_layout.h"
#include "kernels.h"

namespace {
TRACE_DECL
#include "decode_dev.h"

#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int FLOW_NSH = 16, FLOW_NREP = 16;
constexpr int FLOW_SYNC_WORDS = (FLOW_NSH + 1 + FLOW_NREP) * 32;      // per edge: shard counters | top counter | READY replicas, 128 B apart
constexpr unsigned FLOW_SPIN_MAX = 1u << 18;       // x ~0.5 us

// ---- hand-off primitives ---------------------------------------------------------------------------------------------
DEVI void flow_arrive(uint32_t* sync, int b, int n) {          // ONE lane, after the workgroup's stores have drained
    const int s = b & (FLOW_NSH - 1);
    const uint32_t per = (uint32_t)(n / FLOW_NSH + (s < (n & (FLOW_NSH - 1)) ? 1 : 0));
    if (__hip_atomic_fetch_add(sync + s * 32, 1u, RLX) != per - 1) return;
    const uint32_t nsh = n < FLOW_NSH ? n : FLOW_NSH;
    if (__hip_atomic_fetch_add(sync + FLOW_NSH * 32, 1u, RLX) != nsh - 1) return;
#pragma unroll
    for (int r = 0; r < FLOW_NREP; ++r) __hip_atomic_store(sync + (FLOW_NSH + 1 + r) * 32, 1u, RLX);
}
// ONE wave: spin on this workgroup's replica of the READY word
DEVI void flow_poll(uint32_t* sync, uint32_t* err, int blk) {
    uint32_t* flag = sync + (FLOW_NSH + 1 + (blk & (FLOW_NREP - 1))) * 32;
    unsigned spins = 0;
    while (__hip_atomic_load(flag, RLX) == 0u) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > FLOW_SPIN_MAX) { if ((threadIdx.x & 63) == 0) __hip_atomic_store(err, 1u, RLX); break; }
    }
    asm volatile("" ::: "memory");
}
DEVI void flow_delay(int n) {                      // n x 512 cycles (~0.21 us), wave-uniform
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);
}
#define FLOW_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#ifdef DOTS_TRACE
#define FTRACE(gblk, slot) do { if ((threadIdx.x & 63) == 0 && dots_trace_buf) dots_trace_buf[((size_t)(gblk) * 4 + (threadIdx.x >> 6)) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define FTRACE(gblk, slot)
#endif

// L1-bypassing (sc1) 16-byte load / store through a buffer descriptor of a wave-uniform base
DEVI u32x4 ld16_sc1(const void* base, uint32_t byte_off) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);
}
DEVI void st4_sc1(void* p, uint32_t v) { __hip_atomic_store(reinterpret_cast<uint32_t*>(p), v, RLX); }
DEVI void st8_sc1(void* p, u32x2 v) { __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v[0] | ((unsigned long long)v[1] << 32), RLX); }
// one 1 KiB piece global -> LDS by DMA (lane i: 16 bytes at src + 16 i -> dst + 16 i), L1-bypassing
DEVI void dma1k_sc1(const void* src_lane, void* dst_wave) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_lane, (__attribute__((address_space(3))) void*)dst_wave, 16, 0, 16);
}

}  // namespace

// ---- launch arguments (by value) ---------------------------------------------------------------------------------------
struct FlowStep {
    bf16_t* h;                  // [B][H] residual stream (in place)
    bf16_t* qkvn;               // [B][(Hq + 2 Hkv) * 128]: q | k | v of the token of this step (after bias / RoPE, bf16)
    float* part_o;              // [B][Hkv][n_splits][group][128]
    float* part_ml;             // [B][Hkv][n_splits][group][2]
    bf16_t* att;                // X image [Nq/8][XR][8]
    bf16_t* act;                // X image [I/8][XR][8]
    const float* inv_freq;
    const int32_t* ctx_len;
    const int32_t* block_table;
    uint32_t* err;
    int max_pages, B, H, Hq, Hkv, I, n_splits, XR;
    float eps, scale_log2e;
    int delay_attn, delay_gu;   // flow_delay units before the prefetch of the consumer role's waves 1-3
};
struct FlowLayer {
    const bf16_t *ln1, *ln2, *qkv_b;
    const void *qkv_w, *o_w, *w13;
    const float *qkv_s, *o_s, *w13_s;
    bf16_t* pool;               // this layer's KV pages
    uint32_t* sync;             // [FLOW_EDGES][FLOW_SYNC_WORDS] of this layer
};

namespace {
enum { E_QKV = 0, E_O, FLOW_EDGES };
DEVI uint32_t* edge(uint32_t* sync, int e) { return sync + (size_t)e * FLOW_SYNC_WORDS; }

// ------------------------------------------------------------------------------------------------ role: qkv
// dec_qkv_kernel (decode_fused.hip) with 4 waves: wave w computes the partial sums of the 16-wave kernel's waves 4w .. 4w+3, and one
// whole 16-row tile per workgroup ((Hq + 2 Hkv) * 8 workgroups) instead of an 8-row half tile — half as many workgroups, so that
// [qkv + attention] are co-resident on 256 CUs x 2 workgroups; the arithmetic per output element is the same.
template <int NC, typename WT>
DEVI void role_qkv(const FlowStep& st, const FlowLayer& ly, int blk, int n_self, char* smem) {
    const int H = st.H, Hq = st.Hq, Hkv = st.Hkv, B = st.B, XR = st.XR, NQKV = (Hq + 2 * Hkv) * 128;
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                                 // [H/8][XR][8]
    f32x4* red = reinterpret_cast<f32x4*>(smem + (size_t)XR * H * 2);             // [16][64]
    const int lane = threadIdx.x & 63, wv = wave_id();
    constexpr bool FULL = true;
    const int half = 0, tile = blk, head = tile >> 3, j = tile & 7;
    const int KS = H / 32;
    const int m = lane & 15, g = lane >> 4;
    const bool rot = head < Hq + Hkv;
    const bool epi = wv == 3 && m < B && (FULL || g < 2);
    FTRACE(blk, 0);
    Rows<2, NC> R;
    rows_issue<2, NC>(R, st.h, ly.ln1, B, H, wv, 4, lane);          // h comes from the previous launch: plain loads
    const int gg = FULL ? g : (g & 1);                 // accumulator rows 4 gg .. 4 gg + 3 of the tile (FULL) / of its half
    const int f0 = rot ? 8 * j + 4 * half + 2 * gg : 16 * j + 8 * half + 4 * gg;
    const int f1 = rot ? f0 + 64 : f0 + 2;
    const int mc = min(m, B - 1);
    const int pos = st.ctx_len[mc];
    const bf16_t* bp = ly.qkv_b ? ly.qkv_b + head * 128 : ly.ln1;
    const uint32_t bia0 = *reinterpret_cast<const uint32_t*>(bp + f0), bia1 = *reinterpret_cast<const uint32_t*>(bp + f1);
    const float fr0 = st.inv_freq[f0 & 63], fr1 = st.inv_freq[(f0 + 1) & 63];
    f32x2 sc0 = {1.f, 1.f}, sc1 = {1.f, 1.f};
    if constexpr (is_fp8<WT>::value) {
        sc0 = *reinterpret_cast<const f32x2*>(ly.qkv_s + head * 128 + f0);
        sc1 = *reinterpret_cast<const f32x2*>(ly.qkv_s + head * 128 + f1);
    }
    __builtin_amdgcn_sched_barrier(0);
    const WT* wp = reinterpret_cast<const WT*>(ly.qkv_w) + ((size_t)tile * KS) * 64 + lane_slot<WT>(g, FULL ? m : (m & 7) + 8 * half);
    WT a[4][NC];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int s = 4 * wv + t;
        weights_issue<NC>(a[t], wp, s * KS / 16, (s + 1) * KS / 16, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int page = st.block_table[mc * st.max_pages + (pos >> 6)];
    __builtin_amdgcn_sched_barrier(0);
    pin_rows<2, NC>(R);
    PIN(fr0); PIN(fr1); PIN(bia0); PIN(bia1);
    if constexpr (is_fp8<WT>::value) { PIN(sc0); PIN(sc1); }
    FTRACE(blk, 1);
    float rc[2] = {1.f, 1.f}, rs[2] = {0.f, 0.f};
    if (wv == 3 && rot) {
        sincosf((float)pos * fr0, &rs[0], &rc[0]);
        sincosf((float)pos * fr1, &rs[1], &rc[1]);
    }
    rows_norm_to_lds<2, NC>(R, B, H, st.eps, xs, XR, wv, 4, lane);
    PIN(page);
    __syncthreads();
    FTRACE(blk, 2);
    const bf16x8* xp = reinterpret_cast<const bf16x8*>(xs) + g * XR + (m & (XR - 1));
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int s = 4 * wv + t;
        red[s * 64 + lane] = mfma_lds<NC>(a[t], xp, 4 * XR, s * KS / 16, KS);
    }
    FTRACE(blk, 3);
    __syncthreads();
    if (!epi) return;
    f32x4 x = {0, 0, 0, 0};
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) x += red[sl * 64 + lane];
    const int key = pos & 63;
    const bool hb = ly.qkv_b != nullptr;
    const float b0 = hb ? lo_bf(bia0) : 0.f, b1 = hb ? (rot ? lo_bf(bia1) : hi_bf(bia0)) : 0.f;
    const float b2 = hb ? (rot ? hi_bf(bia0) : lo_bf(bia1)) : 0.f, b3 = hb ? hi_bf(bia1) : 0.f;
    if constexpr (is_fp8<WT>::value) {
        x[0] *= sc0[0]; x[1] *= rot ? sc1[0] : sc0[1]; x[2] *= rot ? sc0[1] : sc1[0]; x[3] *= sc1[1];
    }
    const float y[4] = {bf2f(f2bf(x[0] + b0)), bf2f(f2bf(x[1] + b1)), bf2f(f2bf(x[2] + b2)), bf2f(f2bf(x[3] + b3))};
    bf16_t* hand = st.qkvn + (size_t)m * NQKV + head * 128;          // this step's q / k / v row: the in-launch hand-off
    if (rot) {
        const int d = f0;
        const uint32_t lo = pack_bf2(y[0] * rc[0] - y[1] * rs[0], y[2] * rc[1] - y[3] * rs[1]);     // features d, d + 1
        const uint32_t hi = pack_bf2(y[1] * rc[0] + y[0] * rs[0], y[3] * rc[1] + y[2] * rs[1]);     // features d + 64, d + 65
        st4_sc1(hand + d, lo);
        st4_sc1(hand + d + 64, hi);
        if (head >= Hq) {                                            // K page append: read by LATER steps (behind a kernel boundary)
            bf16_t* kp = ly.pool + ((size_t)(page * Hkv + (head - Hq)) * 2) * PAGE_ELEMS;
            *reinterpret_cast<uint32_t*>(kp + k_chunk(key, d) * 8 + (d & 7)) = lo;
            *reinterpret_cast<uint32_t*>(kp + k_chunk(key, d + 64) * 8 + (d & 7)) = hi;
        }
    } else {
        const u32x2 pk = {pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3])};
        st8_sc1(hand + f0, pk);
        bf16_t* vp = ly.pool + ((size_t)(page * Hkv + (head - Hq - Hkv)) * 2 + 1) * PAGE_ELEMS;
#pragma unroll
        for (int r = 0; r < 4; ++r) vp[v_off(key, f0 + r)] = f2bf(y[r]);
    }
    FTRACE(blk, 4);
    FLOW_DRAIN();
    FTRACE(blk, 5);
    if (lane == 0) flow_arrive(edge(ly.sync, E_QKV), blk, n_self);
    FTRACE(blk, 6);
}

// ------------------------------------------------------------------------------------------------ role: attention
// decode_attn_kernel<4> (decode.hip): workgroup = (split, kv head, sequence); wave w walks pages split*4 + w, += 4 n_splits.
// Waves 1-3 request their first page before the hand-off; wave 0 polls READY(qkv), fetches q and the new K / V row (st.qkvn) into
// LDS for everybody, and requests its page after that.  The partials go to the combine kernel behind a kernel boundary.
constexpr int AT_NW = 4, AT_LD = 132;
DEVI void role_attn(const FlowStep& st, const FlowLayer& ly, int blk, int gb, char* smem) {
    const int Hq = st.Hq, Hkv = st.Hkv, n_splits = st.n_splits, group = Hq / Hkv, NQKV = (Hq + 2 * Hkv) * 128;
    const int split = blk % n_splits, hkv = (blk / n_splits) % Hkv, b = blk / (n_splits * Hkv);
    float* lds_o = reinterpret_cast<float*>(smem);                      // [NW][16][AT_LD]
    float* lds_m = lds_o + AT_NW * 16 * AT_LD;                          // [NW][16]
    float* lds_l = lds_m + AT_NW * 16;
    u32x4* qfrag = reinterpret_cast<u32x4*>(lds_l + AT_NW * 16);        // [4][64]
    bf16_t* knew = reinterpret_cast<bf16_t*>(qfrag + 4 * 64);           // [128]
    bf16_t* vnew = knew + 128;                                          // [128]
    const int l = threadIdx.x & 63, w = wave_id(), i = l & 15, g = l >> 4;
    const int ctx = st.ctx_len[b] + 1;                                  // includes the token appended this step
    const int n_pages = (ctx + PAGE - 1) / PAGE;
    const int* table = st.block_table + (size_t)b * st.max_pages;
    if (split * AT_NW >= n_pages) return;                               // no page for this split (the combine kernel skips it too)
    FTRACE(gb, 0);
    int p = split * AT_NW + w;
    bf16x8 kf[16], vf[16];
    auto load_page = [&](int pg) {
        const bf16x8* kp = reinterpret_cast<const bf16x8*>(ly.pool + ((size_t)(pg * Hkv + hkv) * 2) * PAGE_ELEMS) + l;
        const bf16x8* vp = kp + PAGE_ELEMS / 8;
#pragma unroll
        for (int c = 0; c < 16; ++c) kf[c] = __builtin_nontemporal_load(kp + c * 64);
#pragma unroll
        for (int c = 0; c < 16; ++c) vf[c] = __builtin_nontemporal_load(vp + c * 64);
    };
    const int pg0 = table[min(p, n_pages - 1)];
    if (w != 0) {
        flow_delay(st.delay_attn);
        if (p < n_pages) load_page(pg0);
    } else {
        // ---- hand-off: q (group heads), k, v of this step's token
        flow_poll(edge(ly.sync, E_QKV), st.err, blk);
        FTRACE(gb, 1);
        const bf16_t* row = st.qkvn + (size_t)b * NQKV;
        u32x4 qraw[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qraw[kk] = ld16_sc1(row, (uint32_t)(((hkv * group + min(i, group - 1)) * 128 + kk * 32 + g * 8) * 2));
        // lanes 0-15: k, lanes 16-31: v (16 B each)
        const u32x4 kv = ld16_sc1(row, (uint32_t)(((Hq + (l < 16 ? 0 : Hkv) + hkv) * 128 + (l & 15) * 8) * 2));
        __builtin_amdgcn_sched_barrier(0);
        if (p < n_pages) load_page(pg0);                                // behind q / k / v in this wave's queue
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(32)" ::: "memory");               // q / k / v are in (loads return in order); the page may still fly
        const u32x4 z = {0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qfrag[kk * 64 + l] = i < group ? qraw[kk] : z;
        if (l < 32) reinterpret_cast<u32x4*>(knew)[l] = kv;             // knew[0..127] | vnew[0..127] are contiguous
    }
    __syncthreads();
    FTRACE(gb, 2);
    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = __builtin_bit_cast(bf16x8, qfrag[kk * 64 + l]);
    f32x4 o[8];
#pragma unroll
    for (int dg = 0; dg < 8; ++dg) o[dg] = f32x4{0, 0, 0, 0};
    float m_run = -1e30f, l_run = 0.f;
    const int p_last = (ctx - 1) >> 6, kp_new = (ctx - 1) & 63;
    while (p < n_pages) {
        if (p == p_last) {            // wave-uniform: patch the row of this step's token into the prefetched page
            const int kg = kp_new >> 4, ki = kp_new & 15;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const u32x4 nk = *reinterpret_cast<const u32x4*>(knew + (c & 3) * 32 + g * 8);
                if ((c >> 2) == kg && i == ki) kf[c] = __builtin_bit_cast(bf16x8, nk);
            }
            const int slab = kp_new >> 5, k5 = kp_new & 31, vg = (k5 >> 2) & 3, ve = 4 * (k5 >> 4) + (k5 & 3);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const uint32_t nv = vnew[(c & 7) * 16 + i];
                u32x4 cur = __builtin_bit_cast(u32x4, vf[c]);
                const bool hit = (c >> 3) == slab && g == vg;
#pragma unroll
                for (int wd = 0; wd < 4; ++wd) {
                    const uint32_t lo16 = (cur[wd] & 0xffff0000u) | nv, hi16 = (cur[wd] & 0x0000ffffu) | (nv << 16);
                    if (hit && (ve >> 1) == wd) cur[wd] = (ve & 1) ? hi16 : lo16;
                }
                vf[c] = __builtin_bit_cast(bf16x8, cur);
            }
        }
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
        const float m_new = fmaxf(m_run, mx * st.scale_log2e);
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
                    pv[r] = __builtin_amdgcn_exp2f(fmaf(s[slab * 2 + t][r], st.scale_log2e, -m_new));
                    psum += pv[r];
                }
                pk[t * 2] = pack_bf2(pv[0], pv[1]);
                pk[t * 2 + 1] = pack_bf2(pv[2], pv[3]);
            }
            pf[slab] = __builtin_bit_cast(bf16x8, pk);
        }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int dg = 0; dg < 8; ++dg)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[dg][r] *= alpha;
#pragma unroll
        for (int slab = 0; slab < 2; ++slab)
#pragma unroll
            for (int dg = 0; dg < 8; ++dg)
                o[dg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[slab * 8 + dg], pf[slab], o[dg], 0, 0, 0);
        p += AT_NW * n_splits;
        if (p < n_pages) load_page(table[p]);
    }
    FTRACE(gb, 3);
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (g == 0) { lds_m[w * 16 + i] = m_run; lds_l[w * 16 + i] = l_run; }
    if (i < group) {
#pragma unroll
        for (int dg = 0; dg < 8; ++dg) *reinterpret_cast<f32x4*>(lds_o + (w * 16 + i) * AT_LD + dg * 16 + 4 * g) = o[dg];       // O^T[d = 16dg+4g+r][q = i]
    }
    __syncthreads();
    for (int item = threadIdx.x; item < group * 128; item += AT_NW * 64) {
        const int j = item >> 7, d = item & 127;
        float m = lds_m[j];
#pragma unroll
        for (int ww = 1; ww < AT_NW; ++ww) m = fmaxf(m, lds_m[ww * 16 + j]);
        float acc = 0.f, lsum = 0.f;
#pragma unroll
        for (int ww = 0; ww < AT_NW; ++ww) {
            const float f = __builtin_amdgcn_exp2f(lds_m[ww * 16 + j] - m);
            acc += lds_o[(ww * 16 + j) * AT_LD + d] * f;
            lsum += lds_l[ww * 16 + j] * f;
        }
        const size_t base = (((size_t)b * Hkv + hkv) * n_splits + split) * group + j;
        st.part_o[base * 128 + d] = acc;
        if (d == 0) { st.part_ml[base * 2] = m; st.part_ml[base * 2 + 1] = lsum; }
    }
    FTRACE(gb, 4);
}

// ------------------------------------------------------------------------------------------------ role: o_proj
// dec_proj_kernel (decode_fused.hip): h[m][n] += sum_k X[m][k] W[n][k], one 8-row half tile per workgroup over the full K
// (= Nq <= 2048: at most 4 k-steps per slice).  The 16 K-slices of the 1024-thread kernel are kept (slice s = k-steps
// [s KS / 16, (s+1) KS / 16), even / odd k-steps of a slice on two accumulators, the 16 partial sums added in order): wave w
// computes slices 4w .. 4w+3.  First role of its launch (X = att comes from the combine kernel): it publishes h.
template <typename WT>
DEVI void role_o(const FlowStep& st, const FlowLayer& ly, int blk, int n_self, int gb, char* smem) {
    constexpr int G = 4;
    f32x4* red = reinterpret_cast<f32x4*>(smem);                                  // [16][64]
    const int B = st.B, N = st.H, K = st.Hq * 128, XR = st.XR;
    const int lane = threadIdx.x & 63, wv = wave_id();
    const int half = blk & 1, tile = blk >> 1;
    const int KS = K / 32;
    const int m = lane & 15, g = lane >> 4;
    const bool epi = wv == 3 && m < B && g < 2;
    const WT* wp = reinterpret_cast<const WT*>(ly.o_w) + ((size_t)tile * KS) * 64 + lane_slot<WT>(g, (m & 7) + 8 * half);
    bf16_t* hp = st.h + (size_t)min(m, B - 1) * N + tile * 16 + 8 * half + 4 * (g & 1);
    const bf16x8* xp = reinterpret_cast<const bf16x8*>(st.att) + g * XR + (m & (XR - 1));
    FTRACE(gb, 0);
    const u32x2 res = *reinterpret_cast<const u32x2*>(hp);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f};
    if constexpr (is_fp8<WT>::value) sc = *reinterpret_cast<const f32x4*>(ly.o_s + tile * 16 + 8 * half + 4 * (g & 1));
    bf16x8 x[4][G];
    WT a[4][G];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k0 = (int)((uint32_t)((4 * wv + t) * KS) >> 4);
#pragma unroll
        for (int jj = 0; jj < G; ++jj) x[t][jj] = xp[(size_t)min(k0 + jj, KS - 1) * 4 * XR];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int s = 4 * wv + t;
        weights_issue<G>(a[t], wp, (int)((uint32_t)(s * KS) >> 4), (int)((uint32_t)((s + 1) * KS) >> 4), lane);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
        for (int jj = 0; jj < G; jj += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_a(a[t][jj]), x[t][jj], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_a(a[t][jj + 1]), x[t][jj + 1], acc1, 0, 0, 0);
        }
        red[(4 * wv + t) * 64 + lane] = acc0 + acc1;
    }
    FTRACE(gb, 1);
    PIN(res);
    if constexpr (is_fp8<WT>::value) PIN(sc);
    __syncthreads();
    if (!epi) return;
    f32x4 sum = {0, 0, 0, 0};
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) sum += red[sl * 64 + lane];
    if constexpr (is_fp8<WT>::value) sum *= sc;
    const u32x2 o = {pack_bf2(lo_bf(res[0]) + sum[0], hi_bf(res[0]) + sum[1]), pack_bf2(lo_bf(res[1]) + sum[2], hi_bf(res[1]) + sum[3])};
    st8_sc1(hp, o);
    FTRACE(gb, 4);
    FLOW_DRAIN();
    FTRACE(gb, 5);
    if (lane == 0) flow_arrive(edge(ly.sync, E_O), blk, n_self);
}

// ------------------------------------------------------------------------------------------------ role: gate | up
// dec_gateup_kernel<2, NC, WT, 1> (decode_fused.hip).  Waves 1-3 request their whole K slice (<= 12 k-steps of the gate and of the
// up tile) before the hand-off; wave 0 polls READY(o_proj), copies the residual rows global -> LDS by DMA (no registers, sc1) and
// requests its slice after that; every wave then normalises its two rows from the LDS copy.
constexpr int FGU_G = 12, FGU_WAVES = 4;
template <int NC, typename WT>
DEVI void role_gu(const FlowStep& st, const FlowLayer& ly, int blk, int gb, char* smem) {
    const int B = st.B, H = st.H, XR = st.XR;
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                                 // X image [H/8][XR][8]
    char* raw = smem + (size_t)XR * H * 2;                                        // residual rows [8][NC][1 KiB]
    f32x4* red = reinterpret_cast<f32x4*>(raw);                                   // [4][2][64]: the rows are dead when the partial sums are written
    const int lane = threadIdx.x & 63, wv = wave_id();
    const int kw = wv, Gp = blk >> 1, ab = blk & 1;
    const int KS = H / 32;
    const int k0 = kw * KS / FGU_WAVES, k1 = (kw + 1) * KS / FGU_WAVES;
    const bf16x8* xp = reinterpret_cast<const bf16x8*>(xs) + (lane >> 4) * XR + (lane & (XR - 1));
    const int xstride = 4 * XR;
    const int ls = lane_slot<WT>(lane >> 4, lane & 15);
    const WT* wg = reinterpret_cast<const WT*>(ly.w13) + ((size_t)(Gp * 4 + ab) * KS) * 64 + ls;
    const WT* wu = reinterpret_cast<const WT*>(ly.w13) + ((size_t)(Gp * 4 + 2 + ab) * KS) * 64 + ls;
    FTRACE(gb, 0);
    u32x4 lw[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) lw[c] = *reinterpret_cast<const u32x4*>(ly.ln2 + min(c * 512 + lane * 8, H - 8));
    f32x4 scg = {1.f, 1.f, 1.f, 1.f}, scu = {1.f, 1.f, 1.f, 1.f};
    if constexpr (is_fp8<WT>::value) {
        scg = *reinterpret_cast<const f32x4*>(ly.w13_s + (Gp * 4 + ab) * 16 + 4 * (lane >> 4));
        scu = *reinterpret_cast<const f32x4*>(ly.w13_s + (Gp * 4 + 2 + ab) * 16 + 4 * (lane >> 4));
    }
    __builtin_amdgcn_sched_barrier(0);
    WT a_[FGU_G], u_[FGU_G];
    const WT* zc = reinterpret_cast<const WT*>(g_zero_chunk) + lane;
    auto issue = [&]() {
#pragma unroll
        for (int jj = 0; jj < FGU_G; ++jj) {
            a_[jj] = __builtin_nontemporal_load(k0 + jj < k1 ? wg + (size_t)(k0 + jj) * 64 : zc);
            u_[jj] = __builtin_nontemporal_load(k0 + jj < k1 ? wu + (size_t)(k0 + jj) * 64 : zc);
        }
    };
    if (wv != 0) {
        flow_delay(st.delay_gu);
        issue();
    } else {
        flow_poll(edge(ly.sync, E_O), st.err, blk);
        FTRACE(gb, 1);
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < NC; ++c)
                dma1k_sc1(st.h + (size_t)min(r, B - 1) * H + min(c * 512 + lane * 8, H - 8), raw + (r * NC + c) * 1024);
        __builtin_amdgcn_sched_barrier(0);
        issue();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * FGU_G) : "memory");         // the rows are in LDS (loads return in order)
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    FTRACE(gb, 2);
#pragma unroll
    for (int c = 0; c < NC; ++c) PIN(lw[c]);
    if constexpr (is_fp8<WT>::value) { PIN(scg); PIN(scu); }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wv + i * FGU_WAVES;
        if (r < B) {
            u32x4 v[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) v[c] = *reinterpret_cast<const u32x4*>(raw + (r * NC + c) * 1024 + lane * 16);
            row_norm_to_lds<NC>(v, lw, r, H, st.eps, xs, XR, lane);
        }
    }
    __syncthreads();
    FTRACE(gb, 3);
    f32x4 ag = {0, 0, 0, 0}, au = {0, 0, 0, 0};
#pragma unroll
    for (int jj = 0; jj < FGU_G; ++jj) {
        const bf16x8 bx = xp[(size_t)min(k0 + jj, KS - 1) * xstride];
        ag = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_a(a_[jj]), bx, ag, 0, 0, 0);
        au = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_a(u_[jj]), bx, au, 0, 0, 0);
    }
    red[(wv * 2) * 64 + lane] = ag;
    red[(wv * 2 + 1) * 64 + lane] = au;
    __syncthreads();
    const int m = lane & 15, g = lane >> 4;
    if (kw != 0 || m >= B) return;
    f32x4 gs = ag, us = au;
#pragma unroll
    for (int ww = 1; ww < FGU_WAVES; ++ww) { gs += red[(ww * 2) * 64 + lane]; us += red[(ww * 2 + 1) * 64 + lane]; }
    if constexpr (is_fp8<WT>::value) { gs *= scg; us *= scu; }
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = gs[r] / (1.0f + __expf(-gs[r])) * us[r];
    store_frag4(st.act, m, Gp * 32 + ab * 16 + 4 * g, XR, o[0], o[1], o[2], o[3]);
    FTRACE(gb, 4);
}

// ------------------------------------------------------------------------------------------------ the kernels
// [qkv (whole tiles) -> attention]; the partials go to the combine kernel of decode.hip behind a kernel boundary
template <typename WT>
__global__ __launch_bounds__(256, 2) void flow_qkv_attn_kernel(FlowStep st, FlowLayer ly, int n_qkv) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    if (b < n_qkv) role_qkv<NC_MAX, WT>(st, ly, b, n_qkv, smem);
    else role_attn(st, ly, b - n_qkv, b, smem);
}
// [o_proj -> gate|up]; 3 workgroups per CU: all of them resident from the start
template <typename WT>
__global__ __launch_bounds__(256, 3) void flow_o_gu_kernel(FlowStep st, FlowLayer ly, int n_o) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    if (b < n_o) role_o<WT>(st, ly, b, n_o, b, smem);
    else role_gu<NC_MAX, WT>(st, ly, b - n_o, b, smem);
}

}  // namespace

#ifdef DOTS_TRACE
void dots_trace_set_flow(unsigned long long* buf) { (void)hipMemcpyToSymbol(HIP_SYMBOL(dots_trace_buf), &buf, sizeof(buf)); }
#endif

// ===================================================================================================== host side
size_t flow_sync_bytes_per_layer() { return (size_t)FLOW_EDGES * FLOW_SYNC_WORDS * sizeof(uint32_t); }

bool flow_supported(int B, int H, int Hq, int Hkv, int I) {
    if (B < 1 || B > 8) return false;                                    // one 8-row X image; larger batches use the launch-per-phase kernels
    if (H % 128 || H > 512 * NC_MAX || H / 32 < FGU_WAVES || H / 32 > FGU_G * FGU_WAVES) return false;
    if (Hq % Hkv || Hq / Hkv > 16) return false;
    if ((Hq * 128) / 32 > 16 * 4 || (Hq * 128) / 32 < 16 || I % 32 || I / 32 < 16) return false;
    return true;
}

static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

// what: 1 = the whole layer  [qkv -> attention] | combine | [o_proj -> gate|up] | down_proj;  22 / 23 = the first / second fused launch
// alone (tools/decode_bench).  A.sync: this layer's block (flow_sync_bytes_per_layer() bytes, zeroed for this step).
hipError_t launch_decode_layer_flow(hipStream_t s, int what, const FlowLayerArgs& A) {
    if (!flow_supported(A.B, A.H, A.Hq, A.Hkv, A.I)) return hipErrorInvalidValue;
    static const int d_attn = env_int("DOTS_OCR_FLOW_DELAY_ATTN", 6), d_gu = env_int("DOTS_OCR_FLOW_DELAY_GU", 6);
    FlowStep st{};
    st.h = A.h; st.qkvn = A.qkvn; st.part_o = A.part_o; st.part_ml = A.part_ml; st.att = A.att; st.act = A.act;
    st.inv_freq = A.inv_freq; st.ctx_len = A.ctx_len; st.block_table = A.block_table; st.err = A.err;
    st.max_pages = A.max_pages; st.B = A.B; st.H = A.H; st.Hq = A.Hq; st.Hkv = A.Hkv; st.I = A.I; st.n_splits = A.n_splits; st.XR = 8;
    st.eps = A.eps; st.scale_log2e = A.scale * 1.44269504088896340736f;
    st.delay_attn = d_attn; st.delay_gu = d_gu;
    FlowLayer ly{};
    ly.ln1 = A.ln1; ly.ln2 = A.ln2; ly.qkv_b = A.qkv_b; ly.qkv_w = A.qkv_w; ly.o_w = A.o_w; ly.w13 = A.w13;
    ly.qkv_s = A.qkv_s; ly.o_s = A.o_s; ly.w13_s = A.w13_s;
    ly.pool = A.pool; ly.sync = A.sync;
    const int n_qkv = (A.Hq + 2 * A.Hkv) * 8, n_attn = A.n_splits * A.Hkv * A.B, n_o = A.H / 8, n_gu = A.I / 16;
    const size_t lds_x = (size_t)8 * A.H * 2;
    const size_t lds_a = std::max(lds_x + 16 * 64 * sizeof(f32x4),                                                  // qkv: X image + 16 partial tiles
                                  (size_t)(AT_NW * 16 * AT_LD + 2 * AT_NW * 16) * 4 + 4 * 64 * 16 + 512);           // attention
    const size_t lds_b = std::max(lds_x + std::max((size_t)8 * NC_MAX * 1024, 2 * FGU_WAVES * 64 * sizeof(f32x4)), (size_t)16 * 64 * sizeof(f32x4));
    if (lds_a > 64 * 1024 || lds_b > 64 * 1024) return hipErrorInvalidValue;
    if (what != 23) {
        if (A.qkv_s) hipLaunchKernelGGL(flow_qkv_attn_kernel<u32x2>, dim3(n_qkv + n_attn), dim3(256), lds_a, s, st, ly, n_qkv);
        else hipLaunchKernelGGL(flow_qkv_attn_kernel<bf16x8>, dim3(n_qkv + n_attn), dim3(256), lds_a, s, st, ly, n_qkv);
    }
    if (what == 22) return hipGetLastError();
    hipError_t e = hipSuccess;
    if (what != 23 && (e = launch_decode_attn_combine(s, A.part_o, A.part_ml, A.ctx_len, A.att, A.B, A.Hq, A.Hkv, A.n_splits)) != hipSuccess) return e;
    if (A.qkv_s) hipLaunchKernelGGL(flow_o_gu_kernel<u32x2>, dim3(n_o + n_gu), dim3(256), lds_b, s, st, ly, n_o);
    else hipLaunchKernelGGL(flow_o_gu_kernel<bf16x8>, dim3(n_o + n_gu), dim3(256), lds_b, s, st, ly, n_o);
    if (what == 23) return hipGetLastError();
    return launch_dec_proj(s, A.act, A.down_w, A.down_s, A.h, A.B, A.H, A.I);
}
