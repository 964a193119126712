// Fused dense kernels of the decode step: 6 launches per layer instead of 9.
//
// Round-1 profile of the first decode design (split-K slabs + separate reduce/norm/rope kernels) showed the
// step bound by dependent-launch latency, not HBM: ~4.4 us per tiny kernel x 5 tiny kernels per layer.
// Here every reduction stays inside a workgroup and every row-wise op rides in a GEMM prologue/epilogue:
//
//   dec_qkv      rmsnorm(h) prologue -> qkv GEMM -> bias + RoPE + q store + K/V page append epilogue
//   (decode_attn + combine: decode.hip)
//   dec_proj     o_proj / down_proj GEMM -> h += result                       (residual epilogue, no norm)
//   dec_gateup   rmsnorm(h) prologue -> gate/up GEMM -> SwiGLU epilogue -> fragment-order activations
//   dec_lmhead   rmsnorm(h) prologue -> lm_head GEMM -> fp32 logits
//
// The consumer of a residual-stream row recomputes its RMS statistic itself (8 rows x 3 KB from L2, one wave
// per row) and writes the normalised rows in MFMA fragment order into LDS, from where every wave takes its B
// operand with one ds_read_b128 per MFMA; numerics are identical to the standalone norm kernel
// (bf16(bf16(x*rstd)*w)).  Weights stream from HBM in fragment order straight into the A operand (decode.hip).
// Split-K happens across the waves of ONE workgroup (up to 16) and is reduced through LDS in a fixed order:
// deterministic, no slabs in HBM, no atomics.  N = 1536 projections run as 96 workgroups x 16 waves: a CU with
// 16 waves x 8 KiB in flight sustains its share of HBM bandwidth.
#include "common.h"
#include "decode_layout.h"
#include "kernels.h"

namespace {

// Residual-stream row as the consumer sees it:  x = bf16(h[r] + sum_s slab[s][r])  (n_slabs may be 0), then
// X = rmsnorm(x) * w for rows r < B, written to LDS in fragment order; workgroup 0 also stores x to h_out (the
// producer of the slabs — a K-split projection — leaves the residual add to its consumer; h_out != h, ping-pong).
// LDS image: [K/8][XR][8] with XR = 8 (B <= 8: lanes m >= 8 alias rows m-8, half the LDS, twice the occupancy) or 16.
// Rows >= B are left untouched: column m of the MFMA result depends only on row m of X and columns >= B are never stored.
DEVI void norm_rows_to_lds(const bf16_t* __restrict__ h, const float* __restrict__ slabs, int n_slabs, bf16_t* __restrict__ h_out,
                           const bf16_t* __restrict__ w, int B, int dim, float eps,
                           bf16_t* __restrict__ xs, int XR, int wave, int n_waves, int lane) {
    for (int r = wave; r < B; r += n_waves) {
        const bf16_t* row = h + (size_t)r * dim;
        u32x4 v[4];
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = c * 512 + lane * 8;
            if (k < dim) {
                v[c] = *reinterpret_cast<const u32x4*>(row + k);
                if (n_slabs > 0) {
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { f[2 * e] = lo_bf(v[c][e]); f[2 * e + 1] = hi_bf(v[c][e]); }
                    float add[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    for (int sidx = 0; sidx < n_slabs; ++sidx) {
                        const float* sp = slabs + ((size_t)sidx * 16 + r) * dim + k;
                        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { add[e] += s0[e]; add[4 + e] += s1[e]; }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[c][e] = pack_bf2(f[2 * e] + add[2 * e], f[2 * e + 1] + add[2 * e + 1]);
                    if (blockIdx.x == 0) *reinterpret_cast<u32x4*>(h_out + (size_t)r * dim + k) = v[c];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float a = lo_bf(v[c][e]), b = hi_bf(v[c][e]); ss += a * a + b * b; }
            }
        }
        const float rstd = rsqrtf(wave_sum(ss) / dim + eps);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = c * 512 + lane * 8;
            if (k < dim) {
                const u32x4 ww = *reinterpret_cast<const u32x4*>(w + k);
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = pack_bf2(bf2f(f2bf(lo_bf(v[c][e]) * rstd)) * lo_bf(ww[e]), bf2f(f2bf(hi_bf(v[c][e]) * rstd)) * hi_bf(ww[e]));
                *reinterpret_cast<u32x4*>(xs + ((size_t)(k >> 3) * XR + r) * 8) = o;      // [k/8][XR rows][8]
            }
        }
    }
}

// First group of (up to 8) weight chunks of a wave's K-slice: issued BEFORE the norm prologue so HBM latency runs under it.
DEVI void preload_group(const bf16x8* __restrict__ wp, int k0, int k1, bf16x8 (&a)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (k0 + j < k1) a[j] = __builtin_nontemporal_load(wp + (size_t)(k0 + j) * 64);
}

// acc = W-tile[k0..k1) . X, A from global (fragment order, non-temporal; first group already in `a`), B from LDS or
// global (fragment order).  The next group's weight loads are issued before this group's MFMAs.
DEVI f32x4 stream_tile(const bf16x8* __restrict__ wp, const bf16x8* xp, int xstride, int k0, int k1, bf16x8 (&a)[8]) {
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    for (int ks = k0; ks < k1; ks += 8) {
        bf16x8 b[8], an[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (ks + j < k1) b[j] = xp[(size_t)(ks + j) * xstride];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (ks + 8 + j < k1) an[j] = __builtin_nontemporal_load(wp + (size_t)(ks + 8 + j) * 64);
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            if (ks + j < k1) acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b[j], acc0, 0, 0, 0);
            if (ks + j + 1 < k1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j + 1], b[j + 1], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = an[j];
    }
    return acc0 + acc1;
}

// Same contraction for slices of <= 8 k-steps per round WITHOUT the next-group prefetch registers (the 16-wave qkv
// workgroup is capped at 128 VGPRs; its slices are 6 k-steps at H = 1536, so one round is the whole slice).
DEVI f32x4 stream_tile_lean(const bf16x8* __restrict__ wp, const bf16x8* xp, int xstride, int k0, int k1, bf16x8 (&a)[8]) {
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    for (int ks = k0; ks < k1; ks += 8) {
        if (ks != k0) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (ks + j < k1) a[j] = __builtin_nontemporal_load(wp + (size_t)(ks + j) * 64);
        }
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            if (ks + j < k1) acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], xp[(size_t)(ks + j) * xstride], acc0, 0, 0, 0);
            if (ks + j + 1 < k1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j + 1], xp[(size_t)(ks + j + 1) * xstride], acc1, 0, 0, 0);
        }
    }
    return acc0 + acc1;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_embed_kernel(const int32_t* __restrict__ tokens, const bf16_t* __restrict__ embed,
                                                        bf16_t* __restrict__ h, int dim) {
    const int b = blockIdx.x;
    const bf16_t* src = embed + (size_t)tokens[b] * dim;
    for (int k = threadIdx.x * 8; k < dim; k += 256 * 8)
        *reinterpret_cast<u32x4*>(h + (size_t)b * dim + k) = *reinterpret_cast<const u32x4*>(src + k);
}

// ------------------------------------------------------------------------------------------------
// grid = (Hq + 2 Hkv) * 4 workgroups of 16 waves.  Workgroup (head, t): weight tiles (8*head + t) and (8*head + t + 4),
// i.e. features d in [16t, 16t+16) and their RoPE partners d + 64; wave = (tile, one of 8 K-slices).
__global__ __launch_bounds__(1024) void dec_qkv_kernel(const bf16_t* __restrict__ h, const float* __restrict__ slabs, int n_slabs,
                                                       bf16_t* __restrict__ h_out, const bf16_t* __restrict__ ln_w,
                                                       const bf16_t* __restrict__ Wd, const bf16_t* __restrict__ bias,
                                                       const float* __restrict__ inv_freq, const int32_t* __restrict__ ctx_len,
                                                       const int32_t* __restrict__ block_table, int max_pages,
                                                       bf16_t* __restrict__ pool, bf16_t* __restrict__ q_out,
                                                       int B, int H, int Hq, int Hkv, float eps, int XR) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                                 // [H/8][XR][8]
    f32x4* red = reinterpret_cast<f32x4*>(smem + (size_t)XR * H * 2);             // [16 waves][64 lanes]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int head = blockIdx.x >> 2, t = blockIdx.x & 3;
    const int sel = wv & 1, slice = wv >> 1;
    const int KS = H / 32;
    const int n_tile = head * 8 + t + 4 * sel;
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(Wd) + ((size_t)n_tile * KS) * 64 + lane;
    const int k0 = slice * KS / 8, k1 = (slice + 1) * KS / 8;
    bf16x8 a0[8];
    preload_group(wp, k0, k1, a0);
    // RoPE angles of the epilogue lanes (wave 0), computed while the first weight group is in flight: precise
    // sincosf of a large angle takes the slow range-reduction path
    float rc[4] = {1.f, 1.f, 1.f, 1.f}, rs[4] = {0.f, 0.f, 0.f, 0.f};
    if (wv == 0 && (lane & 15) < B && head < Hq + Hkv) {
        const int pos0 = ctx_len[lane & 15];
#pragma unroll
        for (int r = 0; r < 4; ++r) sincosf((float)pos0 * inv_freq[16 * t + 4 * (lane >> 4) + r], &rs[r], &rc[r]);
    }
    norm_rows_to_lds(h, slabs, n_slabs, h_out, ln_w, B, H, eps, xs, XR, wv, 16, lane);
    __syncthreads();
    red[wv * 64 + lane] = stream_tile_lean(wp, reinterpret_cast<const bf16x8*>(xs) + (lane >> 4) * XR + (lane & (XR - 1)), 4 * XR, k0, k1, a0);
    __syncthreads();
    if (wv != 0) return;
    f32x4 x1 = {0, 0, 0, 0}, x2 = {0, 0, 0, 0};
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) { x1 += red[(2 * sl) * 64 + lane]; x2 += red[(2 * sl + 1) * 64 + lane]; }
    const int m = lane & 15, g = lane >> 4;
    if (m >= B) return;
    const int d0 = 16 * t + 4 * g;                         // this lane: features d0..d0+3 and d0+64..d0+67 of `head`
    const int pos = ctx_len[m];
    const int page = block_table[m * max_pages + (pos >> 6)];
    const int key = pos & 63;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int d = d0 + r;
        float a = x1[r], b = x2[r];
        if (bias) { a += bf2f(bias[head * 128 + d]); b += bf2f(bias[head * 128 + d + 64]); }
        a = bf2f(f2bf(a)); b = bf2f(f2bf(b));              // the qkv projection output is a bf16 tensor
        if (head < Hq + Hkv) {
            const float sn = rs[r], cs = rc[r];
            const bf16_t o1 = f2bf(a * cs - b * sn), o2 = f2bf(b * cs + a * sn);
            if (head < Hq) {
                q_out[(size_t)m * Hq * 128 + head * 128 + d] = o1;
                q_out[(size_t)m * Hq * 128 + head * 128 + d + 64] = o2;
            } else {
                bf16_t* kp = pool + ((size_t)(page * Hkv + (head - Hq)) * 2) * PAGE_ELEMS;
                kp[k_chunk(key, d) * 8 + (d & 7)] = o1;
                kp[k_chunk(key, d + 64) * 8 + (d & 7)] = o2;
            }
        } else {
            bf16_t* vp = pool + ((size_t)(page * Hkv + (head - Hq - Hkv)) * 2 + 1) * PAGE_ELEMS;
            vp[v_off(key, d)] = f2bf(a);
            vp[v_off(key, d + 64)] = f2bf(b);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// o_proj / down_proj.  grid (N/16, ksplit) workgroups x 16 waves; wave = one of 16 K-slices of its workgroup's K-part.
//   ksplit == 1 : h[m][n] += sum_k X[m][k] W[n][k]                      (residual epilogue in place)
//   ksplit  > 1 : slab[part][m][n] = partial sum (fp32); the consumer's prologue folds h + sum(slabs) in fixed order.
// More workgroups = more CUs streaming: a CU sustains only ~25 GB/s with 16 waves x 8 KiB in flight (round-1 profile),
// so the 27.5 MB down projection needs > 96 of them.
__global__ __launch_bounds__(1024) void dec_proj_kernel(const bf16_t* __restrict__ Xf, const bf16_t* __restrict__ Wd,
                                                        bf16_t* __restrict__ h, float* __restrict__ slabs, int B, int N, int K) {
    __shared__ f32x4 red[16 * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n_tile = blockIdx.x, part = blockIdx.y, parts = gridDim.y;
    const int KS = K / 32;
    const int slice = part * 16 + wv, slices = parts * 16;
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(Wd) + ((size_t)n_tile * KS) * 64 + lane;
    const int k0 = (int)((int64_t)slice * KS / slices), k1 = (int)((int64_t)(slice + 1) * KS / slices);
    bf16x8 a0[8];
    preload_group(wp, k0, k1, a0);
    red[wv * 64 + lane] = stream_tile(wp, reinterpret_cast<const bf16x8*>(Xf) + lane, 64, k0, k1, a0);
    __syncthreads();
    if (wv != 0) return;
    f32x4 a = {0, 0, 0, 0};
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) a += red[sl * 64 + lane];
    const int m = lane & 15, g = lane >> 4;
    if (m >= B) return;
    if (parts > 1) {
        *reinterpret_cast<f32x4*>(slabs + ((size_t)part * 16 + m) * N + n_tile * 16 + 4 * g) = a;
        return;
    }
    bf16_t* hp = h + (size_t)m * N + n_tile * 16 + 4 * g;
    const u32x2 x = *reinterpret_cast<const u32x2*>(hp);
    const u32x2 o = {pack_bf2(lo_bf(x[0]) + a[0], hi_bf(x[0]) + a[1]), pack_bf2(lo_bf(x[1]) + a[2], hi_bf(x[1]) + a[3])};
    *reinterpret_cast<u32x2*>(hp) = o;
}

// ------------------------------------------------------------------------------------------------
// act = silu(gate) * up with gate/up = rmsnorm(h) @ W13^T.  grid I/16 workgroups x GU_WAVES waves (K-slices);
// workgroup = one (gate tile, up tile) pair of the packed W13 (64-row groups: 32 gate rows | 32 up rows).

constexpr int GU_G = 12;         // k-steps per wave held in registers (H = 1536: 48 k-steps / 4 waves)
constexpr int GU_WAVES = 4;     // 12 (single round) was measured slower: 12 waves x 49 KB LDS leaves 560 workgroups non-resident

__global__ __launch_bounds__(GU_WAVES * 64) void dec_gateup_kernel(const bf16_t* __restrict__ h, const bf16_t* __restrict__ ln_w,
                                                                   const bf16_t* __restrict__ Wd, bf16_t* __restrict__ act,
                                                                   int B, int H, int I, float eps, int XR) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
    f32x4* red = reinterpret_cast<f32x4*>(smem + (size_t)XR * H * 2);             // [GU_WAVES][2][64]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int pair = blockIdx.x, G = pair >> 1, a = pair & 1;
    const int KS = H / 32;
    const int k0 = wv * KS / GU_WAVES, k1 = (wv + 1) * KS / GU_WAVES;
    const bf16x8* xp = reinterpret_cast<const bf16x8*>(xs) + (lane >> 4) * XR + (lane & (XR - 1));
    const int xstride = 4 * XR;
    const bf16x8* wg = reinterpret_cast<const bf16x8*>(Wd) + ((size_t)(G * 4 + a) * KS) * 64 + lane;
    const bf16x8* wu = reinterpret_cast<const bf16x8*>(Wd) + ((size_t)(G * 4 + 2 + a) * KS) * 64 + lane;
    // the wave's WHOLE K-slice (<= GU_G k-steps x (gate, up) = 24 KiB) is in flight before (and under) the norm prologue:
    // one HBM round trip per wave instead of three
    bf16x8 a_[GU_G], u_[GU_G];
#pragma unroll
    for (int j = 0; j < GU_G; ++j)
        if (k0 + j < k1) {
            a_[j] = __builtin_nontemporal_load(wg + (size_t)(k0 + j) * 64);
            u_[j] = __builtin_nontemporal_load(wu + (size_t)(k0 + j) * 64);
        }
    norm_rows_to_lds(h, nullptr, 0, nullptr, ln_w, B, H, eps, xs, XR, wv, GU_WAVES, lane);
    __syncthreads();
    f32x4 ag = {0, 0, 0, 0}, au = {0, 0, 0, 0};
    for (int ks = k0; ks < k1; ks += GU_G) {       // one trip when the slice fits (H <= 1536)
#pragma unroll
        for (int j = 0; j < GU_G; ++j)
            if (ks + j < k1) {
                const bf16x8 b = xp[(size_t)(ks + j) * xstride];
                ag = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_[j], b, ag, 0, 0, 0);
                au = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u_[j], b, au, 0, 0, 0);
            }
        if (ks + GU_G < k1) {
#pragma unroll
            for (int j = 0; j < GU_G; ++j)
                if (ks + GU_G + j < k1) {
                    a_[j] = __builtin_nontemporal_load(wg + (size_t)(ks + GU_G + j) * 64);
                    u_[j] = __builtin_nontemporal_load(wu + (size_t)(ks + GU_G + j) * 64);
                }
        }
    }
    red[(wv * 2) * 64 + lane] = ag;
    red[(wv * 2 + 1) * 64 + lane] = au;
    __syncthreads();
    const int m = lane & 15, g = lane >> 4;
    if (wv != 0 || m >= B) return;
    f32x4 gs = ag, us = au;
#pragma unroll
    for (int ww = 1; ww < GU_WAVES; ++ww) { gs += red[(ww * 2) * 64 + lane]; us += red[(ww * 2 + 1) * 64 + lane]; }
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = gs[r] / (1.0f + __expf(-gs[r])) * us[r];
    store_frag4(act, m, G * 32 + a * 16 + 4 * g, o[0], o[1], o[2], o[3]);
}

// ------------------------------------------------------------------------------------------------
// logits[m][n] = rmsnorm(h[m]) . lm_head[n]   (fp32, never rounded).  grid ceil(V/64) workgroups x 4 waves, wave = one
// 16-row vocabulary tile over the full K.
__global__ __launch_bounds__(256) void dec_lmhead_kernel(const bf16_t* __restrict__ h, const float* __restrict__ slabs, int n_slabs,
                                                         bf16_t* __restrict__ h_out, const bf16_t* __restrict__ ln_w,
                                                         const bf16_t* __restrict__ Wd, float* __restrict__ logits,
                                                         int B, int H, int V, float eps, int XR) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n_tile = min(blockIdx.x * 4 + wv, V / 16 - 1);          // tail waves recompute the last tile (same values)
    const int KS = H / 32;
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(Wd) + ((size_t)n_tile * KS) * 64 + lane;
    bf16x8 a0[8];
    preload_group(wp, 0, KS, a0);
    norm_rows_to_lds(h, slabs, n_slabs, h_out, ln_w, B, H, eps, xs, XR, wv, 4, lane);
    __syncthreads();
    const f32x4 acc = stream_tile(wp, reinterpret_cast<const bf16x8*>(xs) + (lane >> 4) * XR + (lane & (XR - 1)), 4 * XR, 0, KS, a0);
    const int m = lane & 15, g = lane >> 4;
    if (m < B) *reinterpret_cast<f32x4*>(logits + (size_t)m * V + n_tile * 16 + 4 * g) = acc;
}

template <typename Kern>
hipError_t ensure_lds(Kern kern, size_t bytes, bool* done) {
    if (bytes > 64 * 1024 && !*done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        *done = true;
    }
    return hipSuccess;
}

}  // namespace

hipError_t launch_dec_embed(hipStream_t s, const int32_t* tokens, const bf16_t* embed, bf16_t* h, int B, int dim) {
    if (dim % 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(dec_embed_kernel, dim3(B), dim3(256), 0, s, tokens, embed, h, dim);
    return hipGetLastError();
}

hipError_t launch_dec_qkv(hipStream_t s, const bf16_t* h, const float* slabs, int n_slabs, bf16_t* h_out, const bf16_t* ln_w,
                          const bf16_t* Wd, const bf16_t* bias,
                          const float* inv_freq, const int32_t* ctx_len, const int32_t* block_table, int max_pages,
                          bf16_t* pool_layer, bf16_t* q_out, int B, int H, int Hq, int Hkv, float eps) {
    if (H % 256 || H > 2048) return hipErrorInvalidValue;      // 8 K-slices of whole k-steps; norm prologue covers <= 2048
    static bool attr = false;
    const int XR = B <= 8 ? 8 : 16;
    const size_t lds = (size_t)XR * H * 2 + 16 * 64 * sizeof(f32x4);
    hipError_t e = ensure_lds(dec_qkv_kernel, (size_t)16 * H * 2 + 16 * 64 * sizeof(f32x4), &attr);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(dec_qkv_kernel, dim3((Hq + 2 * Hkv) * 4), dim3(1024), lds, s, h, slabs, n_slabs, h_out, ln_w, Wd, bias, inv_freq, ctx_len,
                       block_table, max_pages, pool_layer, q_out, B, H, Hq, Hkv, eps, XR);
    return hipGetLastError();
}

hipError_t launch_dec_proj(hipStream_t s, const bf16_t* Xf, const bf16_t* Wd, bf16_t* h, float* slabs, int ksplit, int B, int N, int K) {
    if (N % 16 || K % 32 || ksplit < 1 || K / 32 < 16 * ksplit || (ksplit > 1 && !slabs)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(dec_proj_kernel, dim3(N / 16, ksplit), dim3(1024), 0, s, Xf, Wd, h, slabs, B, N, K);
    return hipGetLastError();
}

hipError_t launch_dec_gateup(hipStream_t s, const bf16_t* h, const bf16_t* ln_w, const bf16_t* W13d, bf16_t* act,
                             int B, int H, int I, float eps) {
    if (I % 32 || H % 128 || H > 2048 || H / 32 < GU_WAVES) return hipErrorInvalidValue;
    static bool attr = false;
    const int XR = B <= 8 ? 8 : 16;
    const size_t lds = (size_t)XR * H * 2 + 2 * GU_WAVES * 64 * sizeof(f32x4);
    hipError_t e = ensure_lds(dec_gateup_kernel, (size_t)16 * H * 2 + 2 * GU_WAVES * 64 * sizeof(f32x4), &attr);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(dec_gateup_kernel, dim3(I / 16), dim3(GU_WAVES * 64), lds, s, h, ln_w, W13d, act, B, H, I, eps, XR);
    return hipGetLastError();
}

hipError_t launch_dec_lmhead(hipStream_t s, const bf16_t* h, const float* slabs, int n_slabs, bf16_t* h_out, const bf16_t* ln_w,
                             const bf16_t* Wd, float* logits,
                             int B, int H, int V, float eps) {
    if (V % 16 || H % 32 || H > 2048) return hipErrorInvalidValue;
    static bool attr = false;
    const int XR = B <= 8 ? 8 : 16;
    const size_t lds = (size_t)XR * H * 2;
    hipError_t e = ensure_lds(dec_lmhead_kernel, (size_t)16 * H * 2, &attr);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(dec_lmhead_kernel, dim3((V / 16 + 3) / 4), dim3(256), lds, s, h, slabs, n_slabs, h_out, ln_w, Wd, logits, B, H, V, eps, XR);
    return hipGetLastError();
}
