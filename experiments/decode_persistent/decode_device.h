// Device helpers shared by the decode kernels (decode_fused.hip: one launch per phase; decode_persistent.hip: one
// persistent kernel per step with device-wide barriers between the phases).
#pragma once
#include "common.h"
#include "decode_layout.h"

// Write-through stores / agent-scope loads for data that another workgroup — possibly on another XCD, behind another L2 —
// reads later in the SAME kernel (csrc/probe_sync.hip measured: plain and non-temporal accesses go stale across XCDs,
// agent-scope accesses do not and cost nothing extra; agent-scope fences cost 45 us per barrier and are never used).
DEVI void st_wt_u16(bf16_t* p, bf16_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEVI void st_wt_u32(void* p, uint32_t v) { __hip_atomic_store(reinterpret_cast<uint32_t*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEVI void st_wt_f32(float* p, float v) { st_wt_u32(p, __float_as_uint(v)); }
DEVI void st_wt_u64(void* p, u32x2 v) {
    __hip_atomic_store(reinterpret_cast<uint64_t*>(p), __builtin_bit_cast(uint64_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
DEVI void st_wt_f32x4(float* p, f32x4 v) {
    st_wt_u64(p, u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])});
    st_wt_u64(p + 2, u32x2{__float_as_uint(v[2]), __float_as_uint(v[3])});
}
DEVI float ld_agent_f32(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Residual-stream row as the consumer sees it:  x = bf16(h[r] + sum_s slab[s][r])  (n_slabs may be 0), then
// X = rmsnorm(x) * w for rows r < B, written to LDS in fragment order; one workgroup (`write_out`) also stores x to
// h_out (the producer of the slabs — a K-split projection — leaves the residual add to its consumer; h_out != h).
// row_ids != nullptr: row r is h[row_ids[r]] (embedding gather).  wt: h_out is read by OTHER workgroups later in the
// same kernel (persistent decode step), so it is stored write-through at agent scope.  tiled: h and the slabs are in the
// persistent step's producer-owned layout [column tile of 16][16 rows][16] (every 128-B line has ONE writing workgroup).
// LDS image: [K/8][XR][8] with XR = 8 (B <= 8: lanes m >= 8 alias rows m-8, half the LDS, twice the occupancy) or 16.
// Rows >= B are left untouched: column m of the MFMA result depends only on row m of X and columns >= B are never stored.
DEVI void norm_rows_to_lds(const bf16_t* __restrict__ h, const int32_t* __restrict__ row_ids, const float* __restrict__ slabs, int n_slabs,
                           bf16_t* __restrict__ h_out, bool write_out, bool wt, const bf16_t* __restrict__ w, int B, int dim, float eps,
                           bf16_t* __restrict__ xs, int XR, int wave, int n_waves, int lane, bool tiled = false) {
    for (int r = wave; r < B; r += n_waves) {
        const bf16_t* row = h + (size_t)(row_ids ? row_ids[r] : r) * dim;
        u32x4 v[4];
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = c * 512 + lane * 8;
            if (k < dim) {
                v[c] = *reinterpret_cast<const u32x4*>(tiled ? h + ((size_t)(k >> 4) * 16 + r) * 16 + (k & 15) : row + k);
                if (n_slabs > 0) {
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { f[2 * e] = lo_bf(v[c][e]); f[2 * e + 1] = hi_bf(v[c][e]); }
                    float add[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    for (int sidx = 0; sidx < n_slabs; ++sidx) {
                        const float* sp = tiled ? slabs + (((size_t)sidx * (dim >> 4) + (k >> 4)) * 16 + r) * 16 + (k & 15)
                                                : slabs + ((size_t)sidx * 16 + r) * dim + k;
                        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { add[e] += s0[e]; add[4 + e] += s1[e]; }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[c][e] = pack_bf2(f[2 * e] + add[2 * e], f[2 * e + 1] + add[2 * e + 1]);
                }
                if (write_out) {
                    bf16_t* dst = h_out + (size_t)r * dim + k;
                    if (wt) {
                        st_wt_u64(dst, u32x2{v[c][0], v[c][1]});
                        st_wt_u64(dst + 4, u32x2{v[c][2], v[c][3]});
                    } else {
                        *reinterpret_cast<u32x4*>(dst) = v[c];
                    }
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

