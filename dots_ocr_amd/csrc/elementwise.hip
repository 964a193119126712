// HBM-bound row kernels of the dots.ocr hot path (SURVEY §2.3 V0,V2,V4,V8,L0,L1,L3).
// All of them move 16 B per lane per memory instruction (guide G13) and keep a row in registers.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int MAX_CHUNKS = 8;   // row length <= 8 * 512 = 4096 elements

// One wave per row; lane handles chunks c*512 + lane*8 .. +7.
// NCH > 0: dim == NCH * 512 exactly — every load is unconditional, so the row's NCH loads (and the weight's) are in flight together;
// the generic instantiation (NCH = 0) guards each chunk with a branch, and hipcc drains the memory queue at every such join
// (measured on the ViT's 1536-wide rows: 3.5 TB/s guarded).
template <bool LAYERNORM, int NCH>
__global__ __launch_bounds__(256) void norm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                   const bf16_t* __restrict__ b, bf16_t* __restrict__ y,
                                                   int64_t rows, int dim, float eps) {
    const int l = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* xr = x + row * dim;
    constexpr int NC = NCH > 0 ? NCH : MAX_CHUNKS;
    u32x4 v[NC];
    float s = 0.f, s2 = 0.f;
    if constexpr (NCH > 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) v[c] = *reinterpret_cast<const u32x4*>(xr + c * 512 + l * 8);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int off = c * 512 + l * 8;
        if (NCH > 0 || off < dim) {
            if constexpr (NCH == 0) v[c] = *reinterpret_cast<const u32x4*>(xr + off);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = lo_bf(v[c][e]), bb = hi_bf(v[c][e]);
                s += a + bb;
                s2 += a * a + bb * bb;
            }
        }
    }
    s = wave_sum(s);
    s2 = wave_sum(s2);
    float mean = 0.f, rstd;
    if (LAYERNORM) {
        mean = s / dim;
        // two-pass variance from registers (matches F.layer_norm's numerics better than E[x^2]-m^2)
        float d2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int off = c * 512 + l * 8;
            if (NCH > 0 || off < dim) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = lo_bf(v[c][e]) - mean, bb = hi_bf(v[c][e]) - mean;
                    d2 += a * a + bb * bb;
                }
            }
        }
        d2 = wave_sum(d2);
        rstd = rsqrtf(d2 / dim + eps);
    } else {
        rstd = rsqrtf(s2 / dim + eps);
    }
    bf16_t* yr = y + row * dim;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int off = c * 512 + l * 8;
        if (NCH > 0 || off < dim) {
            u32x4 ww = *reinterpret_cast<const u32x4*>(w + off);
            u32x4 bv = {0, 0, 0, 0};
            if (LAYERNORM) bv = *reinterpret_cast<const u32x4*>(b + off);
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = lo_bf(v[c][e]), bb = hi_bf(v[c][e]);
                float oa, ob;
                if (LAYERNORM) {
                    oa = (a - mean) * rstd * lo_bf(ww[e]) + lo_bf(bv[e]);
                    ob = (bb - mean) * rstd * hi_bf(ww[e]) + hi_bf(bv[e]);
                } else {
                    // modeling_qwen2.py:246-252: normalise in fp32, cast to bf16, then * weight
                    oa = bf2f(f2bf(a * rstd)) * lo_bf(ww[e]);
                    ob = bf2f(f2bf(bb * rstd)) * hi_bf(ww[e]);
                }
                o[e] = pack_bf2(oa, ob);
            }
            *reinterpret_cast<u32x4*>(yr + off) = o;
        }
    }
}

__global__ __launch_bounds__(256) void patch_prep_kernel(const float* __restrict__ x, bf16_t* __restrict__ y,
                                                         int64_t rows, int in_dim, int out_dim) {
    const int per_row = out_dim / 4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * per_row) return;
    const int64_t r = i / per_row;
    const int c = (int)(i - r * per_row) * 4;
    u32x2 o = {0, 0};
    if (c < in_dim) {   // in_dim % 4 == 0
        f32x4 v = *reinterpret_cast<const f32x4*>(x + r * in_dim + c);
        o[0] = pack_bf2(v[0], v[1]);
        o[1] = pack_bf2(v[2], v[3]);
    }
    *reinterpret_cast<u32x2*>(y + r * out_dim + c) = o;
}

// cs[t][j] = (cos, sin)(pos * inv_freq); 2-D: j<32 -> h axis, j>=32 -> w axis (VisionRotaryEmbedding(64)
// flattened over (h,w), modeling_qwen2_vl.py:239-248); 1-D: 64 frequencies of one position.
__global__ __launch_bounds__(256) void rope_table_kernel(const int32_t* __restrict__ pos, const float* __restrict__ inv_freq,
                                                         float2* __restrict__ cs, int64_t T, int two_d) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= T * 64) return;
    const int64_t t = i >> 6;
    const int j = (int)(i & 63);
    float ang;
    if (two_d) ang = (float)pos[t * 2 + (j >> 5)] * inv_freq[j & 31];
    else ang = (float)pos[t] * inv_freq[j];
    float sn, c;
    sincosf(ang, &sn, &c);
    cs[i] = make_float2(c, sn);
}

// grid (tiles, ceil((Hq + Hkv) / ROPE_HPB) + Hkv).  Slot < the number of head groups: rope the 64-token x 128-d tiles of ROPE_HPB q / k heads
// into head-major q / k — the (cos, sin) row of a token is fetched ONCE per workgroup and re-used for its heads (round 5: one workgroup per
// head read the 32-KB table tile again for every head, as many bytes as the head tile it moved; same arithmetic per element).
// Other slots: transpose a V tile into V^T [Hkv][128][Tpad] with the 16-key group order
// 0-3,8-11,4-7,12-15 (the k-index order of the PV MFMA's B operand, see attn_prefill.hip) and zero
// padding for tokens past the sequence end.
constexpr int ROPE_HPB = 4;
__global__ __launch_bounds__(256) void qkv_rope_split_kernel(const bf16_t* __restrict__ qkv, const float2* __restrict__ cs,
                                                             const Tile64* __restrict__ tiles, bf16_t* __restrict__ q,
                                                             bf16_t* __restrict__ k, bf16_t* __restrict__ vt,
                                                             int64_t T, int64_t Tpad, int Hq, int Hkv, int slot0) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[64 * 136];
    const Tile64 tl = tiles[blockIdx.x];
    const int slot = blockIdx.y + slot0;                          // slot0 > 0: the V^T slots only (q / k were written by the qkv GEMM's rope epilogue)
    const int ld = (Hq + 2 * Hkv) * 128;
    const int tid = threadIdx.x;
    const int n_groups = (Hq + Hkv + ROPE_HPB - 1) / ROPE_HPB;
    if (slot < n_groups) {
        const int h_end = min((slot + 1) * ROPE_HPB, Hq + Hkv);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = it * 256 + tid;
            const int tok = item >> 3, c = item & 7;           // 8 lanes per token, 8 d's each (+ partner d+64)
            if (tok >= tl.n) continue;
            const int64_t t = tl.tok0 + tok;
            const float2* tab = cs + t * 64 + c * 8;
            float2 cc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) cc[e] = tab[e];
            for (int hh = slot * ROPE_HPB; hh < h_end; ++hh) {
                bf16_t* dst = hh < Hq ? q + (size_t)hh * T * 128 : k + (size_t)(hh - Hq) * T * 128;
                const bf16_t* src = qkv + t * ld + hh * 128 + c * 8;
                u32x4 xl = *reinterpret_cast<const u32x4*>(src);
                u32x4 xh = *reinterpret_cast<const u32x4*>(src + 64);
                u32x4 ol, oh;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float2 c0 = cc[2 * e], c1 = cc[2 * e + 1];
                    float a0 = lo_bf(xl[e]), a1 = hi_bf(xl[e]), b0 = lo_bf(xh[e]), b1 = hi_bf(xh[e]);
                    // q*cos + rotate_half(q)*sin, rotate_half = cat(-x2, x1).  The fma contractions are written out (they are the ones hipcc chose for
                    // `a*c - b*s` / `b*c + a*s` in rounds 1-5, so no bit moved): the qkv GEMM's rope epilogue (gemm.hip: w4_epilogue_qkrope) repeats them
                    // and tests/test_kernels_gpu.py holds the two paths to the same bits
                    ol[e] = pack_bf2(__builtin_fmaf(a0, c0.x, -(b0 * c0.y)), __builtin_fmaf(a1, c1.x, -(b1 * c1.y)));
                    oh[e] = pack_bf2(__builtin_fmaf(a0, c0.y, b0 * c0.x), __builtin_fmaf(b1, c1.x, a1 * c1.y));
                }
                bf16_t* d = dst + t * 128 + c * 8;
                *reinterpret_cast<u32x4*>(d) = ol;
                *reinterpret_cast<u32x4*>(d + 64) = oh;
            }
        }
    } else {
        const int h = slot - n_groups;
        // load 64 tokens x 128 d (16 B per lane), zero rows past n
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int item = it * 256 + tid;
            const int tok = item >> 4, c = item & 15;
            u32x4 v = {0, 0, 0, 0};
            if (tok < tl.n) v = *reinterpret_cast<const u32x4*>(qkv + (size_t)(tl.tok0 + tok) * ld + (Hq + Hkv + h) * 128 + c * 8);
            // position of this key inside its 16-group: swap bits 2 and 3
            const int p = (tok & ~15) | (tok & 3) | (((tok >> 3) & 1) << 2) | (((tok >> 2) & 1) << 3);
            *reinterpret_cast<u32x4*>(&lds[p * 136 + c * 8]) = v;
        }
        __syncthreads();
        // write 128 rows (d) x 64 positions: thread -> (d, 16-position chunk)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int item = it * 256 + tid;
            const int d = item >> 3, pc = item & 7;
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t a = lds[(pc * 8 + 2 * e) * 136 + d], b = lds[(pc * 8 + 2 * e + 1) * 136 + d];
                o[e] = a | (b << 16);
            }
            *reinterpret_cast<u32x4*>(vt + ((size_t)h * 128 + d) * Tpad + tl.pad0 + pc * 8) = o;
        }
    }
}

__global__ __launch_bounds__(256) void embed_gather_kernel(const int32_t* __restrict__ src, const bf16_t* __restrict__ embed,
                                                           const bf16_t* __restrict__ vision, bf16_t* __restrict__ x,
                                                           int64_t T, int dim) {
    const int l = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const int sidx = src[t];
    const bf16_t* s = sidx >= 0 ? embed + (size_t)sidx * dim : vision + (size_t)(-sidx - 1) * dim;
    for (int off = l * 8; off < dim; off += 512)
        *reinterpret_cast<u32x4*>(x + t * dim + off) = *reinterpret_cast<const u32x4*>(s + off);
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ x, const int32_t* __restrict__ rows,
                                                          const int32_t* __restrict__ dst, bf16_t* __restrict__ y, int n, int dim) {
    const int l = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const bf16_t* s = x + (size_t)rows[r] * dim;
    bf16_t* d = y + (size_t)(dst ? dst[r] : r) * dim;
    for (int off = l * 8; off < dim; off += 512) *reinterpret_cast<u32x4*>(d + off) = *reinterpret_cast<const u32x4*>(s + off);
}

}  // namespace

hipError_t launch_rmsnorm(hipStream_t s, const bf16_t* x, const bf16_t* w, bf16_t* y, int64_t rows, int dim, float eps) {
    if (rows <= 0) return hipSuccess;
    if (dim % 8 != 0 || dim > MAX_CHUNKS * 512) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((rows + 3) / 4));
    const bf16_t* nob = nullptr;
    if (dim == 1536) hipLaunchKernelGGL((norm_kernel<false, 3>), grid, dim3(256), 0, s, x, w, nob, y, rows, dim, eps);
    else if (dim == 1024) hipLaunchKernelGGL((norm_kernel<false, 2>), grid, dim3(256), 0, s, x, w, nob, y, rows, dim, eps);
    else if (dim == 512) hipLaunchKernelGGL((norm_kernel<false, 1>), grid, dim3(256), 0, s, x, w, nob, y, rows, dim, eps);
    else hipLaunchKernelGGL((norm_kernel<false, 0>), grid, dim3(256), 0, s, x, w, nob, y, rows, dim, eps);
    return hipGetLastError();
}

hipError_t launch_layernorm(hipStream_t s, const bf16_t* x, const bf16_t* w, const bf16_t* b, bf16_t* y,
                            int64_t rows, int dim, float eps) {
    if (rows <= 0) return hipSuccess;
    if (dim % 8 != 0 || dim > MAX_CHUNKS * 512) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (dim == 1536) hipLaunchKernelGGL((norm_kernel<true, 3>), grid, dim3(256), 0, s, x, w, b, y, rows, dim, eps);
    else hipLaunchKernelGGL((norm_kernel<true, 0>), grid, dim3(256), 0, s, x, w, b, y, rows, dim, eps);
    return hipGetLastError();
}

hipError_t launch_patch_prep(hipStream_t s, const float* x, bf16_t* y, int64_t rows, int in_dim, int out_dim) {
    if (rows <= 0) return hipSuccess;
    if (in_dim % 4 != 0 || out_dim % 4 != 0 || out_dim < in_dim) return hipErrorInvalidValue;
    const int64_t n = rows * (out_dim / 4);
    hipLaunchKernelGGL(patch_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, rows, in_dim, out_dim);
    return hipGetLastError();
}

hipError_t launch_rope_table(hipStream_t s, const int32_t* pos, const float* inv_freq, float2* cs, int64_t T, int two_d) {
    if (T <= 0) return hipSuccess;
    hipLaunchKernelGGL(rope_table_kernel, dim3((unsigned)((T * 64 + 255) / 256)), dim3(256), 0, s, pos, inv_freq, cs, T, two_d);
    return hipGetLastError();
}

hipError_t launch_qkv_rope_split(hipStream_t s, const bf16_t* qkv, const float2* cs, const Tile64* tiles, int n_tiles,
                                 bf16_t* q, bf16_t* k, bf16_t* vt, int64_t T, int64_t Tpad, int Hq, int Hkv, bool v_only) {
    if (n_tiles <= 0) return hipSuccess;
    const int n_groups = (Hq + Hkv + ROPE_HPB - 1) / ROPE_HPB;
    if (v_only) hipLaunchKernelGGL(qkv_rope_split_kernel, dim3(n_tiles, Hkv), dim3(256), 0, s, qkv, cs, tiles, q, k, vt, T, Tpad, Hq, Hkv, n_groups);
    else hipLaunchKernelGGL(qkv_rope_split_kernel, dim3(n_tiles, n_groups + Hkv), dim3(256), 0, s, qkv, cs, tiles, q, k, vt, T, Tpad, Hq, Hkv, 0);
    return hipGetLastError();
}

hipError_t launch_embed_gather(hipStream_t s, const int32_t* src, const bf16_t* embed, const bf16_t* vision, bf16_t* x,
                               int64_t T, int dim) {
    if (T <= 0) return hipSuccess;
    hipLaunchKernelGGL(embed_gather_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, s, src, embed, vision, x, T, dim);
    return hipGetLastError();
}

hipError_t launch_gather_rows(hipStream_t s, const bf16_t* x, const int32_t* rows, const int32_t* dst, bf16_t* y, int n, int dim) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, s, x, rows, dst, y, n, dim);
    return hipGetLastError();
}
