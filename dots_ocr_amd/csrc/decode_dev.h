// Device helpers of the decode dense kernels (decode_fused.hip): residual rows -> RMSNorm -> LDS X image, non-temporal weight slices,
// the streamed fragment -> MFMA operand conversion (bf16 / e4m3) and the split-K MFMA loop.
#pragma once
#include "common.h"
#include "decode_layout.h"

// NC = 16-B chunks per lane per residual row: hidden size <= 512 NC.  The rows stay in registers while the weight stream is
// in flight and the 1024-thread kernels have 128 VGPRs per lane; dots.ocr has 1536 = 512 * 3.
constexpr int NC_MAX = 3;

typedef float f32x2 __attribute__((ext_vector_type(2)));

DEVI int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }       // provably wave-uniform: scalar branches, SGPR slices

// PIN: an empty asm that consumes a loaded value.  It keeps LLVM from sinking the load into the (conditional) block of
// its first real use — i.e. from re-ordering it behind the weight stream — at the price of a wait at the pin, so pins sit
// where the value's loads have to be back anyway (in front of the norm prologue / in front of the first barrier).
#define PIN(x) asm volatile("" ::"v"(x))

// ---- residual-stream rows --------------------------------------------------------------------------------------------
// Wave `wave` owns rows wave, wave + n_waves, ... (MAXR of them); rows >= B are clamped to B - 1 for the LOADS (every load
// is unconditional: a branch around a load makes hipcc drain the whole memory queue at the join, guide §5 trap (c)) and
// skipped for the math.  X = rmsnorm(x) * w goes to LDS as the X image [K/8][XR][8].
// Rows >= B of the image are left untouched: column m of the MFMA result depends only on row m of X and columns >= B are never stored.
template <int MAXR, int NC>
struct Rows {
    u32x4 v[MAXR][NC];
    u32x4 w[NC];
};

template <int MAXR, int NC>
DEVI void rows_issue(Rows<MAXR, NC>& R, const bf16_t* __restrict__ h, const bf16_t* __restrict__ w, int B, int dim, int wave, int n_waves, int lane) {
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const int r = min(wave + i * n_waves, B - 1);
#pragma unroll
        for (int c = 0; c < NC; ++c) R.v[i][c] = *reinterpret_cast<const u32x4*>(h + (size_t)r * dim + min(c * 512 + lane * 8, dim - 8));
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) R.w[c] = *reinterpret_cast<const u32x4*>(w + min(c * 512 + lane * 8, dim - 8));
}

template <int MAXR, int NC>
DEVI void pin_rows(const Rows<MAXR, NC>& R) {
#pragma unroll
    for (int i = 0; i < MAXR; ++i)
#pragma unroll
        for (int c = 0; c < NC; ++c) PIN(R.v[i][c]);
#pragma unroll
    for (int c = 0; c < NC; ++c) PIN(R.w[c]);
}

// RMS statistic of one row held as NC 16-B chunks per lane (chunk c of lane l = elements c * 512 + 8 l .. + 7) -> 1 / rms.  ONE definition for
// every kernel that normalises a residual row (the LDS-image prologues below and the wide kernels of decode_fused.hip, which need the
// statistic only): the bits of a normalised row must not depend on the kernel that produced them.
template <int NC>
DEVI float row_rstd(const u32x4 (&v)[NC], int dim, float eps, int lane) {
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float pc = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {       // (v_dot2c_f32_bf16 was tried for the statistic: its result failed the oracle comparison by ~3 %)
            const float a = lo_bf(v[c][e]), b = hi_bf(v[c][e]);
            pc += a * a + b * b;
        }
        ss += (c * 512 + lane * 8 < dim) ? pc : 0.f;              // clamped (repeated) chunks past the row end do not count
    }
    return rsqrtf(wave_sum(ss) / dim + eps);
}

// 8 consecutive elements of a row: bf16(bf16(x * rstd) * w)   (modeling_qwen2.py:246-252: normalise in fp32, cast, then * weight)
DEVI u32x4 norm8(const u32x4 v, const u32x4 w, float rstd) {
    const f32x2 rs2 = {rstd, rstd};
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const f32x2 x = f32x2{lo_bf(v[e]), hi_bf(v[e])} * rs2;
        const uint32_t t = pack_bf2(x[0], x[1]);
        const f32x2 y = f32x2{lo_bf(t), hi_bf(t)} * f32x2{lo_bf(w[e]), hi_bf(w[e])};
        o[e] = pack_bf2(y[0], y[1]);
    }
    return o;
}

// one row -> rmsnorm -> LDS image; ~12 VALU per bf16 pair (packed fp32 multiplies, v_cvt_pk_bf16_f32 roundings)
template <int NC>
DEVI void row_norm_to_lds(const u32x4 (&v)[NC], const u32x4 (&w)[NC], int r, int dim, float eps, bf16_t* __restrict__ xs, int XR, int lane) {
    const float rstd = row_rstd<NC>(v, dim, eps, lane);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int k = c * 512 + lane * 8;
        const u32x4 o = norm8(v[c], w[c], rstd);
        if (k < dim) *reinterpret_cast<u32x4*>(xs + ((size_t)(k >> 3) * XR + r) * 8) = o;
    }
}

template <int MAXR, int NC>
DEVI void rows_norm_to_lds(const Rows<MAXR, NC>& R, int B, int dim, float eps, bf16_t* __restrict__ xs, int XR, int wave, int n_waves, int lane) {
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const int r = wave + i * n_waves;                              // wave-uniform
        if (r < B) row_norm_to_lds<NC>(R.v[i], R.w, r, dim, eps, xs, XR, lane);
    }
}

// ---- weight slices ---------------------------------------------------------------------------------------------------
// The wave's slice [k0, min(k0 + G, k1)) of one tile / half tile, 1 KiB chunk per k-step, non-temporal (each byte is read
// once).  Every load is unconditional: the slots of a slice shorter than G read a chunk of zeros instead (pointer select on a
// wave-uniform condition), so nothing has to be masked afterwards and a full slice costs no extra traffic.
static __device__ u32x4 g_zero_chunk[64];          // zero-initialised, one 1 KiB MFMA operand

template <int G, typename WT>
DEVI void weights_issue(WT (&a)[G], const WT* __restrict__ wp, int k0, int k1, int lane) {
    const WT* z = reinterpret_cast<const WT*>(g_zero_chunk) + lane;
#pragma unroll
    for (int j = 0; j < G; ++j) a[j] = __builtin_nontemporal_load(k0 + j < k1 ? wp + (size_t)(k0 + j) * 64 : z);
}

// streamed fragment -> MFMA A operand
DEVI bf16x8 as_a(bf16x8 v) { return v; }
DEVI bf16x8 as_a(u32x2 v) {          // 8 e4m3 bytes (k ascending) -> 8 bf16, exact
    const u32x4 o = {__builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v[0], 1.0f, false)),
                     __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v[0], 1.0f, true)),
                     __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v[1], 1.0f, false)),
                     __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v[1], 1.0f, true))};
    return __builtin_bit_cast(bf16x8, o);
}
template <typename WT> struct is_fp8 { static constexpr bool value = false; };
template <> struct is_fp8<u32x2> { static constexpr bool value = true; };
// this lane's element of a (tile, k-step) chunk: full 16-row tile / 8-row half tile (decode_layout.h: bf16 [g][i], fp8 [i >> 3][g][i & 7])
template <typename WT> DEVI int lane_slot(int g, int i) { return is_fp8<WT>::value ? fp8_lane_slot(g, i) : g * 16 + i; }

// acc = W-tile[k0 .. k0+G) . X with X from the LDS image (xp = this lane's B-operand base, stride in bf16x8 units, KS rows)
template <int G, typename WT>
DEVI f32x4 mfma_lds(const WT (&a)[G], const bf16x8* xp, int xstride, int k0, int KS) {
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < G; j += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_a(a[j]), xp[(size_t)min(k0 + j, KS - 1) * xstride], acc0, 0, 0, 0);
        if (j + 1 < G) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_a(a[j + 1]), xp[(size_t)min(k0 + j + 1, KS - 1) * xstride], acc1, 0, 0, 0);
    }
    return acc0 + acc1;
}

