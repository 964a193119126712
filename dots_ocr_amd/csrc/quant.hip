// fp8 weights (BASELINE configs[4] / SURVEY §8(d) config 5): OCP e4m3 ("e4m3fn", gfx950's native fp8) with one fp32
// scale per OUTPUT channel:  q = e4m3(W[n][k] / scale[n]),  scale[n] = max_k |W[n][k]| / 448.  Two uses of the same quantised model:
//   * DECODE (bandwidth-bound) is WEIGHT-ONLY (W8A16): the kernels stream q as 1 byte per weight in fragment order, convert it to bf16 in
//     registers (exact: every e4m3 value is a bf16 value) and multiply bf16 activations on the bf16 MFMA,
//         y[m][n] = ( sum_k x[m][k] * q[n][k] ) * scale[n]                                           (x bf16, sum in fp32);
//   * the ViT and PREFILL GEMMs (MFMA-bound) are W8A8: quant_act_fp8 below also quantises the activations per TOKEN (row scale
//     = max |x[m][:]| / 448, e4m3), gemm_fp8 (gemm.hip) runs on v_mfma_scale_f32_32x32x64_f8f6f4 and its epilogue applies
//         y[m][n] = ( sum_k xq[m][k] * q[n][k] ) * rowscale[m] * scale[n].
// So the prefill logits of a token and the decode logits of the same token are NOT the same numbers in fp8 mode: they differ by the
// activation quantisation (a step function of the activations — see DESIGN §2 for what that does to tolerances:
// tests/test_fp8_gpu.py compares each phase with the oracle mode that quantises the same things, oracle/model.py fp8_act).
#include "common.h"
#include "decode_layout.h"
#include "kernels.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// One wave per weight row: amax -> scale -> W[n][:] = bf16(q) in place (exact), scale[n].
// Rounding: v_cvt_pk_fp8_f32 (round to nearest even; |W / scale| <= 448 never saturates); tests/test_fp8_gpu.py checks it
// against torch's float8_e4m3fn cast.
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(bf16_t* __restrict__ W, float* __restrict__ scale, int64_t N, int K) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    bf16_t* row = W + n * K;
    float amax = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(row + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(lo_bf(v[e])), fabsf(hi_bf(v[e]))));
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
    if (lane == 0) scale[n] = sc;
    for (int k = lane * 8; k < K; k += 512) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(row + k);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int p = __builtin_amdgcn_cvt_pk_fp8_f32(lo_bf(v[e]) / sc, hi_bf(v[e]) / sc, 0, false);
            const f32x2 q = __builtin_amdgcn_cvt_pk_f32_fp8(p, false);
            o[e] = pack_bf2(q[0], q[1]);
        }
        *reinterpret_cast<u32x4*>(row + k) = o;
    }
}

// bf16(q) row-major [rows, K] -> e4m3 bytes in decode fragment order: chunk (tile, kstep) = 512 B, lane (g, i) holds the 8
// consecutive k of row i at byte ((i >> 3) * 4 + g) * 64 + (i & 7) * 8 — the two 8-row halves of a tile are 256 contiguous
// bytes each (two whole 128-B lines: the half-tile projections of decode_fused.hip read exactly one of them).
// rot_rows: the q / k head row permutation of launch_pack_frag_qkv.
__global__ __launch_bounds__(256) void pack_frag_fp8_kernel(const bf16_t* __restrict__ src, uint8_t* __restrict__ dst, int64_t rows, int K, int rot_rows) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one (chunk, lane) = 8 output bytes
    const int KS = K / 32;
    const int64_t tiles = (rows + 15) / 16;
    if (c >= tiles * KS * 64) return;
    const int lane = (int)(c & 63), g = lane >> 4, i = lane & 15;
    const int64_t t = c >> 6;
    const int ks = (int)(t % KS);
    const int64_t tile = t / KS;
    int64_t row = tile * 16 + i;
    if (row < rot_rows) {
        const int rr = (int)(row & 127), jj = rr >> 4, tt = rr & 15;
        row = (row & ~(int64_t)127) + 8 * jj + (tt >> 1) + 64 * (tt & 1);
    }
    u32x4 v = {0, 0, 0, 0};
    if (row < rows) v = *reinterpret_cast<const u32x4*>(src + row * K + ks * 32 + g * 8);
    u32x2 o;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int p = __builtin_amdgcn_cvt_pk_fp8_f32(lo_bf(v[2 * h]), hi_bf(v[2 * h]), 0, false);
        p = __builtin_amdgcn_cvt_pk_fp8_f32(lo_bf(v[2 * h + 1]), hi_bf(v[2 * h + 1]), p, true);
        o[h] = (uint32_t)p;
    }
    *reinterpret_cast<u32x2*>(dst + (t * 64 + fp8_lane_slot(g, i)) * 8) = o;
}

// Dynamic per-token activation quantisation for the fp8-MFMA GEMMs (gemm.hip: gemm_fp8_256pp_kernel): x bf16 [M][lda] ->
// q e4m3 [M][K] row-major + scale[m] = max|x[m][:]| / 448 (1 for a zero row): the same definition as the weights' (per row).
__global__ __launch_bounds__(256) void quant_act_fp8_kernel(const bf16_t* __restrict__ X, uint8_t* __restrict__ Q, float* __restrict__ scale,
                                                            int64_t M, int K, int lda) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const bf16_t* row = X + m * lda;
    float amax = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(row + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(lo_bf(v[e])), fabsf(hi_bf(v[e]))));
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
    if (lane == 0) scale[m] = sc;
    for (int k = lane * 8; k < K; k += 512) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(row + k);
        u32x2 o;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int p = __builtin_amdgcn_cvt_pk_fp8_f32(lo_bf(v[2 * h]) / sc, hi_bf(v[2 * h]) / sc, 0, false);
            p = __builtin_amdgcn_cvt_pk_fp8_f32(lo_bf(v[2 * h + 1]) / sc, hi_bf(v[2 * h + 1]) / sc, p, true);
            o[h] = (uint32_t)p;
        }
        *reinterpret_cast<u32x2*>(Q + m * K + k) = o;
    }
}

// bf16(q) row-major [N][K] (exact e4m3 values, launch_quant_rows_fp8) -> the e4m3 bytes, row-major: the fp8 GEMM's weight operand
__global__ __launch_bounds__(256) void bf16q_to_fp8_kernel(const bf16_t* __restrict__ src, uint8_t* __restrict__ dst, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const u32x4 v = *reinterpret_cast<const u32x4*>(src + i * 8);
    u32x2 o;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int p = __builtin_amdgcn_cvt_pk_fp8_f32(lo_bf(v[2 * h]), hi_bf(v[2 * h]), 0, false);
        p = __builtin_amdgcn_cvt_pk_fp8_f32(lo_bf(v[2 * h + 1]), hi_bf(v[2 * h + 1]), p, true);
        o[h] = (uint32_t)p;
    }
    *reinterpret_cast<u32x2*>(dst + i * 8) = o;
}

}  // namespace

hipError_t launch_quant_act_fp8(hipStream_t s, const bf16_t* X, uint8_t* Q, float* scale, int64_t M, int K, int lda) {
    if (K % 8 != 0 || lda % 8 != 0 || M < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(quant_act_fp8_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, X, Q, scale, M, K, lda);
    return hipGetLastError();
}

hipError_t launch_bf16q_to_fp8(hipStream_t s, const bf16_t* src, uint8_t* dst, int64_t n) {
    if (n % 8 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bf16q_to_fp8_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, s, src, dst, n / 8);
    return hipGetLastError();
}

hipError_t launch_quant_rows_fp8(hipStream_t s, bf16_t* W, float* scale, int64_t N, int K) {
    if (K % 8 != 0 || N < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(quant_rows_fp8_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, W, scale, N, K);
    return hipGetLastError();
}

hipError_t launch_pack_frag_fp8(hipStream_t s, const bf16_t* src, uint8_t* dst, int64_t rows, int K, int rot_rows) {
    if (K % 32 != 0) return hipErrorInvalidValue;
    const int64_t n = (rows + 15) / 16 * (K / 32) * 64;
    hipLaunchKernelGGL(pack_frag_fp8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, rows, K, rot_rows);
    return hipGetLastError();
}
