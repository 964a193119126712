// bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue(A[M,K] @ W[N,K]^T + bias)
//
// Used for every dense contraction of the ViT encoder, the patch merger and the LM prefill
// (SURVEY §2.3 V1,V3,V6,V7,V9, L2,L7,L8).  MFMA-bound: 2*M*N*K flops per launch.
//
// Structure (one workgroup = 4 waves = one 128(n) x 128(m) output tile, BK = 64):
//   * both operands are K-contiguous ("B^T input"), so W rows and A rows are staged the same way:
//     HBM -> LDS by LDS-DMA (global_load_lds, 16 B/lane, 1 KiB per wave-instruction), double
//     buffered, one barrier per K-tile, next tile's DMA in flight under this tile's MFMAs;
//   * LDS image = 128 rows x 128 B, 16-B slot XOR-swizzled by (row>>1)&7.  The DMA destination
//     is lane-linear, so the swizzle is applied to the per-lane SOURCE address and again on the
//     ds_read_b128 (guide §5.4 rule 21); conflict-free for the 32x32x16 fragment gather;
//   * v_mfma_f32_32x32x16_bf16 with the WEIGHT tile as the A operand: D[n][m].  A lane then owns
//     4 consecutive n for one m, i.e. 8 contiguous bytes of row-major C -> bias/residual/activation
//     epilogues are lane-local and stores are 8-byte (16-byte for fp32 output);
//   * wave tile 64(n) x 64(m): 2x2 accumulators of 32x32 (64 VGPRs), 16 MFMAs + 16 ds_read_b128 per
//     K-tile per wave;
//   * 1-D grid, XCD-aware remap + grouped-M raster so the 32 CUs of one XCD share panels in L2.
//
// Requirements (checked by the launcher): N % 128 == 0, K % 64 == 0, 16-B aligned rows.  M is free
// (rows are clamped on load, masked on store).
#include <cstdlib>
#include <utility>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 128;            // one operand tile: 128 rows x 64 bf16
#ifndef GEMM_GROUP_M
#define GEMM_GROUP_M 8
#endif
constexpr int GROUP_M = GEMM_GROUP_M;

DEVI float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// silu(x) = x / (1 + e^-x).  The IEEE division hipcc emits for the plain expression is ~12 VALU instructions (v_div_scale x2, rcp, 4 fma,
// v_div_fmas, v_div_fixup); 128 of them per lane made the SwiGLU epilogue of the one-wave-per-SIMD kernel 17 k cycles per tile against
// 9 k for a plain one (profiles/r05_gemm_w4_cycle_stamps.txt).  Here: reciprocal + ONE Newton correction of the quotient (error < 1 ulp
// of fp32 before the single rounding to bf16; the exponent is clamped so that 1 + e^-x stays finite and 0 * inf cannot appear).
// -DGEMM_EXACT_SILU restores the division.  BOTH GEMM plans and the fp8 twin share this function, so "bit-identical" holds between plans, not
// against builds before round 5: an output may differ from the exact quotient's bf16 rounding by one bf16 ulp, and for x < -88 the result is a
// tiny negative instead of -0 (tests/test_kernels_gpu.py: test_gemm_swiglu_epilogue_is_within_one_bf16_ulp_of_the_exact_quotient).
#ifdef GEMM_EXACT_SILU
DEVI float silu(float x) { return x / (1.0f + __expf(-x)); }
#else
DEVI float silu(float x) {
    const float d = 1.0f + __expf(fminf(-x, 80.f));
    const float r = __builtin_amdgcn_rcpf(d);
    const float q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-q, d, x), r, q);
}
#endif

// ---- epilogue shared by both tile shapes: lane owns (m, 4 consecutive n) per (fn, fm, rq) of its 64x64 wave tile.
// nw0 = first weight row (n) of the wave tile, mw0 = first activation row (m) of the wave tile.
template <int EPI, int FN, int FM = 2>
DEVI void gemm_epilogue(const f32x16 (&acc)[FN][FM], const bf16_t* __restrict__ bias, const float* __restrict__ colscale, const bf16_t* R,
                        void* Cout, int M, int ldc, int nw0, int mw0, int l31, int hi, const float* __restrict__ rowscale = nullptr) {
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        const int m = mw0 + fm * 32 + l31;
        if (m >= M) continue;
        const float rs = rowscale ? rowscale[m] : 1.f;           // fp8 activations: per-token scale (quant.hip)
        if constexpr (EPI == EPI_SWIGLU) {
            bf16_t* C = reinterpret_cast<bf16_t*>(Cout);
#pragma unroll
            for (int fp = 0; fp < FN / 2; ++fp)                  // 64-row group: fragment 2fp = gate, 2fp+1 = up
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int nb = nw0 + fp * 64 + 8 * rq + 4 * hi;          // packed gate row; up = nb + 32
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float g = acc[2 * fp][fm][4 * rq + e], u = acc[2 * fp + 1][fm][4 * rq + e];
                    if (colscale) { g *= colscale[nb + e] * rs; u *= colscale[nb + 32 + e] * rs; }       // fp8 weights: per-output-channel scale (quant.hip)
                    if (bias) { g += bf2f(bias[nb + e]); u += bf2f(bias[nb + 32 + e]); }
                    o[e] = silu(g) * u;
                }
                const int j = (nw0 + fp * 64) / 2 + 8 * rq + 4 * hi;
                u32x2 pk = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
                *reinterpret_cast<u32x2*>(C + (size_t)m * ldc + j) = pk;
            }
        } else {
#pragma unroll
            for (int fn = 0; fn < FN; ++fn)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int nb = nw0 + fn * 32 + 8 * rq + 4 * hi;
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = acc[fn][fm][4 * rq + e];
                    if (colscale) {
                        const f32x4 sc = *reinterpret_cast<const f32x4*>(colscale + nb);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] *= sc[e] * rs;
                    }
                    if (bias) {
                        u32x2 bb = *reinterpret_cast<const u32x2*>(bias + nb);
                        o[0] += lo_bf(bb[0]); o[1] += hi_bf(bb[0]); o[2] += lo_bf(bb[1]); o[3] += hi_bf(bb[1]);
                    }
                    if constexpr (EPI == EPI_RESIDUAL) {
                        u32x2 rr = *reinterpret_cast<const u32x2*>(R + (size_t)m * ldc + nb);
                        o[0] += lo_bf(rr[0]); o[1] += hi_bf(rr[0]); o[2] += lo_bf(rr[1]); o[3] += hi_bf(rr[1]);
                    }
                    if constexpr (EPI == EPI_GELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = gelu_erf(o[e]);
                    }
                    if constexpr (EPI == EPI_F32) {
                        f32x4 v = {o[0], o[1], o[2], o[3]};
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(Cout) + (size_t)m * ldc + nb) = v;
                    } else {
                        u32x2 pk = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
                        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(Cout) + (size_t)m * ldc + nb) = pk;
                    }
                }
        }
    }
}

// ---- epilogue of the 256 x 256 kernels, staged through LDS.  The direct epilogue above stores 8 bytes per lane at a row stride
// (32 partial lines per store instruction; measured 17 % of the GEMM time at K = 1536: tools/gemm_bench.py, -DPP_NO_STORE).
// Here each wave transposes one 32(m) x 128(n) fp32 half of its tile through its own 16 KB of the (now idle) pipeline stages and
// writes whole lines: a lane owns 8 consecutive n of one row -> 16-byte stores, 4 rows x 256 B per instruction; residual reads
// are coalesced the same way.  Same arithmetic as gemm_epilogue (fp32: * colscale, + bias, + residual, activation, ONE rounding).
// LDS image [32 rows][32 slots of 16 B], slot ^= (row & 15) << 1: conflict-free ds_write_b128 (8 consecutive rows per group),
// and the slot pair (2k, 2k + 1) of a lane's 8 floats stays adjacent.  Private to the wave: no barrier, LDS ops of one wave are
// executed in order.
template <int EPI, int FM = 2>
DEVI void gemm_epilogue_lds(const f32x16 (&acc)[4][FM], const bf16_t* __restrict__ bias, const float* __restrict__ colscale, const bf16_t* R,
                            bf16_t* __restrict__ C, int M, int ldc, int nw0, int mw0, int l, char* stage, const float* __restrict__ rowscale = nullptr) {
    const int hi = l >> 5, l31 = l & 31;
    constexpr bool SW = EPI == EPI_SWIGLU;
    // read-side geometry: plain: 4 rows x 16 lanes (8 n each); SwiGLU: 8 rows x 8 lanes (8 outputs each = 8 gate + 8 up inputs)
    const int rrow = SW ? (l >> 3) : (l >> 4);
    const int gcol = SW ? ((l & 7) >> 2) * 64 + (l & 3) * 8 : (l & 15) * 8;          // first of the lane's 8 (gate) columns in the wave tile
    float sc0[8], sc1[8], bi0[8], bi1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc0[e] = colscale ? colscale[nw0 + gcol + e] : 1.f;
        bi0[e] = bias ? bf2f(bias[nw0 + gcol + e]) : 0.f;
        sc1[e] = (SW && colscale) ? colscale[nw0 + gcol + 32 + e] : 1.f;
        bi1[e] = (SW && bias) ? bf2f(bias[nw0 + gcol + 32 + e]) : 0.f;
    }
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
#pragma unroll
        for (int fn = 0; fn < 4; ++fn)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int slot = (fn * 8 + rq * 2 + hi) ^ ((l31 & 15) << 1);
                const f32x4 v = {acc[fn][fm][4 * rq], acc[fn][fm][4 * rq + 1], acc[fn][fm][4 * rq + 2], acc[fn][fm][4 * rq + 3]};
                *reinterpret_cast<f32x4*>(stage + l31 * 512 + slot * 16) = v;
            }
#pragma unroll
        for (int it = 0; it < (SW ? 4 : 8); ++it) {
            const int row = it * (SW ? 8 : 4) + rrow;
            const int m = mw0 + fm * 32 + row;
            const int sw = (row & 15) << 1;
            const char* rp = stage + row * 512;
            const int s0 = gcol >> 2;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(rp + ((s0 ^ sw) << 4)), a1 = *reinterpret_cast<const f32x4*>(rp + (((s0 + 1) ^ sw) << 4));
            float o[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            const float rs = rowscale ? rowscale[min(m, M - 1)] : 1.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = o[e] * (sc0[e] * rs) + bi0[e];
            if constexpr (SW) {
                const f32x4 u0 = *reinterpret_cast<const f32x4*>(rp + (((s0 + 8) ^ sw) << 4)), u1 = *reinterpret_cast<const f32x4*>(rp + (((s0 + 9) ^ sw) << 4));
                const float u[8] = {u0[0], u0[1], u0[2], u0[3], u1[0], u1[1], u1[2], u1[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = silu(o[e]) * (u[e] * (sc1[e] * rs) + bi1[e]);
            }
            if (m >= M) continue;
            const int col = SW ? nw0 / 2 + ((l & 7) >> 2) * 32 + (l & 3) * 8 : nw0 + gcol;
            if constexpr (EPI == EPI_RESIDUAL) {
                const u32x4 rr = *reinterpret_cast<const u32x4*>(R + (size_t)m * ldc + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[2 * e] += lo_bf(rr[e]); o[2 * e + 1] += hi_bf(rr[e]); }
            }
            if constexpr (EPI == EPI_GELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = gelu_erf(o[e]);
            }
            const u32x4 pk = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7])};
            *reinterpret_cast<u32x4*>(C + (size_t)m * ldc + col) = pk;
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const bf16_t* __restrict__ bias,
    const float* __restrict__ colscale, const bf16_t* R, void* Cout, int M, int N, int K, int lda, int ldc, int m_tiles, int n_tiles) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * TILE_BYTES];

    const int tid = threadIdx.x;
    const int l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = l >> 5, l31 = l & 31;

    // ---- tile raster: XCD remap, then grouped-M (GROUP_M m-tiles sweep all n-tiles) ----
    int bid = xcd_remap(blockIdx.x, m_tiles * n_tiles);
    const int per_group = GROUP_M * n_tiles;
    const int grp = bid / per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = min(m_tiles - first_m, GROUP_M);
    const int in_grp = bid - grp * per_group;
    const int tm = first_m + in_grp % gsz;
    const int tn = in_grp / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-lane DMA source pointers: piece p = i*4 + w covers rows 8p..8p+7 ----
    const bf16_t* gw[4];
    const bf16_t* gx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i * 4 + w) * 8 + (l >> 3);
        const int chunk = (l & 7) ^ ((row >> 1) & 7);
        gw[i] = W + (size_t)(n0 + row) * K + chunk * 8;
        const int mrow = min(m0 + row, M - 1);
        gx[i] = A + (size_t)mrow * lda + chunk * 8;
    }

    auto issue = [&](int kt, int buf) {
        char* base = smem + buf * 2 * TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = i * 4 + w;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(gw[i] + kt * BK),
                (__attribute__((address_space(3))) void*)(base + p * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(gx[i] + kt * BK),
                (__attribute__((address_space(3))) void*)(base + TILE_BYTES + p * 1024), 16, 0, 0);
        }
    };

    const int wn = w >> 1, wm = w & 1;
    const int rsw = (l31 >> 1) & 7;
    const int a_off = (wn * 64 + l31) * 128;                 // + fn*32*128
    const int b_off = TILE_BYTES + (wm * 64 + l31) * 128;    // + fm*32*128

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = K / BK;
    issue(0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nt) issue(t + 1, (t + 1) & 1);
        const char* base = smem + (t & 1) * 2 * TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int slot = ((ks * 2 + hi) ^ rsw) << 4;
            bf16x8 a0 = *reinterpret_cast<const bf16x8*>(base + a_off + slot);
            bf16x8 a1 = *reinterpret_cast<const bf16x8*>(base + a_off + 32 * 128 + slot);
            bf16x8 b0 = *reinterpret_cast<const bf16x8*>(base + b_off + slot);
            bf16x8 b1 = *reinterpret_cast<const bf16x8*>(base + b_off + 32 * 128 + slot);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
    }

    gemm_epilogue<EPI, 2>(acc, bias, colscale, R, Cout, M, ldc, n0 + wn * 64, m0 + wm * 64, l31, hi);
}

// ------------------------------------------------------------------------------------------------
// Large-tile variant: 256(n) x 256(m) x 64, 8 waves (2 over n x 4 over m), wave tile 128(n) x 64(m) =
// 4x2 accumulators of 32x32 (128 VGPRs), 32 MFMAs + 24 ds_read_b128 per K-tile per wave.
//
// Why: at full MFMA rate a CU retires 4069 flop/clk; a BNxBM tile needs (BN+BM)/(BN*BM) bytes of
// global->LDS traffic per flop, i.e. 64 B/clk for 128x128 — exactly the CU's L1/TA path (measured in
// round 1: 128x128 and a 3-stage 256x128 variant both sit at 30-37 % MFMA utilisation with waves parked
// 38 % of the time; profiles/r01_pmc_flash_gemm.json).  256x256 needs 32 B/clk and halves LDS-read
// traffic per flop as well.  LDS: 2 stages x 64 KB (dynamic), LDS-DMA with the same source-side XOR
// swizzle, one raw s_barrier per K-tile; the next tile's 8 DMA pieces per thread have a whole tile of
// MFMAs (2048 cycles per SIMD) to land.  Used when N % 256 == 0.
constexpr int BN2 = 256, BM2 = 256;
constexpr int OP2_BYTES = 256 * 128, STAGE2 = 2 * OP2_BYTES;

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_256_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const bf16_t* __restrict__ bias,
    const float* __restrict__ colscale, const bf16_t* R, void* Cout, int M, int N, int K, int lda, int ldc, int m_tiles, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem2[];

    const int tid = threadIdx.x;
    const int l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = l >> 5, l31 = l & 31;

    int bid = xcd_remap(blockIdx.x, m_tiles * n_tiles);
    const int per_group = GROUP_M * n_tiles;
    const int grp = bid / per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = min(m_tiles - first_m, GROUP_M);
    const int in_grp = bid - grp * per_group;
    const int tm = first_m + in_grp % gsz;
    const int tn = in_grp / gsz;
    const int m0 = tm * BM2, n0 = tn * BN2;

    // DMA pieces (8 rows x 128 B = 1 KiB): 32 per operand, wave w takes pieces i*8 + w (i < 4)
    const bf16_t* gw[4];
    const bf16_t* gx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i * 8 + w) * 8 + (l >> 3);
        const int chunk = (l & 7) ^ ((row >> 1) & 7);
        gw[i] = W + (size_t)(n0 + row) * K + chunk * 8;
        gx[i] = A + (size_t)min(m0 + row, M - 1) * lda + chunk * 8;
    }
    // one (W piece, X piece) pair of the next tile; called once per k-step so the 8 DMA issues of a tile are
    // spread between the MFMA clusters instead of stalling every wave right after the barrier
    auto issue_pair = [&](int kt, int stage, int i) {
        char* base = smem2 + stage * STAGE2;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gw[i] + kt * BK),
                                         (__attribute__((address_space(3))) void*)(base + (i * 8 + w) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gx[i] + kt * BK),
                                         (__attribute__((address_space(3))) void*)(base + OP2_BYTES + (i * 8 + w) * 1024), 16, 0, 0);
    };

    const int wn = w >> 2, wm = w & 3;                         // 2 x 4 waves
    const int rsw = (l31 >> 1) & 7;
    const int a_off = (wn * 128 + l31) * 128;                   // + fn*32*128
    const int b_off = OP2_BYTES + (wm * 64 + l31) * 128;        // + fm*32*128

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = K / BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_pair(0, 0, i);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const bool has_next = t + 1 < nt;
        const char* base = smem2 + (t & 1) * STAGE2;
        bf16x8 af[2][4], bf_[2][2];
        auto load_frags = [&](int ks, int set) {
            const int slot = ((ks * 2 + hi) ^ rsw) << 4;
            bf_[set][0] = *reinterpret_cast<const bf16x8*>(base + b_off + slot);
            bf_[set][1] = *reinterpret_cast<const bf16x8*>(base + b_off + 32 * 128 + slot);
#pragma unroll
            for (int fn = 0; fn < 4; ++fn) af[set][fn] = *reinterpret_cast<const bf16x8*>(base + a_off + fn * 32 * 128 + slot);
        };
        load_frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) load_frags(ks + 1, (ks + 1) & 1);          // fragments of the next k-step under this one's MFMAs
            if (has_next) issue_pair(t + 1, (t + 1) & 1, ks);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int fn = 0; fn < 4; ++fn) {
                acc[fn][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][fn], bf_[ks & 1][0], acc[fn][0], 0, 0, 0);
                acc[fn][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][fn], bf_[ks & 1][1], acc[fn][1], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    gemm_epilogue<EPI, 4>(acc, bias, colscale, R, Cout, M, ldc, n0 + wn * 128, m0 + wm * 64, l31, hi);
}

// ------------------------------------------------------------------------------------------------
// Ping-pong variant of the 256 x 256 tile (same wave tiles, same epilogue): K is cut into SUB-TILES of 32 (16 MFMAs per wave),
// four LDS stages of 32 KB, and the two wave groups (waves 0-3 / 4-7 = the two waves of every SIMD) run half a sub-tile period
// apart: while one group issues its 16 MFMAs the other fetches its 12 operand fragments from LDS and issues its 4 DMA pieces,
// then they swap.  Per group and sub-tile p:
//     L(p): 12 ds_read_b128 of sub-tile p | 4 DMA pieces of sub-tile p + 3 | vmcnt -> own pieces of p + 1 landed | lgkmcnt(0) | barrier
//     M(p): 16 MFMAs at raised priority                                                                                     | barrier
// and group 1 passes one extra barrier first, so its L(p) coincides with group 0's M(p).  The lock-step kernel above loses the
// MFMA pipe at every K-tile to barrier skew + the first fragments' LDS latency (48 % MFMA-busy, waves 36 % parked:
// profiles/r01_pmc_flash_gemm.json); here the pipe always has one wave per SIMD inside an MFMA cluster.
// Ordering: a sub-tile is read one full part after every wave waited for its own DMA pieces and passed a barrier; its stage is
// re-filled (sub-tile p + 3 -> stage (p - 1) & 3) only after the barrier that follows group 1's last read of sub-tile p - 1.
// LDS image of a sub-tile operand: 256 rows x 64 B, 16-B slot XOR ((row >> 2) & 3): the 16 lanes of a ds_read_b128 group
// (rows 0-3, 12-15, 20-27 / 4-11, 16-19, 28-31) land on 16 distinct slots of the 256-B bank row; applied on the DMA source
// side (piece = 16 rows x 64 B, lane l = row l >> 2, slot l & 3) and again on the read.
// Timing ablations (never defined in the product build; tools/build_variant_gemm.sh + tools/gemm_bench.py, results in DESIGN §5.1b):
// -DPP_NO_DMA / PP_NO_LDS / PP_NO_MFMA / PP_NO_STORE drop one ingredient of the loop (the results are then garbage), -DPP_DMA_SAME makes
// every DMA hit L2.  PINV keeps a value alive without using it.
#define PINV(x) asm volatile("" ::"v"(x))
constexpr int SUBK = 32, SUB_OP = 256 * 64, SUB_STAGE = 2 * SUB_OP, PP_STAGES = 4, PP_LOOKAHEAD = 3;

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_256pp_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const bf16_t* __restrict__ bias,
    const float* __restrict__ colscale, const bf16_t* R, void* Cout, int M, int N, int K, int lda, int ldc, int m_tiles, int n_tiles, int group_m) {
    extern __shared__ __attribute__((aligned(16))) char smem2[];

    const int tid = threadIdx.x;
    const int l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = l >> 5, l31 = l & 31;
    const int grp = w >> 2;                                     // waves w and w + 4 share a SIMD

    int bid = xcd_remap(blockIdx.x, m_tiles * n_tiles);
    const int per_group = group_m * n_tiles;
    const int g = bid / per_group;
    const int first_m = g * group_m;
    const int gsz = min(m_tiles - first_m, group_m);
    const int in_grp = bid - g * per_group;
    const int tm = first_m + in_grp % gsz;
    const int tn = in_grp / gsz;
    const int m0 = tm * BM2, n0 = tn * BN2;

    // DMA pieces (16 rows x 64 B = 1 KiB): 16 per operand and sub-tile, wave w takes pieces w and w + 8
    const bf16_t* gw[2];
    const bf16_t* gx[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (i * 8 + w) * 16 + (l >> 2);
        const int slot = (l & 3) ^ ((l >> 4) & 3);             // = (l & 3) ^ ((row >> 2) & 3)
        gw[i] = W + (size_t)(n0 + row) * K + slot * 8;
        gx[i] = A + (size_t)min(m0 + row, M - 1) * lda + slot * 8;
    }
    auto issue_sub = [&](int p) {
        char* base = smem2 + (p & (PP_STAGES - 1)) * SUB_STAGE;
#ifdef PP_DMA_SAME
        p &= 1;                                                 // timing experiment: every DMA hits L2
#endif
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gw[i] + p * SUBK),
                                             (__attribute__((address_space(3))) void*)(base + (i * 8 + w) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gx[i] + p * SUBK),
                                             (__attribute__((address_space(3))) void*)(base + SUB_OP + (i * 8 + w) * 1024), 16, 0, 0);
        }
    };

    const int wn = w >> 2, wm = w & 3;                          // 2 x 4 waves: group 0 = the first 128 weight rows
    const int rsw = (l31 >> 2) & 3;
    const int a_off = (wn * 128 + l31) * 64;                    // + fn * 32 * 64
    const int b_off = SUB_OP + (wm * 64 + l31) * 64;            // + fm * 32 * 64

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int ns = K / SUBK;
    // prologue: sub-tiles 0 .. 2 requested, sub-tile 0 landed
    issue_sub(0);
    if (ns > 1) issue_sub(1);
    if (ns > 2) issue_sub(2);
    if (ns > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ns > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();                 // half a period behind
    __builtin_amdgcn_sched_barrier(0);

    for (int p = 0; p < ns; ++p) {
        // ---- L(p)
        const char* base = smem2 + (p & (PP_STAGES - 1)) * SUB_STAGE;
        bf16x8 af[2][4], bf_[2][2];
#ifdef PP_NO_LDS
        if (p == 0)
#endif
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int slot = ((ks * 2 + hi) ^ rsw) << 4;
            bf_[ks][0] = *reinterpret_cast<const bf16x8*>(base + b_off + slot);
            bf_[ks][1] = *reinterpret_cast<const bf16x8*>(base + b_off + 32 * 64 + slot);
#pragma unroll
            for (int fn = 0; fn < 4; ++fn) af[ks][fn] = *reinterpret_cast<const bf16x8*>(base + a_off + fn * 32 * 64 + slot);
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef PP_NO_DMA
        if (false) {
#else
        if (p + PP_LOOKAHEAD < ns) {
#endif
            issue_sub(p + PP_LOOKAHEAD);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // own pieces of sub-tile p + 1 landed (p + 2, p + 3 may fly)
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- M(p)
        __builtin_amdgcn_s_setprio(1);
#ifdef PP_NO_MFMA
        if (p == 0)
#endif
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int fn = 0; fn < 4; ++fn) {
                acc[fn][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][fn], bf_[ks][0], acc[fn][0], 0, 0, 0);
                acc[fn][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][fn], bf_[ks][1], acc[fn][1], 0, 0, 0);
            }
#ifdef PP_NO_MFMA
        else { PINV(af[0][0]); PINV(af[1][3]); PINV(bf_[0][0]); PINV(bf_[1][1]); }
#endif
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 0 || p + 1 < ns) __builtin_amdgcn_s_barrier();          // group 1 ran one extra barrier at the start
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef PP_NO_STORE
#pragma unroll
    for (int i = 0; i < 4; ++i) { PINV(acc[i][0]); PINV(acc[i][1]); }
#else
    if constexpr (EPI == EPI_F32 || EPI == EPI_SWIGLU) {      // SwiGLU halves the output bytes: the direct epilogue measured faster there
        gemm_epilogue<EPI, 4>(acc, bias, colscale, R, Cout, M, ldc, n0 + wn * 128, m0 + wm * 64, l31, hi);
    } else {
        __syncthreads();                                        // every wave is done reading the pipeline stages (group 1 skipped its last barrier)
        gemm_epilogue_lds<EPI>(acc, bias, colscale, R, reinterpret_cast<bf16_t*>(Cout), M, ldc, n0 + wn * 128, m0 + wm * 64, l, smem2 + w * 16384);
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Round 5: ONE wave per SIMD (the structure of flash_attn64_kernel applied to the GEMM).  256(n) x 256(m) tile, 4 waves as
// 2(n) x 2(m), wave tile 128 x 128 = 4 x 4 accumulators of 32 x 32 in 256 AGPRs; every operand fragment now feeds FOUR MFMAs
// (the 8-wave kernels: two), so a K step of 16 costs 8 ds_read_b128 per 16 MFMAs per wave and the LDS pipe is a quarter busy.
// What the ping-pong kernel loses (profiles/r02_pmc_gemm.json: MFMA pipe 58 % busy; waves parked 26 %, issue-stalled 50 %) is
// two barriers per 16 MFMAs per wave group and a global->LDS path run in HALF cache lines (its sub-tile rows are 64 B: every
// 128-B line of an operand is requested twice, a kilobyte of DMA per 24-32 cycles measured).  Here:
//   * K tiles of 64: an operand row in LDS is one whole 128-B line, a DMA piece (buffer_load ... lds, 1 KiB per
//     wave-instruction) is 8 full lines; LDS image [256 rows][128 B], 16-B slot XOR (row >> 1) & 7 on the source side and on
//     the read (the lock-step kernel's image: conflict-free ds_read_b128);
//   * a ring of FIVE 32-KB units (= all 160 KB of LDS), unit 2t = W rows of K tile t, unit 2t + 1 = X rows: while tile t is
//     multiplied, tile t + 1 is resident, W(t + 2) is in flight, and the two units of tile t are re-filled (X(t + 2), W(t + 3))
//     as soon as every wave holds its last fragments of tile t — ONE barrier per 64 MFMAs, placed in front of the tile's last
//     K step (16 MFMAs), so the requests have 1.5-2.5 tile periods (3-5 k cycles) to land;
//   * the tile body is 64 `asm volatile` MFMA statements (compile-time fragment / accumulator selection); between them, in
//     program order, the 32 fragment reads of the next K step (one per gap, first half of each step), the 16 DMA pieces and the
//     barrier: ~1.5 issue slots per gap against the 4 that are free behind a 32x32x16 MFMA (profiles/r04_mfma_filler_probe.txt);
//   * the last two K tiles run copies of the body that request nothing (MODE 1: waits for everything; MODE 2: no barrier), so
//     the steady-state body carries no branches and no request ever leaves the operands' bounds.
// Timing ablations (never in the product build; tools/build_variant_gemm.sh): -DW4_NO_DMA requests nothing after the prologue,
// -DW4_NO_STORE drops the epilogue (results are then garbage).
// Ordering (RAW): a wave waits for its own pieces of tile t + 1 (vmcnt(8) leaves the 8 W(t + 2) pieces of this tile's head in
// flight) before the barrier of tile t; fragments of tile t + 1 are read behind that barrier.  (WAR): the units of tile t are
// re-filled behind the same barrier, which every wave passes only with all its reads of tile t returned (lgkmcnt(0)).
constexpr int W4_UNIT = 256 * 128, W4_RING = 5;

// ---- epilogue of the one-wave-per-SIMD kernel.  Same arithmetic, same LDS transposition and the same full-line stores as
// gemm_epilogue_lds, but scheduled for a wave that is ALONE on its SIMD: there nothing hides a round trip, and the straight
// version (write 16, then 8 x {read 2, wait, load residual, wait, store}) measured 7 us per tile without and 17-21 us with a residual
// (profiles/r05_gemm_ab_first.txt: a third of the kernel at K = 1536) — on 64 CUs as on 256, i.e. latency, not bandwidth
// (profiles/r05_gemm_epilogue_probe.txt).  Here every round trip is taken once per 32-row block: the residual rows of block
// fm + 1 and its 16 ds_write_b128 are issued BEFORE block fm is read back (two 16-KB images per wave), the 16 ds_read_b128 of a block
// are issued together, and the 8 stores of a block follow each other.
// per-lane column constants of the epilogue (scale, bias of the lane's 8 output columns; SwiGLU: of the 8 gate and the 8 up columns):
// fetched by vector loads at kernel start — the straight version's 16-32 scalarised loads, each behind its own branch and wait,
// were a dozen serial round trips per tile on their own
struct W4Cols { float sc0[8], bi0[8], sc1[8], bi1[8]; };
template <bool SW>
DEVI void w4_load_cols(W4Cols& cc, const bf16_t* __restrict__ bias, const float* __restrict__ colscale, int nw0, int l) {
    const int gcol = SW ? ((l & 7) >> 2) * 64 + (l & 3) * 8 : (l & 15) * 8;
    u32x4 b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
    f32x4 s00 = {1.f, 1.f, 1.f, 1.f}, s01 = s00, s10 = s00, s11 = s00;
    if (bias) {
        b0 = *reinterpret_cast<const u32x4*>(bias + nw0 + gcol);
        if (SW) b1 = *reinterpret_cast<const u32x4*>(bias + nw0 + gcol + 32);
    }
    if (colscale) {
        s00 = *reinterpret_cast<const f32x4*>(colscale + nw0 + gcol);
        s01 = *reinterpret_cast<const f32x4*>(colscale + nw0 + gcol + 4);
        if (SW) {
            s10 = *reinterpret_cast<const f32x4*>(colscale + nw0 + gcol + 32);
            s11 = *reinterpret_cast<const f32x4*>(colscale + nw0 + gcol + 36);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        cc.sc0[e] = s00[e]; cc.sc0[4 + e] = s01[e]; cc.sc1[e] = s10[e]; cc.sc1[4 + e] = s11[e];
        cc.bi0[2 * e] = lo_bf(b0[e]); cc.bi0[2 * e + 1] = hi_bf(b0[e]);
        cc.bi1[2 * e] = lo_bf(b1[e]); cc.bi1[2 * e + 1] = hi_bf(b1[e]);
    }
}

// FULL: every row of the wave tile is inside the matrix (all tiles but the ragged last m-tile): no per-row guards, so nothing splits the
// blocks of loads / stores
template <int EPI, bool FULL>
DEVI void w4_epilogue_lds(const f32x16 (&acc)[4][4], const W4Cols& cc, bool has_cols, const bf16_t* R,
                          bf16_t* __restrict__ C, int M, int ldc, int nw0, int mw0, int l, char* stage2) {
    const int hi = l >> 5, l31 = l & 31;
    constexpr bool SW = EPI == EPI_SWIGLU;
    constexpr int ITS = SW ? 4 : 8, RPI = SW ? 8 : 4;            // iterations per 32-row block, rows per iteration
    const int rrow = SW ? (l >> 3) : (l >> 4);
    const int gcol = SW ? ((l & 7) >> 2) * 64 + (l & 3) * 8 : (l & 15) * 8;          // first of the lane's 8 (gate) columns in the wave tile
    const int col = SW ? nw0 / 2 + ((l & 7) >> 2) * 32 + (l & 3) * 8 : nw0 + gcol;   // first output column
    u32x4 rr[2][8];                                               // residual rows of two blocks
    auto load_r = [&](int fm, u32x4 (&dst)[8]) {
        if constexpr (EPI == EPI_RESIDUAL) {
#pragma unroll
            for (int it = 0; it < ITS; ++it) {
                const int m = FULL ? mw0 + fm * 32 + it * RPI + rrow : min(mw0 + fm * 32 + it * RPI + rrow, M - 1);
                dst[it] = *reinterpret_cast<const u32x4*>(R + (size_t)m * ldc + col);
            }
        }
    };
    // Slot swizzle of the transposition image (a row = 32 slots of 16 B = TWO passes over the 64 banks, so slots s and s + 16 share banks).  Round 5's
    // f(row) = (row & 15) << 1 left every ds_read_b128 two-way conflicted: the 16 lanes one LDS cycle serves come from two (SwiGLU: four) consecutive
    // rows and asked for the same eight EVEN slot residues mod 16 in each (r05_pmc_gemm.json: 19 % of the LDS-active cycles were conflict cycles, all
    // here).  Round 6: odd rows also flip slot bit 0, rows with bit 1 set flip slot bit 3 — the rows of a lane group now cover disjoint residues
    // (even / odd, low / high) on the read side, and the eight rows of a ds_write_b128 lane group still hit eight distinct residues.  Same values
    // to the same addresses of C: bit-identical results.  -DW4_OLD_SWZ restores the round-5 swizzle (A/B).
#ifdef W4_OLD_SWZ
    auto swz = [](int row) { return (row & 15) << 1; };
#else
    auto swz = [](int row) { return ((row & 15) << 1) ^ (row & 1) ^ ((row & 2) << 2); };
#endif
    const int wslot = swz(l31);
    char* const wrow = stage2 + l31 * 512;
#define W4_PUT(FM, IMG)                                                                                                   \
    _Pragma("unroll") for (int fn = 0; fn < 4; ++fn) _Pragma("unroll") for (int rq = 0; rq < 4; ++rq) {                     \
        const f32x4 v = {acc[fn][FM][4 * rq], acc[fn][FM][4 * rq + 1], acc[fn][FM][4 * rq + 2], acc[fn][FM][4 * rq + 3]};   \
        *reinterpret_cast<f32x4*>(wrow + (IMG) * 16384 + (((fn * 8 + rq * 2 + hi) ^ wslot) << 4)) = v;                      \
    }
    load_r(0, rr[0]);
    W4_PUT(0, 0)
#pragma unroll
    for (int fm = 0; fm < 4; ++fm) {
        if (fm + 1 < 4) {
            load_r(fm + 1, rr[(fm + 1) & 1]);
            if (fm == 0) { W4_PUT(1, 1) } else if (fm == 1) { W4_PUT(2, 0) } else { W4_PUT(3, 1) }
        }
        __builtin_amdgcn_sched_barrier(0);
        const char* img = stage2 + (fm & 1) * 16384;
        f32x4 a[ITS][SW ? 4 : 2];
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            const int row = it * RPI + rrow;
            const int sw = swz(row);
            const char* rp = img + row * 512;
            const int s0 = gcol >> 2;
            a[it][0] = *reinterpret_cast<const f32x4*>(rp + ((s0 ^ sw) << 4));
            a[it][1] = *reinterpret_cast<const f32x4*>(rp + (((s0 + 1) ^ sw) << 4));
            if constexpr (SW) {
                a[it][2] = *reinterpret_cast<const f32x4*>(rp + (((s0 + 8) ^ sw) << 4));
                a[it][3] = *reinterpret_cast<const f32x4*>(rp + (((s0 + 9) ^ sw) << 4));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            const int m = mw0 + fm * 32 + it * RPI + rrow;
            float o[8] = {a[it][0][0], a[it][0][1], a[it][0][2], a[it][0][3], a[it][1][0], a[it][1][1], a[it][1][2], a[it][1][3]};
            float u[8];
            if constexpr (SW) {
#pragma unroll
                for (int e = 0; e < 8; ++e) u[e] = a[it][2 + (e >> 2)][e & 3];
            }
            if (has_cols) {                                       // wave-uniform: no bias and no column scale -> o * 1 + 0 is skipped (exact)
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = o[e] * cc.sc0[e] + cc.bi0[e];
                if constexpr (SW) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) u[e] = u[e] * cc.sc1[e] + cc.bi1[e];
                }
            }
            if constexpr (SW) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = silu(o[e]) * u[e];
            }
            if constexpr (EPI == EPI_RESIDUAL) {
                const u32x4 r4 = rr[fm & 1][it];
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[2 * e] += lo_bf(r4[e]); o[2 * e + 1] += hi_bf(r4[e]); }
            }
            if constexpr (EPI == EPI_GELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = gelu_erf(o[e]);
            }
            if (FULL || m < M) {
                const u32x4 pk = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7])};
                *reinterpret_cast<u32x4*>(C + (size_t)m * ldc + col) = pk;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef W4_PUT
}

// ---- EPI_QKROPE (round 6): the q / k heads of a fused qkv projection leave the GEMM rotated and head-major.  A wave tile is 128 rows x ONE head
// (128 columns): the transposition image already holds a token's whole head row, so a lane reads its own 8 columns AND the 8 columns 64 further
// (d < 64) or 64 back (d >= 64) — slot ^ 16: the same banks as its own slots, the same conflict-free swizzle — rounds both to bf16 exactly as the
// unfused GEMM would have stored them, and applies qkv_rope_split_kernel's arithmetic bit for bit:
//     d < 64 : x[d] cos_j - x[d + 64] sin_j        d >= 64 : x[d] cos_j + x[d - 64] sin_j        (j = d & 63; the same two fma contractions)
// (cos, sin)(t, j) come from the engine's fp32 table, 64 B per lane and row.  One wave per SIMD hides nothing, so the table rows are requested
// A QUARTER BLOCK AHEAD (8 rows: 8 x 16 B per lane; half a block ahead spilled) and a quarter block's 8 ds_read_b128 are issued together; image writes as w4_epilogue_lds.
DEVI void w4_load_cols_rope(W4Cols& cc, const bf16_t* __restrict__ bias, const float* __restrict__ colscale, int nw0, int l) {
    const int gcol = (l & 15) * 8, pcol = gcol ^ 64;
    u32x4 b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
    f32x4 s00 = {1.f, 1.f, 1.f, 1.f}, s01 = s00, s10 = s00, s11 = s00;
    if (bias) {
        b0 = *reinterpret_cast<const u32x4*>(bias + nw0 + gcol);
        b1 = *reinterpret_cast<const u32x4*>(bias + nw0 + pcol);
    }
    if (colscale) {
        s00 = *reinterpret_cast<const f32x4*>(colscale + nw0 + gcol);
        s01 = *reinterpret_cast<const f32x4*>(colscale + nw0 + gcol + 4);
        s10 = *reinterpret_cast<const f32x4*>(colscale + nw0 + pcol);
        s11 = *reinterpret_cast<const f32x4*>(colscale + nw0 + pcol + 4);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        cc.sc0[e] = s00[e]; cc.sc0[4 + e] = s01[e]; cc.sc1[e] = s10[e]; cc.sc1[4 + e] = s11[e];
        cc.bi0[2 * e] = lo_bf(b0[e]); cc.bi0[2 * e + 1] = hi_bf(b0[e]);
        cc.bi1[2 * e] = lo_bf(b1[e]); cc.bi1[2 * e + 1] = hi_bf(b1[e]);
    }
}

template <bool FULL>
DEVI void w4_epilogue_qkrope(const f32x16 (&acc)[4][4], const W4Cols& cc, bool has_cols, const float2* __restrict__ cs, bf16_t* __restrict__ dst,
                             int M, int mw0, int l, char* stage2) {
    const int hi = l >> 5, l31 = l & 31;
    const int rrow = l >> 4, gcol = (l & 15) * 8;
    const bool lo_half = (l & 8) == 0;                            // d < 64
    const float2* const tab = cs + (gcol & 63);
    auto swz = [](int row) { return ((row & 15) << 1) ^ (row & 1) ^ ((row & 2) << 2); };
    const int wslot = swz(l31);
    char* const wrow = stage2 + l31 * 512;
#define W4_PUT(FM, IMG)                                                                                                   \
    _Pragma("unroll") for (int fn = 0; fn < 4; ++fn) _Pragma("unroll") for (int rq = 0; rq < 4; ++rq) {                     \
        const f32x4 v = {acc[fn][FM][4 * rq], acc[fn][FM][4 * rq + 1], acc[fn][FM][4 * rq + 2], acc[fn][FM][4 * rq + 3]};   \
        *reinterpret_cast<f32x4*>(wrow + (IMG) * 16384 + (((fn * 8 + rq * 2 + hi) ^ wslot) << 4)) = v;                      \
    }
    f32x4 tb[2][2][4];                                            // (cos, sin) of two quarter blocks: [quarter][row iteration][4 x {c0 s0 c1 s1}]
    auto load_t = [&](int qb, f32x4 (&d)[2][4]) {                 // quarter block qb (0-15): rows mw0 + 8 qb + 4 it + rrow
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int m = FULL ? mw0 + qb * 8 + it * 4 + rrow : min(mw0 + qb * 8 + it * 4 + rrow, M - 1);
            const f32x4* tp = reinterpret_cast<const f32x4*>(tab + (size_t)m * 64);
#pragma unroll
            for (int c = 0; c < 4; ++c) d[it][c] = tp[c];
        }
    };
    load_t(0, tb[0]);
    W4_PUT(0, 0)
#pragma unroll
    for (int qb = 0; qb < 16; ++qb) {
        const int fm = qb >> 2;
        if (qb + 1 < 16) load_t(qb + 1, tb[(qb + 1) & 1]);
        if ((qb & 3) == 0 && fm + 1 < 4) {
            if (fm == 0) { W4_PUT(1, 1) } else if (fm == 1) { W4_PUT(2, 0) } else { W4_PUT(3, 1) }
        }
        __builtin_amdgcn_sched_barrier(0);
        const char* img = stage2 + (fm & 1) * 16384;
        f32x4 a[2][4];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = (qb & 3) * 8 + it * 4 + rrow;
            const int sw = swz(row);
            const char* rp = img + row * 512;
            const int s0 = gcol >> 2;
            a[it][0] = *reinterpret_cast<const f32x4*>(rp + ((s0 ^ sw) << 4));
            a[it][1] = *reinterpret_cast<const f32x4*>(rp + (((s0 + 1) ^ sw) << 4));
            a[it][2] = *reinterpret_cast<const f32x4*>(rp + ((s0 ^ 16 ^ sw) << 4));
            a[it][3] = *reinterpret_cast<const f32x4*>(rp + (((s0 + 1) ^ 16 ^ sw) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int m = mw0 + qb * 8 + it * 4 + rrow;
            float o[8] = {a[it][0][0], a[it][0][1], a[it][0][2], a[it][0][3], a[it][1][0], a[it][1][1], a[it][1][2], a[it][1][3]};
            float u[8] = {a[it][2][0], a[it][2][1], a[it][2][2], a[it][2][3], a[it][3][0], a[it][3][1], a[it][3][2], a[it][3][3]};
            if (has_cols) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { o[e] = o[e] * cc.sc0[e] + cc.bi0[e]; u[e] = u[e] * cc.sc1[e] + cc.bi1[e]; }
            }
            uint32_t pk[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // what the unfused GEMM stores and the split kernel reads back: ONE rounding to bf16 before the rotation
                const uint32_t ow = pack_bf2(o[2 * e], o[2 * e + 1]), uw = pack_bf2(u[2 * e], u[2 * e + 1]);
                const float x0 = lo_bf(ow), x1 = hi_bf(ow), y0 = lo_bf(uw), y1 = hi_bf(uw);
                const f32x4 t4 = tb[qb & 1][it][e];               // {cos j, sin j, cos j+1, sin j+1}, j = (gcol & 63) + 2 e
                // qkv_rope_split_kernel's contractions (elementwise.hip writes them out):  d < 64: fma(own, cos, -(partner sin));  d >= 64, even d:
                // fma(partner, sin, own cos), odd d: fma(own, cos, partner sin).  tests/test_kernels_gpu.py holds 25 M elements to the split kernel's bits.
                const float f0 = lo_half ? x0 : y0, g0 = lo_half ? t4[0] : t4[1], m0 = lo_half ? y0 : x0, n0_ = lo_half ? t4[1] : t4[0];
                const float p0 = m0 * n0_, p1 = y1 * t4[3];
                const float r0 = __builtin_fmaf(f0, g0, lo_half ? -p0 : p0), r1 = __builtin_fmaf(x1, t4[2], lo_half ? -p1 : p1);
                pk[e] = pack_bf2(r0, r1);
            }
            if (FULL || m < M) *reinterpret_cast<u32x4*>(dst + (size_t)m * 128 + gcol) = u32x4{pk[0], pk[1], pk[2], pk[3]};
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef W4_PUT
}

template <int... I, class F> DEVI void gemm_static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> DEVI void gemm_static_for(F&& f) { gemm_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

template <int EPI>
DEVI void gemm_bf16_w4_body(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const bf16_t* __restrict__ bias,
    const float* __restrict__ colscale, const bf16_t* R, void* Cout, int M, int N, int K, int lda, int ldc, int m_tiles, int n_tiles, int group_m, const QkRope& rope) {
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    (void)rope;

    const int tid = threadIdx.x;
    const int l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = l >> 5, l31 = l & 31;

    int bid = xcd_remap(blockIdx.x, m_tiles * n_tiles);
    const int per_group = group_m * n_tiles;
    const int g = bid / per_group;
    const int first_m = g * group_m;
    const int gsz = min(m_tiles - first_m, group_m);
    const int in_grp = bid - g * per_group;
    const int tm = first_m + in_grp % gsz;
    const int tn = in_grp / gsz;
    const int m0 = tm * BM2, n0 = tn * BN2;

    // ---- LDS-DMA: piece p (0-31 per unit) = rows 8p .. 8p+7 x 128 B; wave w copies pieces w + 4i (i < 8): rows advance by 32 per i,
    // which leaves the swizzle term (row >> 1) & 7 unchanged -> one per-lane source offset for W (the row step enters through the
    // scalar offset); the X rows are clamped to M - 1 per piece (8 per-lane offsets), so no request leaves the matrix
    const int prow = 8 * w + (l >> 3);
    const int chunk = ((l & 7) ^ ((4 * w + (l >> 4)) & 7)) << 4;
    const int w_src = prow * (K * 2) + chunk;
    int x_src[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x_src[i] = (min(m0 + prow + 32 * i, M - 1) - m0) * (lda * 2) + chunk;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)n0 * K), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * lda), 0, 0x7ffffff0, 0x00020000);
    const int w_step = 32 * K * 2;                               // bytes between W rows 32 apart
    auto dma_w = [&](int t, int slot, int i) {                   // piece i of W(t) -> ring slot
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(smem2 + slot * W4_UNIT + (w + 4 * i) * 1024), 16, w_src,
                                                 t * 128 + i * w_step, 0, 0);
    };
    auto dma_x = [&](int t, int slot, int i) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(smem2 + slot * W4_UNIT + (w + 4 * i) * 1024), 16, x_src[i],
                                                 t * 128, 0, 0);
    };

    const int wn = w >> 1, wm = w & 1;                           // 2 x 2 waves
    // fragment gather: row r, K step ks -> slot ((2 ks + hi) ^ ((r >> 1) & 7)) << 4 = lane constant ^ (ks << 5); r = wave base + 32 f + l31
    const int rsw = (l31 >> 1) & 7;
    const int lane_sw = ((hi ^ (rsw & 1)) << 4) | ((rsw >> 1) << 5);
    const int lds0 = (int)(uintptr_t)(__attribute__((address_space(3))) char*)smem2;     // dynamic segment: the kernel has no static LDS
    const int a_lane = lds0 + (wn * 128 + l31) * 128 + lane_sw;
    const int b_lane = lds0 + (wm * 128 + l31) * 128 + lane_sw;
    auto frag = [&](int base, int ks, int f) {
        return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>((uintptr_t)(uint32_t)((base ^ (ks << 5)) + f * 4096));
    };

#ifdef W4_PROF
    const unsigned long long pt0 = __builtin_amdgcn_s_memtime(), pw0 = wall_clock64();
#endif
    W4Cols cols;                                                 // the epilogue's column constants: their latency hides behind the main loop
    if constexpr (EPI == EPI_QKROPE) w4_load_cols_rope(cols, bias, colscale, n0 + wn * 128, l);
    else if constexpr (EPI != EPI_F32) w4_load_cols<EPI == EPI_SWIGLU>(cols, bias, colscale, n0 + wn * 128, l);

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = K / BK;                                       // >= 3 (launcher)
#ifdef W4_STAGGER
    // experiment: the first round's workgroups start in 8 phases an eighth of a tile period apart, so that the CUs do not reach their
    // epilogues (128 KB of stores each) all at once
    if (blockIdx.x < 256) {
        const int naps = ((blockIdx.x >> 3) & 7) * nt * W4_STAGGER / 8;
        for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(8);
    }
#endif
    // prologue: W(0) X(0) W(1) X(1) -> units 0-3
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (u & 1) dma_x(u >> 1, u, i);
            else dma_w(u >> 1, u, i);
        }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");           // own pieces of tile 0
    __builtin_amdgcn_s_barrier();
    bf16x8 fa[2][4], fb[2][4];                                   // fragments of the current / the next K step
#pragma unroll
    for (int f = 0; f < 4; ++f) { fa[0][f] = frag(a_lane, 0, f); fb[0][f] = frag(b_lane + W4_UNIT, 0, f); }
#ifdef W4_PROF
    const unsigned long long pt1 = __builtin_amdgcn_s_memtime(), pw1 = wall_clock64();
#endif

    // one K tile.  u0 = ring slot of W(t) (X(t): u0 + 1, W(t+1): u0 + 2, X(t+1): u0 + 3, free: u0 + 4; all mod 5)
    auto tile_body = [&](auto mode_c, int t, int u0) {
        constexpr int MODE = decltype(mode_c)::value;
        const int s1 = u0 + 1 >= W4_RING ? u0 + 1 - W4_RING : u0 + 1, s2 = u0 + 2 >= W4_RING ? u0 + 2 - W4_RING : u0 + 2;
        const int s3 = u0 + 3 >= W4_RING ? u0 + 3 - W4_RING : u0 + 3, s4 = u0 + 4 >= W4_RING ? u0 + 4 - W4_RING : u0 + 4;
        const int aw = a_lane + u0 * W4_UNIT, ax = b_lane + s1 * W4_UNIT;          // this tile (K steps 1-3)
        const int awn = a_lane + s2 * W4_UNIT, axn = b_lane + s3 * W4_UNIT;        // next tile (K step 0)
        gemm_static_for<64>([&acc, &fa, &fb, &frag, &dma_w, &dma_x, aw, ax, awn, axn, t, u0, s4](auto gc) {
            constexpr int gp = decltype(gc)::value;
            (void)acc; (void)t; (void)u0; (void)s4; (void)dma_w; (void)dma_x;           // acc: asm operand only; the rest: MODE 1 / 2 request nothing
            constexpr int ks = gp >> 4, idx = gp & 15, fn = idx >> 2, fm = idx & 3, cur = ks & 1;
            if constexpr (gp == 48 && MODE < 2) {
                // the tile's barrier: own pieces of tile t + 1 landed (MODE 0: the 8 pieces of W(t + 2) requested in this tile may fly),
                // every read of tile t returned
                if constexpr (MODE == 0) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if constexpr (idx < 8 && (ks < 3 || MODE < 2)) {      // one fragment of the next K step per gap
                constexpr int f = idx & 3;
                if constexpr (ks < 3) {
                    if constexpr (idx < 4) fa[cur ^ 1][f] = frag(aw, ks + 1, f);
                    else fb[cur ^ 1][f] = frag(ax, ks + 1, f);
                } else {
                    if constexpr (idx < 4) fa[cur ^ 1][f] = frag(awn, 0, f);
                    else fb[cur ^ 1][f] = frag(axn, 0, f);
                }
            }
#ifndef W4_NO_DMA
            if constexpr (MODE == 0) {
                // head: W(t + 2) -> the free unit, one piece per three gaps outside the fragment gaps; tail: X(t + 2) -> W(t)'s unit
                if constexpr (gp == 8 || gp == 11 || gp == 14 || gp == 24 || gp == 27 || gp == 30 || gp == 40 || gp == 43) {
                    constexpr int i = gp < 16 ? (gp - 8) / 3 : gp < 32 ? 3 + (gp - 24) / 3 : 6 + (gp - 40) / 3;
                    dma_w(t + 2, s4, i);
                }
                if constexpr (gp >= 56) dma_x(t + 2, u0, gp - 56);
            }
#endif
            asm volatile("v_mfma_f32_32x32x16_bf16 %[d], %[a], %[b], %[d]" : [d] "+a"(acc[fn][fm]) : [a] "v"(fa[cur][fn]), [b] "v"(fb[cur][fm]));
        });
    };
    int u0 = 0;
    for (int t = 0; t < nt - 2; ++t) {
        tile_body(std::integral_constant<int, 0>{}, t, u0);
        u0 = u0 + 2 >= W4_RING ? u0 + 2 - W4_RING : u0 + 2;
    }
    tile_body(std::integral_constant<int, 1>{}, nt - 2, u0);
    u0 = u0 + 2 >= W4_RING ? u0 + 2 - W4_RING : u0 + 2;
    tile_body(std::integral_constant<int, 2>{}, nt - 1, u0);

#ifdef W4_PROF
    const unsigned long long pt2 = __builtin_amdgcn_s_memtime(), pw2 = wall_clock64();
#endif
    __syncthreads();                                              // every wave is done with the ring (nothing is in flight: MODE 1 drained it)
#ifdef W4_NO_STORE
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" ::"a"(acc[i][j]));
    return;
#endif
#ifdef W4_OLD_EPILOGUE
    if constexpr (EPI == EPI_F32 || EPI == EPI_SWIGLU) {
        gemm_epilogue<EPI, 4, 4>(acc, bias, colscale, R, Cout, M, ldc, n0 + wn * 128, m0 + wm * 128, l31, hi);
    } else {
        gemm_epilogue_lds<EPI, 4>(acc, bias, colscale, R, reinterpret_cast<bf16_t*>(Cout), M, ldc, n0 + wn * 128, m0 + wm * 128, l, smem2 + w * 16384);
    }
#else
    if constexpr (EPI == EPI_QKROPE) {
        const int head = (n0 + wn * 128) >> 7;                    // wave-uniform: a wave tile is one head
        const bool full = m0 + wm * 128 + 128 <= M, has_cols = bias != nullptr || colscale != nullptr;
        if (head < rope.n_rope_heads) {
            bf16_t* dst = head < rope.Hq ? rope.q + (size_t)head * rope.T * 128 : rope.k + (size_t)(head - rope.Hq) * rope.T * 128;
            if (full) w4_epilogue_qkrope<true>(acc, cols, has_cols, rope.cs, dst, M, m0 + wm * 128, l, smem2 + w * 32768);
            else w4_epilogue_qkrope<false>(acc, cols, has_cols, rope.cs, dst, M, m0 + wm * 128, l, smem2 + w * 32768);
        } else if (full) {                                        // the v heads: stored as EPI_NONE stores them
            w4_epilogue_lds<EPI_NONE, true>(acc, cols, has_cols, R, reinterpret_cast<bf16_t*>(Cout), M, ldc, n0 + wn * 128, m0 + wm * 128, l, smem2 + w * 32768);
        } else {
            w4_epilogue_lds<EPI_NONE, false>(acc, cols, has_cols, R, reinterpret_cast<bf16_t*>(Cout), M, ldc, n0 + wn * 128, m0 + wm * 128, l, smem2 + w * 32768);
        }
    } else if constexpr (EPI == EPI_F32) {
        gemm_epilogue<EPI, 4, 4>(acc, bias, colscale, R, Cout, M, ldc, n0 + wn * 128, m0 + wm * 128, l31, hi);
    } else if (m0 + wm * 128 + 128 <= M) {                       // wave-uniform
        w4_epilogue_lds<EPI, true>(acc, cols, bias != nullptr || colscale != nullptr, R, reinterpret_cast<bf16_t*>(Cout), M, ldc, n0 + wn * 128, m0 + wm * 128, l, smem2 + w * 32768);
    } else {
        w4_epilogue_lds<EPI, false>(acc, cols, bias != nullptr || colscale != nullptr, R, reinterpret_cast<bf16_t*>(Cout), M, ldc, n0 + wn * 128, m0 + wm * 128, l, smem2 + w * 32768);
    }
#endif
#ifdef W4_PROF
    const unsigned long long pt2b = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long pt3 = __builtin_amdgcn_s_memtime(), pw3 = wall_clock64();
    if ((blockIdx.x == 1000 || blockIdx.x == 2001 || blockIdx.x == 3002) && tid == 0)
        printf("W4_PROF epi %d N %d K %d blk %d: prologue %llu cyc / %.2f us, main loop %llu cyc / %.2f us (%.0f cyc per K tile), epilogue + drain %llu cyc / %.2f us; issue %llu drain %llu\n", EPI, N, K,
               (int)blockIdx.x, pt1 - pt0, (pw1 - pw0) * 0.01, pt2 - pt1, (pw2 - pw1) * 0.01, (double)(pt2 - pt1) / nt, pt3 - pt2, (pw3 - pw2) * 0.01, pt2b - pt2, pt3 - pt2b);
#endif
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_w4_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const bf16_t* __restrict__ bias,
    const float* __restrict__ colscale, const bf16_t* R, void* Cout, int M, int N, int K, int lda, int ldc, int m_tiles, int n_tiles, int group_m) {
    gemm_bf16_w4_body<EPI>(A, W, bias, colscale, R, Cout, M, N, K, lda, ldc, m_tiles, n_tiles, group_m, QkRope{});
}
// the same kernel with the rope epilogue on its q / k head tiles (launch_gemm_qk_rope)
__global__ __launch_bounds__(256) void gemm_bf16_w4_qkrope_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const bf16_t* __restrict__ bias,
    const float* __restrict__ colscale, void* Cout, int M, int N, int K, int lda, int ldc, int m_tiles, int n_tiles, int group_m, QkRope rope) {
    gemm_bf16_w4_body<EPI_QKROPE>(A, W, bias, colscale, nullptr, Cout, M, N, K, lda, ldc, m_tiles, n_tiles, group_m, rope);
}

// ------------------------------------------------------------------------------------------------
// fp8 (e4m3 x e4m3 -> fp32) twin of the ping-pong kernel for the fp8 configuration's ViT / prefill GEMMs (DotsConfig.fp8_weights):
//     C[m][n] = epilogue( (sum_k Aq[m][k] Wq[n][k]) * rowscale[m] * colscale[n] + bias[n] )
// Aq = per-token quantised activations (quant.hip: quant_act_fp8), Wq = per-output-channel quantised weights, both row-major bytes.
// Same byte geometry as the bf16 kernel — a sub-tile row is 64 B, now 64 k instead of 32 — so staging, swizzle, barriers and the
// epilogues are shared; the 16 bf16 MFMAs of a sub-tile become 8 v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (the only
// K = 64 low-precision MFMA of gfx950; its operand pairing is proven by tests/test_mfma_layout.py): twice the flops per staged byte
// and per MFMA cycle.
typedef __attribute__((ext_vector_type(8))) int i32x8;

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_fp8_256pp_kernel(
    const uint8_t* __restrict__ A, const float* __restrict__ rowscale, const uint8_t* __restrict__ W, const float* __restrict__ colscale,
    const bf16_t* __restrict__ bias, const bf16_t* R, void* Cout, int M, int N, int K, int ldc, int m_tiles, int n_tiles, int group_m) {
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    constexpr int SUBK8 = 64;                                   // k per sub-tile (bytes per staged row)

    const int tid = threadIdx.x;
    const int l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = l >> 5, l31 = l & 31;
    const int grp = w >> 2;

    int bid = xcd_remap(blockIdx.x, m_tiles * n_tiles);
    const int per_group = group_m * n_tiles;
    const int g = bid / per_group;
    const int first_m = g * group_m;
    const int gsz = min(m_tiles - first_m, group_m);
    const int in_grp = bid - g * per_group;
    const int tm = first_m + in_grp % gsz;
    const int tn = in_grp / gsz;
    const int m0 = tm * BM2, n0 = tn * BN2;

    const uint8_t* gw[2];
    const uint8_t* gx[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (i * 8 + w) * 16 + (l >> 2);
        const int slot = (l & 3) ^ ((l >> 4) & 3);
        gw[i] = W + (size_t)(n0 + row) * K + slot * 16;
        gx[i] = A + (size_t)min(m0 + row, M - 1) * K + slot * 16;
    }
    auto issue_sub = [&](int p) {
        char* base = smem2 + (p & (PP_STAGES - 1)) * SUB_STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gw[i] + p * SUBK8),
                                             (__attribute__((address_space(3))) void*)(base + (i * 8 + w) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gx[i] + p * SUBK8),
                                             (__attribute__((address_space(3))) void*)(base + SUB_OP + (i * 8 + w) * 1024), 16, 0, 0);
        }
    };

    const int wn = w >> 2, wm = w & 3;
    const int rsw = (l31 >> 2) & 3;
    const int a_off = (wn * 128 + l31) * 64;
    const int b_off = SUB_OP + (wm * 64 + l31) * 64;
    const int s_lo = ((2 * hi) ^ rsw) << 4, s_hi = ((2 * hi + 1) ^ rsw) << 4;      // this lane's 32 bytes = logical slots 2hi, 2hi + 1

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int ns = K / SUBK8;
    issue_sub(0);
    if (ns > 1) issue_sub(1);
    if (ns > 2) issue_sub(2);
    if (ns > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ns > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    auto frag = [&](const char* rowp) {
        const u32x4 lo = *reinterpret_cast<const u32x4*>(rowp + s_lo), hi4 = *reinterpret_cast<const u32x4*>(rowp + s_hi);
        const i32x8 v = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi4[0], (int)hi4[1], (int)hi4[2], (int)hi4[3]};
        return v;
    };
    for (int p = 0; p < ns; ++p) {
        const char* base = smem2 + (p & (PP_STAGES - 1)) * SUB_STAGE;
        i32x8 af[4], bf_[2];
        bf_[0] = frag(base + b_off);
        bf_[1] = frag(base + b_off + 32 * 64);
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) af[fn] = frag(base + a_off + fn * 32 * 64);
        __builtin_amdgcn_sched_barrier(0);
        if (p + PP_LOOKAHEAD < ns) {
            issue_sub(p + PP_LOOKAHEAD);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) {
            acc[fn][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[fn], bf_[0], acc[fn][0], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            acc[fn][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[fn], bf_[1], acc[fn][1], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 0 || p + 1 < ns) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (EPI == EPI_SWIGLU) {
        gemm_epilogue<EPI, 4>(acc, bias, colscale, R, Cout, M, ldc, n0 + wn * 128, m0 + wm * 64, l31, hi, rowscale);
    } else {
        __syncthreads();
        gemm_epilogue_lds<EPI>(acc, bias, colscale, R, reinterpret_cast<bf16_t*>(Cout), M, ldc, n0 + wn * 128, m0 + wm * 64, l, smem2 + w * 16384, rowscale);
    }
}

}  // namespace

// m-tiles per raster group of the 256-wide kernels: the group's activation panels (256 rows x K) should stay in the XCD's 4 MB L2
// while its 32 CUs sweep the n-tiles.  Measured on the bench's shapes (tools/gemm_bench.py, GEMM_GROUP_M 2 / 4 / 8 / 16 / 32):
// 768 KB panels (K = 1536 bf16) 8, 2.1 MB panels (K = 4224) 4 (-8 %), 4.5 MB panels (K = 8960) 2 (-4 %); 32 is 12 % slower everywhere.
static int raster_group_m(int row_bytes) {
    const int panel = 256 * row_bytes;
    return panel >= (4 << 20) ? 2 : panel >= (3 << 19) ? 4 : GROUP_M;
}

// launch plan of the 256-wide bf16 GEMM (process-wide; results are bit-identical under either plan: the same MFMAs in the same k order
// per output element): 0 = 8 waves, ping-pong halves (round 2), 1 = 4 waves, one per SIMD, K tiles of 64 through a 5-unit ring (round 5)
static int g_gemm_plan = -1;
int gemm_get_plan() {
    int p = __atomic_load_n(&g_gemm_plan, __ATOMIC_RELAXED);
    if (p < 0) {
        const char* ev = getenv("DOTS_OCR_GEMM_PLAN");
        p = ev ? (atoi(ev) != 0) : 1;      // round 5 default: one wave per SIMD (+6-11 % on the A4 shapes, profiles/r05_gemm_*.txt)
        __atomic_store_n(&g_gemm_plan, p, __ATOMIC_RELAXED);
    }
    return p;
}
void gemm_set_plan(int plan) { __atomic_store_n(&g_gemm_plan, plan != 0, __ATOMIC_RELAXED); }

template <int E>
static hipError_t launch_256(hipStream_t s, const bf16_t* A, const bf16_t* W, const bf16_t* bias, const float* colscale, const bf16_t* R, void* C,
                                int M, int N, int K, int lda, int ldc) {
    // dynamic LDS > 64 KB is opted into once per (instantiation, DEVICE): engines on several GPUs may live in one process
    static uint32_t configured = 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint32_t bit = 1u << (dev & 31);
    if (!(__atomic_load_n(&configured, __ATOMIC_ACQUIRE) & bit)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_256_kernel<E>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE2);
        if (e != hipSuccess) return e;
        __atomic_fetch_or(&configured, bit, __ATOMIC_RELEASE);
    }
    const int m_tiles = (M + BM2 - 1) / BM2, n_tiles = N / BN2;
    static const bool lockstep = getenv("DOTS_OCR_GEMM_LOCKSTEP") != nullptr;       // A/B switch: the one-barrier-per-K-tile schedule
    if (lockstep || ldc % 8 != 0) {                 // the LDS-staged epilogue stores 16 bytes per lane
        hipLaunchKernelGGL(gemm_bf16_256_kernel<E>, dim3(m_tiles * n_tiles), dim3(512), 2 * STAGE2, s, A, W, bias, colscale, R, C, M, N, K,
                           lda, ldc, m_tiles, n_tiles);
        return hipGetLastError();
    }
    if (gemm_get_plan() == 1 && K / BK >= 3) {                  // round 5: one wave per SIMD (dots_set_gemm_plan)
        static uint32_t configured_w4 = 0;
        if (!(__atomic_load_n(&configured_w4, __ATOMIC_ACQUIRE) & bit)) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_w4_kernel<E>), hipFuncAttributeMaxDynamicSharedMemorySize, W4_RING * W4_UNIT);
            if (e != hipSuccess) return e;
            __atomic_fetch_or(&configured_w4, bit, __ATOMIC_RELEASE);
        }
        hipLaunchKernelGGL(gemm_bf16_w4_kernel<E>, dim3(m_tiles * n_tiles), dim3(256), W4_RING * W4_UNIT, s, A, W, bias, colscale, R, C, M, N, K,
                           lda, ldc, m_tiles, n_tiles, raster_group_m(K * 2));
        return hipGetLastError();
    }
    static uint32_t configured_pp = 0;
    if (!(__atomic_load_n(&configured_pp, __ATOMIC_ACQUIRE) & bit)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_256pp_kernel<E>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                PP_STAGES * SUB_STAGE);
        if (e != hipSuccess) return e;
        __atomic_fetch_or(&configured_pp, bit, __ATOMIC_RELEASE);
    }
    hipLaunchKernelGGL(gemm_bf16_256pp_kernel<E>, dim3(m_tiles * n_tiles), dim3(512), PP_STAGES * SUB_STAGE, s, A, W, bias, colscale, R, C, M, N, K,
                       lda, ldc, m_tiles, n_tiles, raster_group_m(K * 2));
    return hipGetLastError();
}

hipError_t launch_gemm(hipStream_t s, const bf16_t* A, const bf16_t* W, const bf16_t* bias, const bf16_t* R,
                       void* C, int64_t M, int N, int K, int lda, int ldc, int epi, const float* colscale) {
    if (M <= 0) return hipSuccess;
    if (N % BN != 0 || K % BK != 0 || (lda % 8) != 0 || (ldc % 4) != 0) return hipErrorInvalidValue;
    static const bool force_small = getenv("DOTS_OCR_GEMM_128") != nullptr;
    if (N % BN2 == 0 && !force_small) {
        switch (epi) {
            case EPI_NONE: return launch_256<EPI_NONE>(s, A, W, bias, colscale, R, C, (int)M, N, K, lda, ldc);
            case EPI_RESIDUAL: return launch_256<EPI_RESIDUAL>(s, A, W, bias, colscale, R, C, (int)M, N, K, lda, ldc);
            case EPI_SWIGLU: return launch_256<EPI_SWIGLU>(s, A, W, bias, colscale, R, C, (int)M, N, K, lda, ldc);
            case EPI_GELU: return launch_256<EPI_GELU>(s, A, W, bias, colscale, R, C, (int)M, N, K, lda, ldc);
            case EPI_F32: return launch_256<EPI_F32>(s, A, W, bias, colscale, R, C, (int)M, N, K, lda, ldc);
            default: return hipErrorInvalidValue;
        }
    }
    const int m_tiles = (int)((M + BM - 1) / BM), n_tiles = N / BN;
    dim3 grid(m_tiles * n_tiles), block(256);
#define LAUNCH(E)                                                                                       \
    hipLaunchKernelGGL(gemm_bf16_kernel<E>, grid, block, 0, s, A, W, bias, colscale, R, C, (int)M, N, K, lda, ldc, \
                       m_tiles, n_tiles)
    switch (epi) {
        case EPI_NONE: LAUNCH(EPI_NONE); break;
        case EPI_RESIDUAL: LAUNCH(EPI_RESIDUAL); break;
        case EPI_SWIGLU: LAUNCH(EPI_SWIGLU); break;
        case EPI_GELU: LAUNCH(EPI_GELU); break;
        case EPI_F32: LAUNCH(EPI_F32); break;
        default: return hipErrorInvalidValue;
    }
#undef LAUNCH
    return hipGetLastError();
}

hipError_t launch_gemm_qk_rope(hipStream_t s, const bf16_t* A, const bf16_t* W, const bf16_t* bias, bf16_t* qkv, int64_t M, int N, int K, int lda, int ldc,
                               const float2* cs, bf16_t* q, bf16_t* k, int Hq, int Hkv) {
    static const bool off = getenv("DOTS_OCR_QK_FUSE") && atoi(getenv("DOTS_OCR_QK_FUSE")) == 0;       // A/B switch
    static const bool lockstep = getenv("DOTS_OCR_GEMM_LOCKSTEP") != nullptr, force_small = getenv("DOTS_OCR_GEMM_128") != nullptr;
    if (off || lockstep || force_small || gemm_get_plan() != 1 || N != (Hq + 2 * Hkv) * 128 || N % BN2 != 0 || K % BK != 0 || K / BK < 3 || lda % 8 != 0 || ldc % 8 != 0 ||
        !cs || !q || !k || M > 0x7fffffff)
        return hipErrorNotSupported;
    if (M <= 0) return hipSuccess;
    static uint32_t configured = 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint32_t bit = 1u << (dev & 31);
    if (!(__atomic_load_n(&configured, __ATOMIC_ACQUIRE) & bit)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_w4_qkrope_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W4_RING * W4_UNIT);
        if (e != hipSuccess) return e;
        __atomic_fetch_or(&configured, bit, __ATOMIC_RELEASE);
    }
    const int m_tiles = (int)((M + BM2 - 1) / BM2), n_tiles = N / BN2;
    const QkRope rope{cs, q, k, (long long)M, Hq, Hq + Hkv};
    hipLaunchKernelGGL(gemm_bf16_w4_qkrope_kernel, dim3(m_tiles * n_tiles), dim3(256), W4_RING * W4_UNIT, s, A, W, bias, (const float*)nullptr, (void*)qkv, (int)M, N, K,
                       lda, ldc, m_tiles, n_tiles, raster_group_m(K * 2), rope);
    return hipGetLastError();
}

bool gemm_fp8_supports(int N, int K) { return N % BN2 == 0 && K % 64 == 0; }

template <int E>
static hipError_t launch_fp8_t(hipStream_t s, const uint8_t* Aq, const float* rowscale, const uint8_t* Wq, const float* colscale, const bf16_t* bias,
                               const bf16_t* R, void* C, int M, int N, int K, int ldc) {
    static uint32_t configured = 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint32_t bit = 1u << (dev & 31);
    if (!(__atomic_load_n(&configured, __ATOMIC_ACQUIRE) & bit)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_fp8_256pp_kernel<E>), hipFuncAttributeMaxDynamicSharedMemorySize, PP_STAGES * SUB_STAGE);
        if (e != hipSuccess) return e;
        __atomic_fetch_or(&configured, bit, __ATOMIC_RELEASE);
    }
    const int m_tiles = (M + BM2 - 1) / BM2, n_tiles = N / BN2;
    hipLaunchKernelGGL(gemm_fp8_256pp_kernel<E>, dim3(m_tiles * n_tiles), dim3(512), PP_STAGES * SUB_STAGE, s, Aq, rowscale, Wq, colscale, bias, R, C, M, N, K,
                       ldc, m_tiles, n_tiles, raster_group_m(K));
    return hipGetLastError();
}

hipError_t launch_gemm_fp8(hipStream_t s, const uint8_t* Aq, const float* rowscale, const uint8_t* Wq, const float* colscale, const bf16_t* bias,
                           const bf16_t* R, void* C, int64_t M, int N, int K, int ldc, int epi) {
    if (M <= 0) return hipSuccess;
    if (!gemm_fp8_supports(N, K) || ldc % 8 != 0 || !rowscale || !colscale) return hipErrorInvalidValue;
    switch (epi) {
        case EPI_NONE: return launch_fp8_t<EPI_NONE>(s, Aq, rowscale, Wq, colscale, bias, R, C, (int)M, N, K, ldc);
        case EPI_RESIDUAL: return launch_fp8_t<EPI_RESIDUAL>(s, Aq, rowscale, Wq, colscale, bias, R, C, (int)M, N, K, ldc);
        case EPI_SWIGLU: return launch_fp8_t<EPI_SWIGLU>(s, Aq, rowscale, Wq, colscale, bias, R, C, (int)M, N, K, ldc);
        case EPI_GELU: return launch_fp8_t<EPI_GELU>(s, Aq, rowscale, Wq, colscale, bias, R, C, (int)M, N, K, ldc);
        default: return hipErrorInvalidValue;
    }
}
