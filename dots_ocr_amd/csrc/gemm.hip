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

#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 128;            // one operand tile: 128 rows x 64 bf16
constexpr int GROUP_M = 8;

DEVI float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
DEVI float silu(float x) { return x / (1.0f + __expf(-x)); }

// ---- epilogue shared by both tile shapes: lane owns (m, 4 consecutive n) per (fn, fm, rq) of its 64x64 wave tile.
// nw0 = first weight row (n) of the wave tile, mw0 = first activation row (m) of the wave tile.
template <int EPI, int FN>
DEVI void gemm_epilogue(const f32x16 (&acc)[FN][2], const bf16_t* __restrict__ bias, const float* __restrict__ colscale, const bf16_t* R,
                        void* Cout, int M, int ldc, int nw0, int mw0, int l31, int hi) {
#pragma unroll
    for (int fm = 0; fm < 2; ++fm) {
        const int m = mw0 + fm * 32 + l31;
        if (m >= M) continue;
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
                    if (colscale) { g *= colscale[nb + e]; u *= colscale[nb + 32 + e]; }       // fp8 weights: per-output-channel scale (quant.hip)
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
                        for (int e = 0; e < 4; ++e) o[e] *= sc[e];
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

}  // namespace

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
    hipLaunchKernelGGL(gemm_bf16_256_kernel<E>, dim3(m_tiles * n_tiles), dim3(512), 2 * STAGE2, s, A, W, bias, colscale, R, C, M, N, K,
                       lda, ldc, m_tiles, n_tiles);
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
