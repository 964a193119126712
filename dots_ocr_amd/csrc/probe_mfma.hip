// MFMA fragment-layout probe for gfx950.
//
// Every MFMA kernel in this engine (GEMM, flash attention, decode attention, skinny GEMM)
// relies on the lane->element maps below.  This probe computes D = A*B with the two bf16
// shapes we use, loading A/B and storing D through exactly those maps, so a host test with
// asymmetric operands (tests/test_mfma_layout.py) proves the maps on real hardware.
//
//   v_mfma_f32_32x32x16_bf16 : A[i=l&31][k=8*(l>>5)+e]  B[k=8*(l>>5)+e][j=l&31]   e in [0,8)
//                              D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]          r in [0,16)
//   v_mfma_f32_16x16x32_bf16 : A[i=l&15][k=8*(l>>4)+e]  B[k=8*(l>>4)+e][j=l&15]
//                              D[row=4*(l>>4)+r][col=l&15]                        r in [0,4)
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// A: [32][16] bf16 row-major, Bt: [32][16] bf16 (= B transposed, row j holds B[:, j]), D: [32][32] f32
__global__ void probe_mfma_32x32x16(const uint16_t* A, const uint16_t* Bt, float* D) {
    const int l = threadIdx.x;
    bf16x8 a = *reinterpret_cast<const bf16x8*>(A + (l & 31) * 16 + 8 * (l >> 5));
    bf16x8 b = *reinterpret_cast<const bf16x8*>(Bt + (l & 31) * 16 + 8 * (l >> 5));
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        D[row * 32 + (l & 31)] = c[r];
    }
}

// A: [16][32], Bt: [16][32], D: [16][16]
__global__ void probe_mfma_16x16x32(const uint16_t* A, const uint16_t* Bt, float* D) {
    const int l = threadIdx.x;
    bf16x8 a = *reinterpret_cast<const bf16x8*>(A + (l & 15) * 32 + 8 * (l >> 4));
    bf16x8 b = *reinterpret_cast<const bf16x8*>(Bt + (l & 15) * 32 + 8 * (l >> 4));
    f32x4 c = {0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

// fp8 (OCP e4m3) on the block-scaled instruction with unit scales (E8M0 0x7f = 2^0), the only K = 64 low-precision MFMA of gfx950:
//   v_mfma_scale_f32_32x32x64_f8f6f4 : lane l supplies 32 bytes of A row i = l & 31 and of B column j = l & 31, both the SAME 32
//   contraction positions k = 32*(l>>5) + [0, 32) in the same byte order; D as the bf16 32x32 shape.  Only the A <-> B pairing
//   of byte positions matters for a GEMM (a dot product is order-free), and that is what the test proves.
// A: [32][64] e4m3 row-major, Bt: [32][64] e4m3 (row j = B[:, j]), D: [32][32] f32
typedef __attribute__((ext_vector_type(8))) int i32x8;
__global__ void probe_mfma_32x32x64_fp8(const uint8_t* A, const uint8_t* Bt, float* D) {
    const int l = threadIdx.x;
    i32x8 a = *reinterpret_cast<const i32x8*>(A + (l & 31) * 64 + 32 * (l >> 5));
    i32x8 b = *reinterpret_cast<const i32x8*>(Bt + (l & 31) * 64 + 32 * (l >> 5));
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        D[row * 32 + (l & 31)] = c[r];
    }
}

// LDS-DMA probe: each lane copies 16 B global -> LDS (wave-uniform base + lane*16), then
// the wave writes the LDS image back out.  out[i] must equal in[i] for 256 dwords.
__global__ void probe_glds(const uint32_t* in, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[256];
    const int l = threadIdx.x;
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(in + l * 4),
        (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = lds[l * 4 + i];
}

extern "C" int dots_probe_mfma(int which, const void* A, const void* Bt, void* D, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (which == 0) hipLaunchKernelGGL(probe_mfma_32x32x16, dim3(1), dim3(64), 0, s, (const uint16_t*)A, (const uint16_t*)Bt, (float*)D);
    else if (which == 1) hipLaunchKernelGGL(probe_mfma_16x16x32, dim3(1), dim3(64), 0, s, (const uint16_t*)A, (const uint16_t*)Bt, (float*)D);
    else if (which == 3) hipLaunchKernelGGL(probe_mfma_32x32x64_fp8, dim3(1), dim3(64), 0, s, (const uint8_t*)A, (const uint8_t*)Bt, (float*)D);
    else hipLaunchKernelGGL(probe_glds, dim3(1), dim3(64), 0, s, (const uint32_t*)A, (uint32_t*)D);
    return (int)hipGetLastError();
}
