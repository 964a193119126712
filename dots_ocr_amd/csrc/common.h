// Shared device helpers for the gfx950 kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;   // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) short bf16x8;      // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;     // 32x32 accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;       // 16x16 accumulator
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define DEVI __device__ __forceinline__

DEVI float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// f32 -> bf16, round-to-nearest-even.  The native __bf16 cast lowers to ONE v_cvt_pk_bf16_f32 on gfx950
// (a software RNE with a NaN branch costs ~10 VALU + an exec-mask branch per element: measured as the
// dominant VALU cost of the flash-attention loop in round 1).
typedef __bf16 bf16x2_native __attribute__((ext_vector_type(2)));
DEVI bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
DEVI uint32_t pack_bf2(float lo, float hi) {
    bf16x2_native v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

DEVI float lo_bf(uint32_t w) { return __uint_as_float(w << 16); }
DEVI float hi_bf(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

DEVI float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
DEVI float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware bijective remap of a 1-D block id: the dispatcher places block b on XCD b % 8;
// give every XCD one contiguous chunk of the logical tile order so neighbouring tiles share
// an L2 (speed only, never correctness).
DEVI int xcd_remap(int bid, int nblk) {
    const int NX = 8;
    int q = nblk / NX, r = nblk % NX;
    int xcd = bid % NX, idx = bid / NX;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Phase timestamps for tools/decode_bench.hip (built with -DDOTS_TRACE only; the product build has no trace code):
// TRACE(slot) stores the 100 MHz wall clock of lane 0 of every wave into trace[(block * 16 + wave) * 8 + slot].
#ifdef DOTS_TRACE
#define TRACE_DECL static __device__ unsigned long long* dots_trace_buf = nullptr;
#define TRACE(slot)                                                                                                          \
    do {                                                                                                                     \
        if ((threadIdx.x & 63) == 0 && dots_trace_buf)                                                                       \
            dots_trace_buf[((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 16 + (threadIdx.x >> 6)) * 8 + (slot)] = wall_clock64(); \
    } while (0)
#else
#define TRACE_DECL
#define TRACE(slot)
#endif

#define HIP_CHECK_RET(expr)                                  \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) return _e;                     \
    } while (0)
