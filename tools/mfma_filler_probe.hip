// What does a VALU "filler" between two MFMAs cost on gfx950, one wave per SIMD?  (round 4, flash_attn64 design probe)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_filler_probe.hip -o tools/bin/mfma_filler_probe && tools/bin/mfma_filler_probe
// Every variant: 256 workgroups x 256 threads (1 wave per SIMD), a loop of 8 independent-accumulator v_mfma_f32_32x32x16_bf16 per
// iteration with F filler instructions behind each MFMA.  Reported: shader cycles per MFMA (s_memtime), median over waves.
// Variants: operand placement of the MFMA (A/B in VGPRs or AGPRs, C/D in AGPRs or VGPRs) x filler kind (v_max3, v_fma, v_exp, v_add chain).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE: 0 = A,B VGPR, C/D AGPR (builtin-like);  1 = A VGPR, B AGPR, C/D VGPR (score MFMA of flash_attn64);  2 = A,B,C,D all AGPR;  3 = A VGPR, B AGPR, C/D AGPR
// FILL: number of fillers per MFMA;  KIND: 0 v_max3 (independent chain per slot), 1 v_fma, 2 v_exp, 3 fma+exp+add+cvt mix (the exp group: 7 instr, FILL ignored)
template <int MODE, int FILL, int KIND>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, int iters, const float* seed) {
    f32x16 acc[8];
    bf16x8 a[2], b[2];
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        f[i] = seed[(threadIdx.x + i) & 63];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[i][e] = (short)(0x3c00 + threadIdx.x + e); b[i][e] = (short)(0x3c10 + i); }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if constexpr (MODE == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a[m & 1]), "v"(b[m & 1]));
            if constexpr (MODE == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a[m & 1]), "a"(b[m & 1]));
            if constexpr (MODE == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m]) : "a"(a[m & 1]), "a"(b[m & 1]));
            if constexpr (MODE == 3) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a[m & 1]), "a"(b[m & 1]));
            if constexpr (KIND == 3) {
                float t0_, t1_; unsigned w_;
                asm volatile("v_fma_f32 %2, %4, %6, %7\n\tv_fma_f32 %3, %5, %6, %7\n\tv_exp_f32_e32 %2, %2\n\tv_exp_f32_e32 %3, %3\n\t"
                             "v_add_f32_e32 %0, %0, %2\n\tv_add_f32_e32 %0, %0, %3\n\tv_cvt_pk_bf16_f32 %1, %2, %3"
                             : "+v"(f[0]), "=&v"(w_), "=&v"(t0_), "=&v"(t1_) : "v"(f[1]), "v"(f[2]), "s"(1.0001f), "v"(f[3]));
                f[4] = __uint_as_float(w_ | 0x3f000000u);
            } else {
#pragma unroll
                for (int k = 0; k < FILL; ++k) {
                    if constexpr (KIND == 0) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[k & 7]) : "v"(f[(k + 1) & 7]), "v"(f[(k + 2) & 7]));
                    if constexpr (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k & 7]) : "v"(f[(k + 1) & 7]), "v"(f[(k + 2) & 7]));
                    if constexpr (KIND == 2) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(f[k & 7]));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s += f[i]; s += acc[i][0] + acc[i][7]; }
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (s == 123.456f) out[0] = 0;
}

template <int MODE, int FILL, int KIND>
static void run(const char* name, unsigned long long* d, const float* seed) {
    const int iters = 2000;
    hipLaunchKernelGGL((probe<MODE, FILL, KIND>), dim3(256), dim3(256), 0, 0, d, iters, seed);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((probe<MODE, FILL, KIND>), dim3(256), dim3(256), 0, 0, d, iters, seed);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(1024);
    CK(hipMemcpy(h.data(), d, 8192, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    printf("%-64s %7.2f cycles / MFMA (median wave; min %.2f max %.2f)\n", name, (double)h[512] / iters / 8, (double)h[0] / iters / 8, (double)h[1023] / iters / 8);
}

int main() {
    unsigned long long* d; float* seed;
    CK(hipMalloc(&d, 8192)); CK(hipMalloc(&seed, 256));
    std::vector<float> hs(64); for (int i = 0; i < 64; ++i) hs[i] = 0.5f + 0.01f * i;
    CK(hipMemcpy(seed, hs.data(), 256, hipMemcpyHostToDevice));
    printf("readcyclecounter units; v_mfma_f32_32x32x16_bf16 floor = 32 shader cycles per SIMD if the counter runs at the shader clock\n");
#define R(M, F, K, nm) run<M, F, K>(nm, d, seed)
    R(0, 0, 0, "A,B vgpr  C/D agpr | bare");
    R(0, 1, 0, "A,B vgpr  C/D agpr | 1 x v_max3");
    R(0, 3, 0, "A,B vgpr  C/D agpr | 3 x v_max3");
    R(0, 5, 0, "A,B vgpr  C/D agpr | 5 x v_max3");
    R(0, 7, 0, "A,B vgpr  C/D agpr | 7 x v_max3");
    R(0, 5, 1, "A,B vgpr  C/D agpr | 5 x v_fma");
    R(0, 2, 2, "A,B vgpr  C/D agpr | 2 x v_exp");
    R(0, 0, 3, "A,B vgpr  C/D agpr | exp group (7)");
    R(2, 0, 0, "A,B agpr  C/D agpr | bare");
    R(2, 1, 0, "A,B agpr  C/D agpr | 1 x v_max3");
    R(2, 5, 0, "A,B agpr  C/D agpr | 5 x v_max3");
    R(2, 7, 0, "A,B agpr  C/D agpr | 7 x v_max3");
    R(2, 0, 3, "A,B agpr  C/D agpr | exp group (7)");
    R(3, 0, 0, "A vgpr B agpr  C/D agpr | bare");
    R(3, 5, 0, "A vgpr B agpr  C/D agpr | 5 x v_max3");
    R(3, 0, 3, "A vgpr B agpr  C/D agpr | exp group (7)");
    R(1, 0, 0, "A vgpr B agpr  C/D VGPR | bare");
    R(1, 1, 0, "A vgpr B agpr  C/D VGPR | 1 x v_max3");
    R(1, 5, 0, "A vgpr B agpr  C/D VGPR | 5 x v_max3");
    R(1, 0, 3, "A vgpr B agpr  C/D VGPR | exp group (7)");
    return 0;
}
