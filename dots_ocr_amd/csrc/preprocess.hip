// GPU image preprocessing: uint8 RGB page -> float32 patches [gh*gw, C*P*P], bit-identical to
// the host path (Pillow BICUBIC resize -> x 1/255 -> (x - mean)/std -> patchify; reference parser.py:99-105 via
// transformers image_processing_pil_qwen2_vl.py:126-246).
//
// Pillow's resampler is integer arithmetic: per axis, precomputed taps in 22-bit fixed point, accumulate in int32
// starting from 2^21, arithmetic shift by 22, clip to uint8 — horizontal pass first, its uint8 result feeds the
// vertical pass.  The tap tables come from the host (dots_ocr_amd/image_utils.py: bicubic_resample_tables, checked
// against Pillow itself), the kernels only do the exact integer MACs, so GPU == PIL bit for bit.  HBM-bound byte
// work: ~3 B in + 3 B out per pixel per pass, then 3 B in / 12 B out for normalise + patchify.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int PREC = 22;

DEVI uint8_t clip8(int v) {
    v >>= PREC;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// in [H][W][3] -> out [H][rw][3]; one thread per output pixel
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                       const int32_t* __restrict__ coef, const int32_t* __restrict__ bounds,
                                                       int ksize, int H, int W, int rw) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)H * rw) return;
    const int y = (int)(idx / rw), xx = (int)(idx - (int64_t)y * rw);
    const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int32_t* k = coef + (size_t)xx * ksize;
    const uint8_t* p = in + ((size_t)y * W + x0) * 3;
    int s0 = 1 << (PREC - 1), s1 = s0, s2 = s0;
    for (int t = 0; t < n; ++t) {
        const int c = k[t];
        s0 += p[3 * t] * c; s1 += p[3 * t + 1] * c; s2 += p[3 * t + 2] * c;
    }
    uint8_t* o = out + idx * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// in [H][W][3] -> out [rh][W][3]
__global__ __launch_bounds__(256) void resize_v_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                       const int32_t* __restrict__ coef, const int32_t* __restrict__ bounds,
                                                       int ksize, int W, int rh) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)rh * W) return;
    const int yy = (int)(idx / W), x = (int)(idx - (int64_t)yy * W);
    const int y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
    const int32_t* k = coef + (size_t)yy * ksize;
    const uint8_t* p = in + ((size_t)y0 * W + x) * 3;
    int s0 = 1 << (PREC - 1), s1 = s0, s2 = s0;
    for (int t = 0; t < n; ++t) {
        const int c = k[t];
        const uint8_t* q = p + (size_t)t * W * 3;
        s0 += q[0] * c; s1 += q[1] * c; s2 += q[2] * c;
    }
    uint8_t* o = out + idx * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// img [rh][rw][3] u8 -> pixel_values f32 [gh*gw][3*P*P], patches block-major over merge x merge groups,
// channel-major inside a patch.  Same float32 operation order as numpy on the host: (u8 * r255 - mean) / std.
__global__ __launch_bounds__(256) void normalize_patchify_kernel(const uint8_t* __restrict__ img, float* __restrict__ out,
                                                                 int rw, int gh, int gw, int P, int m, float r255,
                                                                 float mean0, float mean1, float mean2, float std0, float std1, float std2) {
#pragma clang fp contract(off)          // numpy rounds u8*r255, then the subtraction, then the division: no FMA here
    const int pd = 3 * P * P;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)gh * gw * pd) return;
    const int64_t p = idx / pd;
    const int e = (int)(idx - p * pd);
    const int c = e / (P * P), py = (e / P) % P, px = e % P;
    const int mw = (int)(p % m), mh = (int)((p / m) % m);
    const int64_t blk = p / (m * m);
    const int bw = (int)(blk % (gw / m)), bh = (int)(blk / (gw / m));
    const int y = (bh * m + mh) * P + py, x = (bw * m + mw) * P + px;
    const float v = (float)img[((size_t)y * rw + x) * 3 + c] * r255;
    const float mean = c == 0 ? mean0 : (c == 1 ? mean1 : mean2);
    const float sd = c == 0 ? std0 : (c == 1 ? std1 : std2);
    const float centred = v - mean;
    out[idx] = centred / sd;               // IEEE-correct f32 division (hipcc default)
}

}  // namespace

hipError_t launch_resize_h(hipStream_t s, const uint8_t* in, uint8_t* out, const int32_t* coef, const int32_t* bounds,
                           int ksize, int H, int W, int rw) {
    const int64_t n = (int64_t)H * rw;
    hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, coef, bounds, ksize, H, W, rw);
    return hipGetLastError();
}

hipError_t launch_resize_v(hipStream_t s, const uint8_t* in, uint8_t* out, const int32_t* coef, const int32_t* bounds,
                           int ksize, int W, int rh) {
    const int64_t n = (int64_t)rh * W;
    hipLaunchKernelGGL(resize_v_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, coef, bounds, ksize, W, rh);
    return hipGetLastError();
}

hipError_t launch_normalize_patchify(hipStream_t s, const uint8_t* img, float* out, int rw, int gh, int gw, int P, int m,
                                     float r255, const float* mean, const float* stdv) {
    const int64_t n = (int64_t)gh * gw * 3 * P * P;
    hipLaunchKernelGGL(normalize_patchify_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, img, out, rw, gh, gw, P, m, r255,
                       mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
    return hipGetLastError();
}
