// Image preprocessing in front of encode_image: the transform `clip.load` returns and the reference
// applies to every PIL image (predictions_runner.py:212, embeddings_generator.py:72):
//     Resize(n_px, BICUBIC) -> CenterCrop(n_px) -> ToTensor -> Normalize(mean, std)      (mean / std: :121)
// and the stretch variant `clip_transform_full` (:116-122).  torchvision's Resize on a PIL image is
// Image.resize(..., BICUBIC), i.e. Pillow's 8-bit resampler: separable two-pass convolution (horizontal first, uint8
// between the passes), antialiased support 2 * max(scale, 1), coefficients normalised in double and rounded to 22-bit
// fixed point.  Both passes are restated here exactly (double arithmetic in Pillow's operation order, contraction
// off), so the uint8 result is bit-identical to PIL's and the float tensor to ToTensor + Normalize in fp32.
// HBM-bound byte work: one thread per output pixel (3 channels), coefficients recomputed per thread (<= ~40 taps).
#include "common.h"

#pragma clang fp contract(off)

namespace capdec {

constexpr int PIL_PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ double pil_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// Pillow precompute_coeffs for ONE output index: first tap, tap count, and what is needed to regenerate the taps
struct Taps {
    double center, ss, ww;
    int xmin, cnt;
};
__device__ __forceinline__ Taps pil_taps(int in_size, int out_size, int xx) {
    Taps t;
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    t.ss = 1.0 / filterscale;
    t.center = (xx + 0.5) * scale;
    int xmin = (int)(t.center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(t.center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    t.xmin = xmin;
    t.cnt = xmax - xmin;
    double ww = 0.0;
    for (int x = 0; x < t.cnt; ++x) ww += pil_bicubic((x + xmin - t.center + 0.5) * t.ss);
    t.ww = ww;
    return t;
}
// normalize_coeffs_8bpc: tap x as 22-bit fixed point
__device__ __forceinline__ int pil_tap_fixed(const Taps &t, int x) {
    double w = pil_bicubic((x + t.xmin - t.center + 0.5) * t.ss);
    if (t.ww != 0.0) w /= t.ww;
    return w < 0 ? (int)(-0.5 + w * (double)(1 << PIL_PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << PIL_PRECISION_BITS));
}
__device__ __forceinline__ int pil_clip8(int v) {
    v >>= PIL_PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass, only the n_px columns the crop keeps:  inter[img][y][xo][c], y over ALL input rows
__global__ void preprocess_h_kernel(const uint8_t *__restrict__ rgb, const ImageDesc *__restrict__ desc,
                                    uint8_t *__restrict__ inter, int n_px) {
    const ImageDesc d = desc[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.H * n_px) return;
    const int y = i / n_px, xo = i - y * n_px;
    const Taps t = pil_taps(d.W, d.rw, d.left + xo);
    const uint8_t *src = rgb + d.off + ((size_t)y * d.W + t.xmin) * 3;
    int a0 = 1 << (PIL_PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int x = 0; x < t.cnt; ++x) {
        const int k = pil_tap_fixed(t, x);
        a0 += src[3 * x] * k;
        a1 += src[3 * x + 1] * k;
        a2 += src[3 * x + 2] * k;
    }
    uint8_t *dst = inter + d.ioff + (size_t)i * 3;
    dst[0] = (uint8_t)pil_clip8(a0);
    dst[1] = (uint8_t)pil_clip8(a1);
    dst[2] = (uint8_t)pil_clip8(a2);
}

// vertical pass over the crop rows + ToTensor + Normalize:  out[img][c][yo][xo]
__global__ void preprocess_v_kernel(const uint8_t *__restrict__ inter, const ImageDesc *__restrict__ desc,
                                    float *__restrict__ out, int n_px, float m0, float m1, float m2, float s0,
                                    float s1, float s2) {
    const ImageDesc d = desc[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_px * n_px) return;
    const int yo = i / n_px, xo = i - yo * n_px;
    const Taps t = pil_taps(d.H, d.rh, d.top + yo);
    const uint8_t *src = inter + d.ioff + ((size_t)t.xmin * n_px + xo) * 3;
    int a0 = 1 << (PIL_PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int y = 0; y < t.cnt; ++y) {
        const int k = pil_tap_fixed(t, y);
        const uint8_t *p = src + (size_t)y * n_px * 3;
        a0 += p[0] * k;
        a1 += p[1] * k;
        a2 += p[2] * k;
    }
    float *o = out + (size_t)blockIdx.y * 3 * n_px * n_px + i;
    const size_t plane = (size_t)n_px * n_px;
    o[0] = ((float)pil_clip8(a0) / 255.0f - m0) / s0;            // ToTensor (/255) then Normalize, fp32 like torch
    o[plane] = ((float)pil_clip8(a1) / 255.0f - m1) / s1;
    o[2 * plane] = ((float)pil_clip8(a2) / 255.0f - m2) / s2;
}

int launch_preprocess(hipStream_t st, const uint8_t *rgb, const ImageDesc *desc, int n, int max_h, int n_px,
                      uint8_t *inter, float *out, const float *mean, const float *stdv) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(preprocess_h_kernel, dim3((max_h * n_px + 255) / 256, n), dim3(256), 0, st, rgb, desc, inter,
                       n_px);
    hipLaunchKernelGGL(preprocess_v_kernel, dim3((n_px * n_px + 255) / 256, n), dim3(256), 0, st, inter, desc, out,
                       n_px, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

}  // namespace capdec
