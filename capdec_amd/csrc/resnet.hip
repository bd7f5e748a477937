// Glue kernels of CLIP's ModifiedResNet image tower (RN50x4: the reference's default backbone, call sites
// predictions_runner.py:158,220; embeddings_generator.py:89,113).  All arithmetic-heavy work is the MFMA GEMM: a 1x1
// convolution is a GEMM on NHWC activations, a 3x3 convolution is im2col + GEMM, BatchNorm (inference statistics) is
// folded into the convolution's weights and bias at load time, ReLU / residual-add-ReLU live in the GEMM epilogue.
// What remains here is memory-bound data movement: im2col, 2x2 average pooling (the tower's anti-aliased stride), and
// the attention pool's token assembly and single-query attention.
//
// Layout: activations are NHWC fp32 with the channel count padded to a multiple of 64 (zero channels: the folded
// weights have zero rows / columns there), so every GEMM has K % 64 == 0 and runs on the packed-operand kernels.
#include "bf16x3.h"
#include "common.h"

namespace capdec {

// ---- im2col straight into the packed GEMM operand (format fmt, tile-major planes: bf16x3.h): one thread per (matrix
// row = output pixel, k-step = 16 consecutive (tap, channel) columns).  The 128 threads of one (row tile, k-step) block
// are consecutive, so a block of the operand is written whole; rows beyond M are written as zeros.
__global__ void im2col3x3_packed_kernel(const float *__restrict__ in, char *__restrict__ out, int M, int H, int W, int C,
                                        int stride, int Ho, int Wo, int nk, int fmt, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t blk = i >> 7;
    const int ks = (int)(blk % nk);
    const int row = (int)(blk / nk) * 128 + (int)(i & 127);
    const int k0 = ks << 4, tap = k0 / C, c0 = k0 - tap * C;                  // C % 16 == 0: a k-step stays within one tap
    float4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < M) {
        const int ox = row % Wo, oy = (row / Wo) % Ho, n = row / (Wo * Ho);
        const int iy = oy * stride + tap / 3 - 1, ix = ox * stride + tap % 3 - 1;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            const float4 *p = reinterpret_cast<const float4 *>(in + (((size_t)n * H + iy) * W + ix) * C + c0);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = p[q];
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) x3_store_quad(out, nk, row, ks, q, v[q], fmt);
}

// first convolution: NCHW fp32 pixels, 3 channels, K = 27 padded to 64 (four k-steps)
__global__ void im2col3x3_nchw3_packed_kernel(const float *__restrict__ in, char *__restrict__ out, int M, int H, int W,
                                              int stride, int Ho, int Wo, int nk, int fmt, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t blk = i >> 7;
    const int ks = (int)(blk % nk);
    const int row = (int)(blk / nk) * 128 + (int)(i & 127);
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = 0.f;
    if (row < M && ks < 2) {
        const int ox = row % Wo, oy = (row / Wo) % Ho, n = row / (Wo * Ho);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = (ks << 4) + j;
            if (k < 27) {
                const int tap = k / 3, ch = k - tap * 3;
                const int iy = oy * stride + tap / 3 - 1, ix = ox * stride + tap % 3 - 1;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) v[j] = in[(((size_t)n * 3 + ch) * H + iy) * W + ix];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        x3_store_quad(out, nk, row, ks, q, make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]), fmt);
}

int launch_im2col3x3_packed(hipStream_t st, const float *in, void *out, int N, int H, int W, int C, int stride, bool nchw3,
                            int Kp, int fmt) {
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    if (N <= 0) return 0;
    CAPDEC_CHECK((size_t)N * Ho * Wo < ((size_t)1 << 31) - 128, "im2col: too many output pixels for one launch");
    const int M = N * Ho * Wo, nk = Kp / 16;
    const size_t total = (size_t)((M + 127) / 128) * 128 * nk;
    if (nchw3) {
        CAPDEC_CHECK(Kp == 64, "im2col: the 3-channel convolution is padded to K = 64");
        hipLaunchKernelGGL(im2col3x3_nchw3_packed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in,
                           (char *)out, M, H, W, stride, Ho, Wo, nk, fmt, total);
    } else {
        CAPDEC_CHECK(C % 16 == 0 && Kp == 9 * C, "im2col: channel count must be a multiple of 16");
        hipLaunchKernelGGL(im2col3x3_packed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in, (char *)out,
                           M, H, W, C, stride, Ho, Wo, nk, fmt, total);
    }
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// ---- im2col for a 3x3 convolution with padding 1: out[(n, oy, ox)][(ky, kx, c)] = in[n][oy*s+ky-1][ox*s+kx-1][c]
// (NHWC input, C % 4 == 0); one thread per (output pixel, tap, 4 channels)
__global__ void im2col3x3_nhwc_kernel(const float *__restrict__ in, float *__restrict__ out, int N, int H, int W, int C,
                                      int stride, int Ho, int Wo) {
    const int c4 = C >> 2;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)N * Ho * Wo * 9 * c4;
    if (i >= total) return;
    const int cq = (int)(i % c4);
    size_t r = i / c4;
    const int tap = (int)(r % 9);
    r /= 9;                                              // output pixel index
    const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho), n = (int)(r / ((size_t)Wo * Ho));
    const int iy = oy * stride + tap / 3 - 1, ix = ox * stride + tap % 3 - 1;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (iy >= 0 && iy < H && ix >= 0 && ix < W)
        v = reinterpret_cast<const float4 *>(in + (((size_t)n * H + iy) * W + ix) * C)[cq];
    reinterpret_cast<float4 *>(out + r * (size_t)(9 * C) + (size_t)tap * C)[cq] = v;
}

// first convolution: NCHW fp32 pixels with 3 channels -> rows of Kp (>= 27, zero padded) in (ky, kx, c) order
__global__ void im2col3x3_nchw3_kernel(const float *__restrict__ in, float *__restrict__ out, int N, int H, int W,
                                       int stride, int Ho, int Wo, int Kp) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // one thread per (output pixel, k)
    const size_t total = (size_t)N * Ho * Wo * Kp;
    if (i >= total) return;
    const int k = (int)(i % Kp);
    const size_t r = i / Kp;
    float v = 0.f;
    if (k < 27) {
        const int tap = k / 3, c = k - tap * 3;
        const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho), n = (int)(r / ((size_t)Wo * Ho));
        const int iy = oy * stride + tap / 3 - 1, ix = ox * stride + tap % 3 - 1;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = in[(((size_t)n * 3 + c) * H + iy) * W + ix];
    }
    out[i] = v;
}

int launch_im2col3x3(hipStream_t st, const float *in, float *out, int N, int H, int W, int C, int stride, bool nchw3,
                     int Kp) {
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    if (N <= 0) return 0;
    if (nchw3) {
        const size_t tot = (size_t)N * Ho * Wo * Kp;
        hipLaunchKernelGGL(im2col3x3_nchw3_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, in, out, N, H, W,
                           stride, Ho, Wo, Kp);
    } else {
        CAPDEC_CHECK(C % 4 == 0 && Kp == 9 * C, "im2col: channel count must be a multiple of 4");
        const size_t tot = (size_t)N * Ho * Wo * 9 * (C >> 2);
        hipLaunchKernelGGL(im2col3x3_nhwc_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, in, out, N, H, W, C,
                           stride, Ho, Wo);
    }
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// ---- AvgPool2d(2) on NHWC (H, W even): the tower's stride
__global__ void avgpool2_nhwc_kernel(const float *__restrict__ in, float *__restrict__ out, int N, int H, int W, int C) {
    const int c4 = C >> 2, Ho = H >> 1, Wo = W >> 1;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * Ho * Wo * c4) return;
    const int cq = (int)(i % c4);
    const size_t r = i / c4;
    const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho), n = (int)(r / ((size_t)Wo * Ho));
    const float *p = in + (((size_t)n * H + 2 * oy) * W + 2 * ox) * C;
    const float4 a = reinterpret_cast<const float4 *>(p)[cq], b = reinterpret_cast<const float4 *>(p + C)[cq];
    const float4 c = reinterpret_cast<const float4 *>(p + (size_t)W * C)[cq],
                 d = reinterpret_cast<const float4 *>(p + (size_t)W * C + C)[cq];
    float4 o;      // torch avg_pool2d: sum of the window, then one division
    o.x = ((a.x + b.x) + (c.x + d.x)) * 0.25f; o.y = ((a.y + b.y) + (c.y + d.y)) * 0.25f;
    o.z = ((a.z + b.z) + (c.z + d.z)) * 0.25f; o.w = ((a.w + b.w) + (c.w + d.w)) * 0.25f;
    reinterpret_cast<float4 *>(out + r * (size_t)C)[cq] = o;
}

int launch_avgpool2(hipStream_t st, const float *in, float *out, int N, int H, int W, int C) {
    CAPDEC_CHECK(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "avgpool2: even spatial size, C % 4 == 0");
    const size_t tot = (size_t)N * (H >> 1) * (W >> 1) * (C >> 2);
    if (tot == 0) return 0;
    hipLaunchKernelGGL(avgpool2_nhwc_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, in, out, N, H, W, C);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// ---- AvgPool2d(2) between PACKED activations (format fmt): rows = pixels; one thread per (output pixel, k-step of 16
// channels), the 128 threads of one (row tile, k-step) block of the output consecutive.  A pixel's k-step is one 32-byte
// chunk per plane (its two 16-byte halves swapped when bit 3 of the row is set): 16-byte loads and stores throughout.
__device__ __forceinline__ void pk_load16(const char *packed, int nk, int row, int ks, int fmt, float (&v)[16]) {
    const int r = row & 127, sw = (r >> 3) & 1;
    if (fmt == PK_F16X2) {
        const char *p = packed + ((size_t)(row >> 7) * nk + ks) * H2_BLOCK_B + r * X3_ROW_B;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f16x8 h = *reinterpret_cast<const f16x8 *>(p + j * 16), l = *reinterpret_cast<const f16x8 *>(p + X3_PLANE_B + j * 16);
            const int k0 = (j ^ sw) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[k0 + e] = (float)h[e] + (float)l[e] * (1.0f / H2_LO_SCALE);
        }
    } else {
        const char *p = packed + ((size_t)(row >> 7) * nk + ks) * X3_PLANE_B + r * X3_ROW_B;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k0 = (j ^ sw) * 8;
            if (fmt == PK_F16X1) {
                const f16x8 h = *reinterpret_cast<const f16x8 *>(p + j * 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[k0 + e] = (float)h[e];
            } else {
                const bf16x8 h = *reinterpret_cast<const bf16x8 *>(p + j * 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[k0 + e] = (float)h[e];
            }
        }
    }
}
__global__ void avgpool2_packed_kernel(const char *__restrict__ in, char *__restrict__ out, int Mo, int H, int W, int nk,
                                       int fmt, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t blk = i >> 7;
    const int ks = (int)(blk % nk);
    const int row = (int)(blk / nk) * 128 + (int)(i & 127);
    const int Ho = H >> 1, Wo = W >> 1;
    float o[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = 0.f;
    if (row < Mo) {
        const int ox = row % Wo, oy = (row / Wo) % Ho, n = row / (Wo * Ho);
        const int p00 = (n * H + 2 * oy) * W + 2 * ox;
        float a[16], b[16], c[16], d[16];
        pk_load16(in, nk, p00, ks, fmt, a);
        pk_load16(in, nk, p00 + 1, ks, fmt, b);
        pk_load16(in, nk, p00 + W, ks, fmt, c);
        pk_load16(in, nk, p00 + W + 1, ks, fmt, d);
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = ((a[e] + b[e]) + (c[e] + d[e])) * 0.25f;   // torch: window sum, one division
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) x3_store_quad(out, nk, row, ks, q, make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]), fmt);
}

int launch_avgpool2_packed(hipStream_t st, const void *in, void *out, int N, int H, int W, int C, int fmt) {
    CAPDEC_CHECK(H % 2 == 0 && W % 2 == 0 && C % 16 == 0, "avgpool2 (packed): even spatial size, C % 16 == 0");
    CAPDEC_CHECK(fmt == PK_F16X2 || fmt == PK_F16X1 || fmt == PK_BF16X1, "avgpool2 (packed): f16x2 / f16 / bf16 operands");
    const int Mo = N * (H >> 1) * (W >> 1), nk = C / 16;
    const size_t total = (size_t)((Mo + 127) / 128) * 128 * nk;
    if (total == 0) return 0;
    hipLaunchKernelGGL(avgpool2_packed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const char *)in,
                       (char *)out, Mo, H, W, nk, fmt, total);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// ---- packed operand [M rows][C] -> fp32 [M, C] (the features in front of the attention pool)
__global__ void unpack_rows_kernel(const char *__restrict__ in, float *__restrict__ out, int M, int C, int fmt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // one thread per (row, quad)
    const int nq = C >> 2;
    if (i >= (size_t)M * nq) return;
    const int row = (int)(i / nq), qd = (int)(i - (size_t)row * nq);
    *reinterpret_cast<float4 *>(out + (size_t)row * C + 4 * qd) = x3_load_quad(in, C >> 4, row, qd >> 2, qd & 3, fmt);
}

int launch_unpack_rows(hipStream_t st, const void *in, float *out, int M, int C, int fmt) {
    CAPDEC_CHECK(C % 16 == 0 && (fmt == PK_F16X2 || fmt == PK_F16X1 || fmt == PK_BF16X1), "unpack_rows: C % 16 == 0, fp16 / bf16 formats");
    const size_t tot = (size_t)M * (C >> 2);
    if (tot == 0) return 0;
    hipLaunchKernelGGL(unpack_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const char *)in, out, M, C, fmt);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// ---- attention pool, token assembly: t[n][0] = mean over the HW feature rows, t[n][1 + i] = feature row i, + pos
__global__ void attnpool_tokens_kernel(const float *__restrict__ feat, const float *__restrict__ pos, float *__restrict__ t,
                                       int N, int HW, int C) {
    const int c4 = C >> 2;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // one thread per (image, 4 channels)
    if (i >= (size_t)N * c4) return;
    const int cq = (int)(i % c4), n = (int)(i / c4);
    const float *f = feat + (size_t)n * HW * C;
    float *o = t + (size_t)n * (HW + 1) * C;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < HW; ++j) {
        const float4 v = reinterpret_cast<const float4 *>(f + (size_t)j * C)[cq];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        const float4 p = reinterpret_cast<const float4 *>(pos + (size_t)(j + 1) * C)[cq];
        reinterpret_cast<float4 *>(o + (size_t)(j + 1) * C)[cq] = make_float4(v.x + p.x, v.y + p.y, v.z + p.z, v.w + p.w);
    }
    const float inv = 1.0f / (float)HW;
    const float4 p0 = reinterpret_cast<const float4 *>(pos)[cq];
    reinterpret_cast<float4 *>(o)[cq] = make_float4(s.x * inv + p0.x, s.y * inv + p0.y, s.z * inv + p0.z, s.w * inv + p0.w);
}

int launch_attnpool_tokens(hipStream_t st, const float *feat, const float *pos, float *t, int N, int HW, int C) {
    const size_t tot = (size_t)N * (C >> 2);
    if (tot == 0) return 0;
    hipLaunchKernelGGL(attnpool_tokens_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, feat, pos, t, N, HW, C);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// ---- attention pool, the single query: one wavefront per (image, head); head_dim 64 as four 16-lane groups (a group
// owns one key per iteration, a lane a float4 of the head); q is scaled by head_dim^-0.5 (nn.MultiheadAttention)
__global__ __launch_bounds__(256) void attnpool_attend_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                              const float *__restrict__ v, float *__restrict__ out,
                                                              int total, int heads, int T, int C) {
    __shared__ float sc[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, sub = lane & 15;
    const int gw = blockIdx.x * 4 + wave;
    const bool active = gw < total;
    const int n = active ? gw / heads : 0, h = active ? gw - n * heads : 0;
    float4 qv = reinterpret_cast<const float4 *>(q + (size_t)n * C + h * 64)[sub];
    qv.x *= 0.125f; qv.y *= 0.125f; qv.z *= 0.125f; qv.w *= 0.125f;
    const float *kb = k + (size_t)n * T * C + h * 64, *vb = v + (size_t)n * T * C + h * 64;
    for (int t0 = 0; t0 < T; t0 += 4) {
        const int t = t0 + grp;
        if (t < T) {
            const float4 kk = reinterpret_cast<const float4 *>(kb + (size_t)t * C)[sub];
            const float s = row16_sum((qv.x * kk.x + qv.y * kk.y) + (qv.z * kk.z + qv.w * kk.w));
            if (sub == 0) sc[wave][t] = s;
        }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int t = lane; t < T; t += 64) mx = fmaxf(mx, sc[wave][t]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int t = lane; t < T; t += 64) {
        const float e = expf(sc[wave][t] - mx);
        sc[wave][t] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    __syncthreads();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t0 = 0; t0 < T; t0 += 4) {
        const int t = t0 + grp;
        if (t < T) {
            const float4 vv = reinterpret_cast<const float4 *>(vb + (size_t)t * C)[sub];
            const float w = sc[wave][t];
            acc.x += w * vv.x; acc.y += w * vv.y; acc.z += w * vv.z; acc.w += w * vv.w;
        }
    }
    acc.x += __shfl_xor(acc.x, 16, 64); acc.y += __shfl_xor(acc.y, 16, 64);
    acc.z += __shfl_xor(acc.z, 16, 64); acc.w += __shfl_xor(acc.w, 16, 64);
    acc.x += __shfl_xor(acc.x, 32, 64); acc.y += __shfl_xor(acc.y, 32, 64);
    acc.z += __shfl_xor(acc.z, 32, 64); acc.w += __shfl_xor(acc.w, 32, 64);
    if (active && grp == 0) {
        const float inv = 1.0f / sum;
        reinterpret_cast<float4 *>(out + (size_t)n * C + h * 64)[sub] = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    }
}

int launch_attnpool_attend(hipStream_t st, const float *q, const float *k, const float *v, float *out, int N, int heads,
                           int T, int C) {
    CAPDEC_CHECK(C == heads * 64 && T <= 256, "attention pool: head_dim must be 64, at most 256 tokens");
    const int total = N * heads;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(attnpool_attend_kernel, dim3((total + 3) / 4), dim3(256), 0, st, q, k, v, out, total, heads, T, C);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

CAPDEC_SAT_ACCESSOR(sat_count_resnet)

}  // namespace capdec
