// Split-bf16 ("bf16x3") operand format shared by the GEMM and by the kernels that produce its operands.
//
// An fp32 value is exactly a1 + a2 + a3 with three bf16 (8 + 8 + 8 mantissa bits, round-to-nearest at each
// step).  A matrix [R, K] (rows = GEMM M or N index, K contiguous) is stored as three bf16 planes in
// TILE-MAJOR order  P[row_tile][k_step][plane][128 rows][16 bf16]  -- one (row_tile, k_step) block is 12 KB of
// contiguous memory and is byte-for-byte the LDS image of one GEMM stage: rows of 32 B whose two 16-B halves
// are swapped when bit 3 of the row is set (conflict-free ds_read_b128 without padding).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace capdec {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int X3_BK = 16;            // k per step = one v_mfma_f32_32x32x16_bf16
constexpr int X3_ROW_B = 32;         // bytes per row per plane per k-step
constexpr int X3_TILE_ROWS = 128;
constexpr int X3_PLANE_B = X3_TILE_ROWS * X3_ROW_B;   // 4096
constexpr int X3_BLOCK_B = 3 * X3_PLANE_B;            // 12288: one (row_tile, k_step) block

__device__ __forceinline__ void split3(const float4 v, bf16x4 &h, bf16x4 &m, bf16x4 &l) {
    const float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 hh = (__bf16)a[e];
        const float r1 = a[e] - (float)hh;            // exact
        const __bf16 mm = (__bf16)r1;
        const float r2 = r1 - (float)mm;              // exact
        h[e] = hh;
        m[e] = mm;
        l[e] = (__bf16)r2;
    }
}

// byte offset of the 8-byte group holding k = 4*quad .. 4*quad+3 (quad 0..3 within the k-step) of row r (0..127)
__device__ __forceinline__ int x3_group_offset(int r, int quad) {
    return r * X3_ROW_B + (((quad >> 1) ^ ((r >> 3) & 1)) << 4) + ((quad & 1) << 3);
}

// ---- second packed format, "f16x2" (PK_F16X2): TWO fp16 planes per value,
//     a = hi + lo * 2^-11,   hi = fp16(a) (RNE; 0 when |a| < 2^-14, fp16's smallest normal),   lo = fp16((a - hi) * 2^11)
// (11 + 11 significand bits; the low plane is stored scaled by 2^11 so it stays in fp16's normal range wherever it
// matters -- nothing depends on fp16 subnormals: an element below 2^-14 is carried by the low plane alone with an
// absolute error <= 2^-26, an element above it has a relative error <= 2^-23).  A product then needs THREE fp16 MFMAs instead of six:
//     a b = hi_a hi_b + 2^-11 (hi_a lo_b + lo_a hi_b) + 2^-22 lo_a lo_b          (last term dropped: <= 2^-22 |a b|,
// rms 2^-24.6 |a b|, i.e. below the rounding of an fp32 multiply-add), with the 2^-11 terms summed in their own
// accumulator.  Same tile-major layout as above with two planes: one (row_tile, k_step) block is 8 KB.
// |a| is clamped to fp16's largest finite value 65504 (GEMM inputs of this path -- LayerNorm outputs, attention
// outputs, GELU outputs, weights -- are orders of magnitude below it); every clamp is counted (g_h2_saturated below).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
// PK_BF16X1 / PK_F16X1: ONE plane -- the operand rounded to bf16 / fp16 (RNE; fp16 clamped to +-65504): the reduced-
// precision modes (bf16: BASELINE configs[1]; fp16: the arithmetic of the reference's CLIP towers on a GPU), one MFMA per
// product, 4 KB per (row_tile, k_step) block.
enum PackFmt { PK_BF16X3 = 0, PK_F16X2 = 1, PK_BF16X1 = 2, PK_F16X1 = 3 };
constexpr int H2_BLOCK_B = 2 * X3_PLANE_B;            // 8192: one (row_tile, k_step) block of the f16x2 format
constexpr float H2_LO_SCALE = 2048.0f;                // 2^11

// Range accounting of the fp16-plane formats: operands beyond fp16's largest finite value are clamped to +-65504 and
// COUNTED (one count per saturated quad) in a per-translation-unit device counter that capdec_decode_counters sums and
// resets -- a checkpoint whose activations leave the format's range is reported instead of silently saturating.  NaN is
// not clamped: it goes into the high plane and propagates through the MFMAs like it would through an fp32 GEMM.
static __device__ unsigned int g_h2_saturated;
#define CAPDEC_SAT_ACCESSOR(fn)                                                                            \
    unsigned long long fn(bool reset) {                                                                    \
        unsigned int v = 0;                                                                                \
        if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_h2_saturated), sizeof(v)) != hipSuccess) return 0;        \
        if (reset && v) { const unsigned int z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_h2_saturated), &z, sizeof(z)); } \
        return v;                                                                                          \
    }
__device__ __forceinline__ float h2_clamp_count(const float4 v) {
    const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)),
                                    __builtin_fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w)));
    if (m > 65504.f) atomicAdd(&g_h2_saturated, 1u);      // rare by construction: one predicated atomic
    return m;
}
__device__ __forceinline__ float h2_clamp(float a) {       // NaN-preserving (fmaxf(NaN, x) would return x)
    return a != a ? a : __builtin_fminf(__builtin_fmaxf(a, -65504.f), 65504.f);
}

__device__ __forceinline__ void split2h(const float4 v, f16x4 &h, f16x4 &l) {
    const float a[4] = {v.x, v.y, v.z, v.w};
    (void)h2_clamp_count(v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float c = h2_clamp(a[e]);
        const _Float16 hh = __builtin_fabsf(c) < 0x1p-14f ? (_Float16)0.f : (_Float16)c;
        const float r = c - (float)hh;                // exact
        h[e] = hh;
        l[e] = (_Float16)(r * H2_LO_SCALE);
    }
}

__host__ __device__ __forceinline__ int pk_planes(int fmt) {
    return fmt == PK_F16X2 ? 2 : (fmt == PK_BF16X1 || fmt == PK_F16X1) ? 1 : 3;
}

// store the split of 4 consecutive k (one float4) of matrix row `row`, k-step `ks`, quad `quad`
// (fmt is uniform across the launch: PK_BF16X3 = three bf16 planes, PK_F16X2 = two fp16 planes)
__device__ __forceinline__ void x3_store_quad(char *packed, int nk, int row, int ks, int quad, const float4 v,
                                              int fmt = PK_BF16X3) {
    if (fmt == PK_BF16X1 || fmt == PK_F16X1) {
        char *p = packed + ((size_t)(row >> 7) * nk + ks) * X3_PLANE_B + x3_group_offset(row & 127, quad);
        if (fmt == PK_BF16X1) {
            bf16x4 h;
            h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
            *reinterpret_cast<bf16x4 *>(p) = h;
        } else {
            f16x4 h;
            (void)h2_clamp_count(v);
            h[0] = (_Float16)h2_clamp(v.x);
            h[1] = (_Float16)h2_clamp(v.y);
            h[2] = (_Float16)h2_clamp(v.z);
            h[3] = (_Float16)h2_clamp(v.w);
            *reinterpret_cast<f16x4 *>(p) = h;
        }
        return;
    }
    if (fmt == PK_F16X2) {
        f16x4 h, l;
        split2h(v, h, l);
        char *p = packed + ((size_t)(row >> 7) * nk + ks) * H2_BLOCK_B + x3_group_offset(row & 127, quad);
        *reinterpret_cast<f16x4 *>(p) = h;
        *reinterpret_cast<f16x4 *>(p + X3_PLANE_B) = l;
        return;
    }
    bf16x4 h, m, l;
    split3(v, h, m, l);
    char *p = packed + ((size_t)(row >> 7) * nk + ks) * X3_BLOCK_B + x3_group_offset(row & 127, quad);
    *reinterpret_cast<bf16x4 *>(p) = h;
    *reinterpret_cast<bf16x4 *>(p + X3_PLANE_B) = m;
    *reinterpret_cast<bf16x4 *>(p + 2 * X3_PLANE_B) = l;
}

// the inverse for the fp16 / bf16 formats (PK_F16X2, PK_F16X1, PK_BF16X1): the four values of one quad, rejoined
__device__ __forceinline__ float4 x3_load_quad(const char *packed, int nk, int row, int ks, int quad, int fmt) {
    if (fmt == PK_F16X2) {
        const char *p = packed + ((size_t)(row >> 7) * nk + ks) * H2_BLOCK_B + x3_group_offset(row & 127, quad);
        const f16x4 h = *reinterpret_cast<const f16x4 *>(p), l = *reinterpret_cast<const f16x4 *>(p + X3_PLANE_B);
        return make_float4((float)h[0] + (float)l[0] * (1.0f / H2_LO_SCALE), (float)h[1] + (float)l[1] * (1.0f / H2_LO_SCALE),
                           (float)h[2] + (float)l[2] * (1.0f / H2_LO_SCALE), (float)h[3] + (float)l[3] * (1.0f / H2_LO_SCALE));
    }
    const char *p = packed + ((size_t)(row >> 7) * nk + ks) * X3_PLANE_B + x3_group_offset(row & 127, quad);
    if (fmt == PK_F16X1) {
        const f16x4 h = *reinterpret_cast<const f16x4 *>(p);
        return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    }
    const bf16x4 h = *reinterpret_cast<const bf16x4 *>(p);
    return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}

inline size_t x3_packed_bytes(int rows, int K, int fmt = PK_BF16X3) {
    return (size_t)((rows + X3_TILE_ROWS - 1) / X3_TILE_ROWS) * X3_TILE_ROWS * K * pk_planes(fmt) * sizeof(uint16_t);
}

}  // namespace capdec
