// Probe (GPU box): which fp16 MFMA shape delivers more FLOP/s at the package power cap -- v_mfma_f32_32x32x16_f16 or
// v_mfma_f32_16x16x32_f16?  Registers-only loops (no LDS, no memory), every SIMD saturated, random or zero operands,
// ~1.5 s sustained per case.  At the cap the sustained rate IS the energy per FLOP (the clock gives way); with zero
// operands the same loops show the issue-rate peak.  Per MAC 16x16x32 moves half the accumulator bytes and twice the
// operand bytes of 32x32x16 through the register file.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16x8 rnd8(unsigned s, bool zero) {
    f16x8 v;
    for (int i = 0; i < 8; ++i) {
        s = s * 1664525u + 1013904223u;
        v[i] = zero ? (_Float16)0.f : (_Float16)(((int)(s >> 9) & 0xFFFF) / 32768.0f - 1.0f);
    }
    return v;
}

template <bool SMALL>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, int zero) {
    const unsigned seed = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = rnd8(seed + i, zero); b[i] = rnd8(seed + 17 + i, zero); }
    float sink = 0.f;
    if constexpr (!SMALL) {
        f32x16 c[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[(i + 1) & 3], c[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) sink += c[i][0] + c[i][15];
    } else {
        f32x4 c[16];
        for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) c[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i + 1) & 3], c[i], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) sink += c[i][0] + c[i][3];
    }
    if (sink == 123.456f) out[0] = sink;     // keep the loop alive
}

template <bool SMALL>
static void run(const char *name, int zero, float *d) {
    const int blocks = 256 * 8, iters = 20000;                 // 16 waves per CU... 8 blocks x 4 waves / CU-slot
    const double flop_per_launch = (double)blocks * 4 * iters * (SMALL ? 16 * 2.0 * 16 * 16 * 32 : 8 * 2.0 * 32 * 32 * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) mfma_loop<SMALL><<<blocks, 256>>>(d, iters, zero);
    hipDeviceSynchronize();
    int n = 0;
    float ms = 0.f;
    hipEventRecord(e0);
    do {
        mfma_loop<SMALL><<<blocks, 256>>>(d, iters, zero);
        ++n;
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    } while (ms < 1500.f);
    printf("%-28s %-7s %8.1f TFLOP/s (fp16 MFMA executed)   %d launches in %.0f ms\n", name, zero ? "zeros" : "random",
           flop_per_launch * n / (ms * 1e-3) / 1e12, n, ms);
    fflush(stdout);
}

int main() {
    float *d;
    hipMalloc(&d, 64);
    run<false>("v_mfma_f32_32x32x16_f16", 0, d);
    run<true>("v_mfma_f32_16x16x32_f16", 0, d);
    run<false>("v_mfma_f32_32x32x16_f16", 1, d);
    run<true>("v_mfma_f32_16x16x32_f16", 1, d);
    run<false>("v_mfma_f32_32x32x16_f16", 0, d);
    run<true>("v_mfma_f32_16x16x32_f16", 0, d);
    return 0;
}
