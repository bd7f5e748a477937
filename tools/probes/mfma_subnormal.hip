// Probe (GPU box): does v_mfma_f32_32x32x16_f16 honour fp16 SUBNORMAL inputs on gfx950, or flush them to zero?
// A = 2^-20 (subnormal in fp16), B = 2^10: each of the 16 products is 2^-10, the sum 2^-6 = 0.015625 -- or 0 if flushed.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(float *out, float av, float bv) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)av; b[i] = (_Float16)bv; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; out[2] = (float)b[0]; }
}
int main() {
    float *d, h[3];
    hipMalloc(&d, 12);
    const float cases[4][2] = {{0x1p-20f, 0x1p10f}, {0x1p-24f, 0x1p14f}, {0x1p-15f, 0x1p-15f}, {0x1p-10f, 0x1p-10f}};
    for (auto &cs : cases) {
        probe<<<1, 64>>>(d, cs[0], cs[1]);
        hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a=%g (as f16 %g) b=%g (as f16 %g): mfma sum of 16 products = %g, expected %g\n", cs[0], h[1], cs[1], h[2], h[0],
               16.0 * cs[0] * cs[1]);
    }
    return 0;
}
