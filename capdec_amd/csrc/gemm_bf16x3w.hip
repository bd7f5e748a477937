// EXPERIMENTAL, opt-in (CAPDEC_X3_TILE=256); nothing calls it unless the variable is set.  Status at the end of round 1
// (one short run, the last GPU minute): correct (tests/test_hip_parity.py::test_gemm_packed_a_path passes with the
// variable set: fp64 comparison, bias / activation / residual, ragged M, N % 4 != 0) and 224 TFLOP/s fp32-equivalent on
// 4096^3 (128x128 kernel: 204-208), but on the decode shapes 256-row tiles quantise badly -- M = 25 000 is 98 tile rows
// x 3 / 9 / 12 tile columns on 256 one-block CUs: 129 / 171 / 180 TFLOP/s vs 180 / 182 / 184 -- so it is NOT the default.
// The fc shape (12 tile columns, 92 % grid efficiency, still 180) shows the second cost: one block per CU leaves the
// pipeline fill and the epilogue of a 48-k-step tile exposed (~13 %).  Next: persistent / stream-K scheduling.
//
// Wide-tile variant of the packed-A split-bf16 GEMM (gemm_bf16x3.hip):  C[M,N] = epi( A[M,K] . Bt[N,K]^T ).
// The 128x128 kernel issues one LDS-DMA piece per 4 MFMAs and one fragment read per 2 MFMAs, with one barrier per 24
// MFMAs per wavefront; a DMA piece costs 60-185 issue cycles on CDNA4, and that is where its matrix pipe loses ~40 %.
// Here: 256x256 block tile, 8 wavefronts (2 x 4), each 128 x 64 (4 x 2 accumulators of 32x32) ->
//   per k-step per wavefront 48 MFMAs, 18 fragment reads (0.375 per MFMA), 6 DMA pieces (0.125 per MFMA),
//   half the L2 bytes per MFMA; stage = A rows 0-127 | A rows 128-255 | B cols 0-127 | B cols 128-255 (4 x 12 KB
//   packed blocks = 48 KB), ring of 3 stages = 144 KB dynamic LDS, one block per CU.
// Registers: 128 accumulator + 72 fragment VGPRs, so fragments cannot be double buffered; they are ROLLED instead: the
// six products are issued in the order (0,1) (0,0) (1,0) (1,1) | (0,2) (2,0) and a fragment register is reloaded with
// the NEXT tile's data right after its last use (A1,B1 after the 4th product, A0,B2 after the 5th, A2,B0 after the
// 6th), each at least 8 MFMAs before it is needed again.  The wait + barrier that makes tile k+1 visible (and frees
// the stage of tile k for a later DMA) therefore sits after the 4th product, not at the end of the k-step; the DMA of tile
// k+2 is issued at the top of k-step k, into the stage tile k-1 vacated at the previous barrier.
#include <cstdlib>

#include "bf16x3.h"
#include "gemm_epilogue.h"

namespace capdec {

typedef __attribute__((address_space(3))) void lds_void_w;
typedef const __attribute__((address_space(1))) void glb_void_w;

constexpr int XW_BM = 256, XW_BN = 256;
constexpr int XW_STAGE_B = 4 * X3_BLOCK_B;            // 48 KB
constexpr int XW_STAGES = 3;
constexpr int XW_SMEM_B = XW_STAGES * XW_STAGE_B;     // 144 KB (dynamic)

template <bool VEC4>
__global__ __launch_bounds__(512, 1) void gemm_bf16x3w_kernel(const char *__restrict__ Apk, const char *__restrict__ Bpk,
                                                              float *C, int ldc, int M, int N, int K,
                                                              const float *__restrict__ bias, const float *resid,
                                                              int ldr, int act, int tiles_m, int tiles_n,
                                                              char *packed_out) {
    extern __shared__ __attribute__((aligned(16))) char smem_w[];
    int tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int half = lane >> 5, l32 = lane & 31;
    const int nk = K / X3_BK;                                                   // even (K % 64 == 0)
    const int rt_m = (M + X3_TILE_ROWS - 1) / X3_TILE_ROWS, rt_n = (N + X3_TILE_ROWS - 1) / X3_TILE_ROWS;

    // ---- DMA: 3072 pieces of 16 B per stage, six per thread; piece q = t + 512 j lies in packed block q / 768
    // (0, 1: the two 128-row tiles of A; 2, 3: of B); a wavefront's 64 pieces never straddle a block (768 = 12 x 64)
    const char *src[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int q = t + 512 * j, blk = q / 768, within = q - blk * 768;
        const int rt = (blk < 2) ? min(2 * tm + blk, rt_m - 1) : min(2 * tn + (blk - 2), rt_n - 1);   // clamp odd tails
        src[j] = ((blk < 2) ? Apk : Bpk) + (size_t)rt * nk * X3_BLOCK_B + (size_t)within * 16;
    }
    char *dst0 = smem_w + wave * 1024;                                          // wave-uniform LDS base, + 8 KB per j
#define XW_DMA(stage, ks)                                                                                  \
    _Pragma("unroll") for (int j = 0; j < 6; ++j)                                                          \
        __builtin_amdgcn_global_load_lds((glb_void_w *)(src[j] + (size_t)(ks) * X3_BLOCK_B),               \
                                         (lds_void_w *)(dst0 + (stage) * XW_STAGE_B + j * 8192), 16, 0, 0);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int swz = ((half ^ ((l32 >> 3) & 1)) << 4);
    const int a_rd = wm * X3_BLOCK_B + l32 * X3_ROW_B + swz;                                    // + p 4096 + i 1024
    const int b_rd = (2 + (wn >> 1)) * X3_BLOCK_B + ((wn & 1) * 64 + l32) * X3_ROW_B + swz;     // + p 4096 + j 1024
    bf16x8 fa[3][4], fb[3][2];                                                                  // [plane][block]
#define XW_READ_A(p, stage)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                           \
        fa[p][i] = *reinterpret_cast<const bf16x8 *>(smem_w + (stage) * XW_STAGE_B + a_rd + (p) * X3_PLANE_B + i * 1024);
#define XW_READ_B(p, stage)                                                                                 \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                           \
        fb[p][j] = *reinterpret_cast<const bf16x8 *>(smem_w + (stage) * XW_STAGE_B + b_rd + (p) * X3_PLANE_B + j * 1024);
    // transposed accumulators (operands swapped), as in the 128x128 kernel
#define XW_TERM(pa, pb)                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                           \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                       \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[pb][j], fa[pa][i], acc[i][j], 0, 0, 0);
#define XW_SYNC()                                                                       \
    asm volatile("" ::: "memory");                                                      \
    __builtin_amdgcn_s_waitcnt(0x0076); /* vmcnt(6) lgkmcnt(0) */                       \
    __builtin_amdgcn_s_barrier();                                                       \
    asm volatile("" ::: "memory");

    XW_DMA(0, 0)
    XW_DMA(1, min(1, nk - 1))
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F76);                 // vmcnt(6): tile 0 landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    XW_READ_A(0, 0) XW_READ_A(1, 0) XW_READ_A(2, 0)
    XW_READ_B(0, 0) XW_READ_B(1, 0) XW_READ_B(2, 0)

    // k-step kt:  [DMA of tile kt+2 into the stage tile kt-1 left (freed by the previous barrier) + products 1-4 of
    // tile kt]  SYNC (tile kt+1 landed, everybody done with tile kt's stage)  [products 5, 6 with the rolled reloads
    // from tile kt+1 interleaved].  The LDS-DMA pieces are kept out of the reload region: hipcc orders a ds_read
    // behind every LDS-DMA that precedes it in program order.
    int s0 = 0;                                         // kt % 3
    for (int kt = 0; kt < nk; ++kt) {
        const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s1 == 2 ? 0 : s1 + 1;
        XW_DMA(s2, min(kt + 2, nk - 1))                 // unconditional (clamped): exact vmcnt bookkeeping
        XW_TERM(0, 1) XW_TERM(0, 0) XW_TERM(1, 0) XW_TERM(1, 1)          // 32 MFMAs on tile kt
#pragma unroll
        for (int i_ = 0; i_ < 6; ++i_) {                // one DMA piece in the shadow of each of the first MFMAs
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 26, 0);
        __builtin_amdgcn_sched_barrier(0);
        XW_SYNC()
        XW_READ_A(1, s1) XW_READ_B(1, s1)               // rolled reload: planes whose last use is behind us
        XW_TERM(0, 2)
        XW_READ_A(0, s1) XW_READ_B(2, s1)
        XW_TERM(2, 0)
        XW_READ_A(2, s1) XW_READ_B(0, s1)
#pragma unroll
        for (int i_ = 0; i_ < 6; ++i_) {                // A1', B1' under product 5
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
        for (int i_ = 0; i_ < 6; ++i_) {                // A0', B2' under product 6
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);   // A2', B0': first needed 8 / 40 MFMAs into the next k-step
        __builtin_amdgcn_sched_barrier(0);
        s0 = s1;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0070);                 // vmcnt(0) lgkmcnt(0): clamped tail pieces, stray reloads
#undef XW_DMA
#undef XW_READ_A
#undef XW_READ_B
#undef XW_TERM
#undef XW_SYNC

    // ---- epilogue (transposed layout): acc[i][j][r] = C[m0 + wm 128 + i 32 + l32][n0 + wn 64 + j 32 + 8 (r >> 2) + 4 half + (r & 3)]
    const int m0 = tm * XW_BM, n0 = tn * XW_BN;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + wm * 128 + i * 32 + l32;
        if (row >= M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = n0 + wn * 64 + j * 32 + 8 * g + 4 * half;
                if (col >= N) continue;
                float4 v = acc_quad(acc[i][j], g);
                if (VEC4 || packed_out) {
                    if (bias) {
                        const float4 b = *reinterpret_cast<const float4 *>(bias + col);
                        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                    }
                    v.x = act_apply(v.x, act); v.y = act_apply(v.y, act);
                    v.z = act_apply(v.z, act); v.w = act_apply(v.w, act);
                    if (packed_out) {
                        x3_store_quad(packed_out, N >> 4, row, col >> 4, (col >> 2) & 3, v);
                        continue;
                    }
                    if (resid) {
                        const float4 r4 = *reinterpret_cast<const float4 *>(resid + (size_t)row * ldr + col);
                        v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
                    }
                    *reinterpret_cast<float4 *>(C + (size_t)row * ldc + col) = v;
                } else {
                    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (col + q < N) {
                            float x = act_apply(e[q] + (bias ? bias[col + q] : 0.f), act);
                            if (resid) x += resid[(size_t)row * ldr + col + q];
                            C[(size_t)row * ldc + col + q] = x;
                        }
                }
            }
    }
}

bool gemm_bf16x3w_enabled() {
    static const bool on = [] { const char *e = getenv("CAPDEC_X3_TILE"); return e && atoi(e) == 256; }();
    return on;
}

int launch_gemm_bf16x3w(hipStream_t st, const void *Apacked, const void *Bpacked, float *C, int ldc, int M, int N, int K,
                        const GemmEpilogue &epi) {
    CAPDEC_CHECK(M > 0 && N > 0 && K > 0 && K % 64 == 0, "gemm_bf16x3w: K must be a multiple of 64");
    CAPDEC_CHECK(epi.packed_out == nullptr ||
                     (N % 64 == 0 && epi.resid == nullptr && ((uintptr_t)epi.bias & 15) == 0),
                 "gemm_bf16x3w: packed output needs N % 64 == 0, a 16-byte aligned bias and no residual");
    static const int once = [] {
        int rc = 0;
        rc |= hipFuncSetAttribute((const void *)gemm_bf16x3w_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  XW_SMEM_B) != hipSuccess;
        rc |= hipFuncSetAttribute((const void *)gemm_bf16x3w_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  XW_SMEM_B) != hipSuccess;
        return rc;
    }();
    CAPDEC_CHECK(once == 0, "gemm_bf16x3w: cannot reserve 144 KB of LDS");
    const int tiles_m = (M + XW_BM - 1) / XW_BM, tiles_n = (N + XW_BN - 1) / XW_BN;
    const bool vec4 = N % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)C & 15) == 0 &&
                      (epi.bias == nullptr || ((uintptr_t)epi.bias & 15) == 0) &&
                      (epi.resid == nullptr || (epi.ldr % 4 == 0 && ((uintptr_t)epi.resid & 15) == 0));
    if (vec4)
        hipLaunchKernelGGL(gemm_bf16x3w_kernel<true>, dim3(tiles_m * tiles_n), dim3(512), XW_SMEM_B, st,
                           (const char *)Apacked, (const char *)Bpacked, C, ldc, M, N, K, epi.bias, epi.resid, epi.ldr,
                           epi.act, tiles_m, tiles_n, (char *)epi.packed_out);
    else
        hipLaunchKernelGGL(gemm_bf16x3w_kernel<false>, dim3(tiles_m * tiles_n), dim3(512), XW_SMEM_B, st,
                           (const char *)Apacked, (const char *)Bpacked, C, ldc, M, N, K, epi.bias, epi.resid, epi.ldr,
                           epi.act, tiles_m, tiles_n, (char *)epi.packed_out);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

}  // namespace capdec
