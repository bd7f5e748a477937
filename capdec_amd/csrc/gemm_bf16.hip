// Reduced-precision GEMM modes:  C[M,N] = epi( r(A)[M,K] . r(Bt)[N,K]^T ),  r = round to bf16 or fp16 (RNE), fp32 accumulate
//
// bf16 (BASELINE configs[1]) and fp16 (the arithmetic of the reference's CLIP towers on a GPU, `clip.load` -> fp16):
// weights and GEMM-input activations rounded to 16 bits, ONE v_mfma_f32_32x32x16_{bf16,f16} per product instead of
// the three / six of the fp32-accurate split modes; the residual stream, LayerNorm and softmax stay fp32.  Operands
// arrive in the ONE-plane packed formats PK_BF16X1 / PK_F16X1 of bf16x3.h (4 KB per (row_tile, k_step) block, same
// row layout as the split formats), written by the same producers (LayerNorm / attention / fc epilogue); the kernel
// takes the block stride as a parameter, so it can also read plane 0 of a split format (the hi plane IS the rounding).
//
// A product needs 6x less MFMA time than in the split mode, so a 16-deep k-step per barrier would be all barrier:
// one stage here is FOUR k-steps (K = 64): 4 x (4 KB A hi-plane + 4 KB B hi-plane) = 32 KB, moved by 8 LDS-DMA
// pieces per thread (each 4 KB plane is contiguous in the packed layout); ring of 4 stages = 128 KB, one block per
// CU, tile 128x128, 4 wavefronts x (2x2) accumulators.  Inside a stage the fragments of sub-step j+1 are read while
// the 4 MFMAs of sub-step j run; the end-of-stage wait + barrier sits BEFORE the last sub-step's MFMAs so the
// first fragments of the next stage are already being read under them.  DMA runs three stages ahead.
#include "bf16x3.h"
#include "gemm_epilogue.h"

namespace capdec {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

constexpr int X1_SUB = 4;                                  // k-steps of 16 per stage
constexpr int X1_SUB_B = 2 * X3_PLANE_B;                   // A hi plane + B hi plane of one k-step = 8 KB
constexpr int X1_STAGE_B = X1_SUB * X1_SUB_B;              // 32 KB
constexpr int X1_STAGES = 4;
constexpr int X1_SMEM_B = X1_STAGES * X1_STAGE_B;          // 128 KB (dynamic LDS)

template <bool TR, bool F16>
__device__ __forceinline__ void x1_mainloop(const char *__restrict__ Apk, const char *__restrict__ Bpk, int K, int tm,
                                            int tn, char *smem, f32x16 (&acc)[2][2], int blk) {
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l32 = lane & 31;
    const int nk = K / X3_BK, ns = nk / X1_SUB;                              // K % 64 == 0
    const char *ap = Apk + (size_t)tm * nk * blk + t * 16;                   // plane 0 = first 4 KB of every block
    const char *bp = Bpk + (size_t)tn * nk * blk + t * 16;
    char *dst0 = smem + wave * 1024;                                         // wave-uniform LDS base of its pieces
#define X1_DMA(buf, s)                                                                                          \
    {                                                                                                           \
        const size_t o_ = (size_t)(s) * X1_SUB * blk;                                                           \
        char *d_ = dst0 + (buf) * X1_STAGE_B;                                                                   \
        _Pragma("unroll") for (int j = 0; j < X1_SUB; ++j) {                                                    \
            __builtin_amdgcn_global_load_lds((glb_void_t *)(ap + o_ + (size_t)j * blk),                         \
                                             (lds_void_t *)(d_ + j * X1_SUB_B), 16, 0, 0);                      \
            __builtin_amdgcn_global_load_lds((glb_void_t *)(bp + o_ + (size_t)j * blk),                         \
                                             (lds_void_t *)(d_ + j * X1_SUB_B + X3_PLANE_B), 16, 0, 0);         \
        }                                                                                                       \
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int swz = ((half ^ ((l32 >> 3) & 1)) << 4);
    const int a_rd = (wm * 64 + l32) * X3_ROW_B + swz;
    const int b_rd = X3_PLANE_B + (wn * 64 + l32) * X3_ROW_B + swz;
    // fragment sets F0 / F1, each TWO sub-steps (8 fragments, 8 MFMAs): a set is read while the 8 MFMAs of the other
    // run, so a fragment has 256 MFMA cycles to arrive
    // (fragments are 16 raw bytes; reinterpreted as bf16x8 or f16x8 at the MFMA)
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    i32x4_t f0a0[2], f0a1[2], f0b0[2], f0b1[2], f1a0[2], f1a1[2], f1b0[2], f1b1[2];
#define X1_READ(F, buf, j0)                                                                  \
    _Pragma("unroll") for (int u_ = 0; u_ < 2; ++u_) {                                       \
        const char *rs = smem + (buf) * X1_STAGE_B + ((j0) + u_) * X1_SUB_B;                 \
        F##a0[u_] = *reinterpret_cast<const i32x4_t *>(rs + a_rd);                           \
        F##a1[u_] = *reinterpret_cast<const i32x4_t *>(rs + a_rd + 32 * X3_ROW_B);           \
        F##b0[u_] = *reinterpret_cast<const i32x4_t *>(rs + b_rd);                           \
        F##b1[u_] = *reinterpret_cast<const i32x4_t *>(rs + b_rd + 32 * X3_ROW_B);           \
    }
#define X1_MM1(x, y, c) (F16 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0) \
                             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0))
#define X1_MM(x, y, c) (TR ? X1_MM1(y, x, c) : X1_MM1(x, y, c))
#define X1_MFMAS(F)                                              \
    _Pragma("unroll") for (int u_ = 0; u_ < 2; ++u_) {           \
        acc[0][0] = X1_MM(F##a0[u_], F##b0[u_], acc[0][0]);      \
        acc[0][1] = X1_MM(F##a0[u_], F##b1[u_], acc[0][1]);      \
        acc[1][0] = X1_MM(F##a1[u_], F##b0[u_], acc[1][0]);      \
        acc[1][1] = X1_MM(F##a1[u_], F##b1[u_], acc[1][1]);      \
    }
    // vmcnt(16) [imm bits 3:0 + 15:14] lgkmcnt(0), expcnt untouched: the stage after this one has landed (the two
    // younger stages, 8 pieces each, may still be in flight) and this wave's fragment reads are complete
#define X1_SYNC()                                     \
    asm volatile("" ::: "memory");                    \
    __builtin_amdgcn_s_waitcnt(0x4070);               \
    __builtin_amdgcn_s_barrier();                     \
    asm volatile("" ::: "memory");

    X1_DMA(0, 0)
    X1_DMA(1, min(1, ns - 1))
    X1_DMA(2, min(2, ns - 1))
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x4F70);               // vmcnt(16): stage 0 landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    X1_READ(f0, 0, 0)
    int b0 = 0;                                       // ring buffer of stage kt
    for (int kt = 0; kt < ns; ++kt) {
        const int b1 = (b0 + 1) & 3, b3 = (b0 + 3) & 3;
        X1_READ(f1, b0, 2)                            // sub-steps 2, 3 of this stage
        X1_DMA(b3, min(kt + 3, ns - 1))               // unconditional (clamped): the vmcnt count stays exact
        X1_MFMAS(f0)                                  // sub-steps 0, 1
        // (LDS-DMA pieces stay behind the fragment reads that precede them in program order: both touch LDS)
#pragma unroll
        for (int i_ = 0; i_ < 4; ++i_) {              // 4 x (MFMA, 2 fragment reads), then 4 x (MFMA, 2 DMA pieces)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
#pragma unroll
        for (int i_ = 0; i_ < 4; ++i_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        X1_SYNC()
        X1_READ(f0, b1, 0)                            // sub-steps 0, 1 of the next stage
        X1_MFMAS(f1)
#pragma unroll
        for (int i_ = 0; i_ < 8; ++i_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        b0 = b1;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0070);               // vmcnt(0) lgkmcnt(0): clamped tail pieces landed, reads done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#undef X1_DMA
#undef X1_READ
#undef X1_MM
#undef X1_MM1
#undef X1_MFMAS
#undef X1_SYNC
}

template <bool VEC4, bool F16>
__global__ __launch_bounds__(256, 1) void gemm_bf16p_kernel(const char *__restrict__ Apk, const char *__restrict__ Bpk,
                                                            float *C, int ldc, int M, int N, int K,
                                                            const float *__restrict__ bias, const float *resid,
                                                            int ldr, int act, int tiles_m, int tiles_n,
                                                            char *packed_out, int blk, int out_fmt) {
    extern __shared__ __attribute__((aligned(16))) char smem_x1[];
    int tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn);
    f32x16 acc[2][2];
    x1_mainloop<true, F16>(Apk, Bpk, K, tm, tn, smem_x1, acc, blk);
    if (packed_out)
        epilogue_store_packed_t(acc, packed_out, N >> 4, M, N, tm * GEMM_BM, tn * GEMM_BN, bias, act, out_fmt);
    else
        epilogue_store_t<VEC4>(acc, C, ldc, M, N, tm * GEMM_BM, tn * GEMM_BN, bias, resid, ldr, act);
}

template <int KSEL, bool F16>
__global__ __launch_bounds__(256, 1) void gemm_bf16p_topk_kernel(const char *__restrict__ Apk,
                                                                 const char *__restrict__ Bpk, int M, int N, int K,
                                                                 float inv_temp, float *tile_max, float *tile_sum,
                                                                 float *cand_val, int *cand_idx, int tiles_m,
                                                                 int tiles_n, int blk) {
    extern __shared__ __attribute__((aligned(16))) char smem_x1[];
    static_assert(128 * CT_LD * 4 <= X1_SMEM_B, "epilogue tile must fit the staging ring");
    int tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn);
    f32x16 acc[2][2];
    x1_mainloop<false, F16>(Apk, Bpk, K, tm, tn, smem_x1, acc, blk);      // ends with a barrier
    epilogue_topk<KSEL, 2, true>(acc, reinterpret_cast<float *>(smem_x1), M, N, tm * GEMM_BM, tn * GEMM_BN, tn, tiles_n,
                           inv_temp, tile_max, tile_sum, cand_val, cand_idx);
}

template <typename F> static int x1_allow_lds(F f) {
    CAPDEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(f), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   X1_SMEM_B));
    return 0;
}

// fmt: the packed format of BOTH operands (PK_BF16X1 / PK_F16X1, or a split format whose plane 0 is read:
// PK_BF16X3 -> bf16, PK_F16X2 -> fp16); packed output (epi.packed_out) is written in the matching one-plane format
int launch_gemm_bf16p(hipStream_t st, const void *Apacked, const void *Bpacked, float *C, int ldc, int M, int N, int K,
                      const GemmEpilogue &epi, int fmt) {
    CAPDEC_CHECK(M > 0 && N > 0 && K > 0 && K % 64 == 0, "gemm_bf16p: K must be a multiple of 64");
    CAPDEC_CHECK(epi.packed_out == nullptr ||
                     (N % 64 == 0 && epi.resid == nullptr && ((uintptr_t)epi.bias & 15) == 0),
                 "gemm_bf16p: packed output needs N % 64 == 0, a 16-byte aligned bias and no residual");
    static const int once = x1_allow_lds(gemm_bf16p_kernel<true, false>) | x1_allow_lds(gemm_bf16p_kernel<false, false>) |
                            x1_allow_lds(gemm_bf16p_kernel<true, true>) | x1_allow_lds(gemm_bf16p_kernel<false, true>);
    CAPDEC_CHECK(once == 0, "gemm_bf16p: cannot reserve 128 KB of LDS");
    const int tiles_m = (M + GEMM_BM - 1) / GEMM_BM, tiles_n = (N + GEMM_BN - 1) / GEMM_BN;
    const bool vec4 = N % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)C & 15) == 0 &&
                      (epi.bias == nullptr || ((uintptr_t)epi.bias & 15) == 0) &&
                      (epi.resid == nullptr || (epi.ldr % 4 == 0 && ((uintptr_t)epi.resid & 15) == 0));
    const bool f16 = fmt == PK_F16X1 || fmt == PK_F16X2;
    const int blk = pk_planes(fmt) * X3_PLANE_B, out_fmt = fmt;
#define LAUNCH_X1(V4, H)                                                                                              \
    hipLaunchKernelGGL((gemm_bf16p_kernel<V4, H>), dim3(tiles_m * tiles_n), dim3(256), X1_SMEM_B, st,                   \
                       (const char *)Apacked, (const char *)Bpacked, C, ldc, M, N, K, epi.bias, epi.resid, epi.ldr,    \
                       epi.act, tiles_m, tiles_n, (char *)epi.packed_out, blk, out_fmt)
    if (vec4) { if (f16) LAUNCH_X1(true, true); else LAUNCH_X1(true, false); }
    else      { if (f16) LAUNCH_X1(false, true); else LAUNCH_X1(false, false); }
#undef LAUNCH_X1
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

int launch_gemm_bf16p_topk(hipStream_t st, const void *Apacked, const void *Bpacked, int M, int N, int K, int k,
                           float inv_temp, float *tile_max, float *tile_sum, float *cand_val, int *cand_idx, int fmt) {
    CAPDEC_CHECK(M > 0 && N > 0 && K > 0 && K % 64 == 0, "gemm_bf16p_topk: K must be a multiple of 64");
    const int tiles_m = (M + GEMM_BM - 1) / GEMM_BM, tiles_n = (N + GEMM_BN - 1) / GEMM_BN;
    dim3 grid(tiles_m * tiles_n), block(256);
    const bool f16 = fmt == PK_F16X1 || fmt == PK_F16X2;
    const int blk = pk_planes(fmt) * X3_PLANE_B;
#define LAUNCH_TOPK1H(KS, H)                                                                                       \
    {                                                                                                              \
        static const int once = x1_allow_lds(gemm_bf16p_topk_kernel<KS, H>);                                       \
        CAPDEC_CHECK(once == 0, "gemm_bf16p_topk: cannot reserve 128 KB of LDS");                                  \
        hipLaunchKernelGGL((gemm_bf16p_topk_kernel<KS, H>), grid, block, X1_SMEM_B, st, (const char *)Apacked,      \
                           (const char *)Bpacked, M, N, K, inv_temp, tile_max, tile_sum, cand_val, cand_idx, tiles_m, \
                           tiles_n, blk);                                                                          \
    }
#define LAUNCH_TOPK1(KS) if (f16) LAUNCH_TOPK1H(KS, true) else LAUNCH_TOPK1H(KS, false)
    switch (k) {
        case 1: LAUNCH_TOPK1(1); break;
        case 2: LAUNCH_TOPK1(2); break;
        case 3: LAUNCH_TOPK1(3); break;
        case 4: LAUNCH_TOPK1(4); break;
        case 5: LAUNCH_TOPK1(5); break;
        case 6: LAUNCH_TOPK1(6); break;
        case 7: LAUNCH_TOPK1(7); break;
        case 8: LAUNCH_TOPK1(8); break;
        default: CAPDEC_CHECK(false, "gemm_topk: k must be in 1..8");
    }
#undef LAUNCH_TOPK1
#undef LAUNCH_TOPK1H
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// generic packer (any PackFmt): fp32 [N, K] (row stride ldw) -> tile-major planes, rows past N zero.  One thread per
// (padded row, quad of 4 k); used for the one-plane formats (the split formats have their own 16-byte-store packers)
__global__ void pack_planes_fmt_kernel(const float *__restrict__ w, int ldw, char *__restrict__ out, int N, int K,
                                       int rows_pad, int fmt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nq = K >> 2;
    if (i >= (size_t)rows_pad * nq) return;
    const int row = (int)(i / nq), qd = (int)(i - (size_t)row * nq);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < N) v = *reinterpret_cast<const float4 *>(w + (size_t)row * ldw + qd * 4);
    x3_store_quad(out, K >> 4, row, qd >> 2, qd & 3, v, fmt);
}

int launch_pack_planes_fmt(hipStream_t st, const float *w, int ldw, int N, int K, void *out, int fmt) {
    CAPDEC_CHECK(K % 64 == 0 && ldw % 4 == 0, "pack_planes: K must be a multiple of 64");
    const int rows_pad = (N + 127) / 128 * 128;
    const size_t tot = (size_t)rows_pad * (K >> 2);
    hipLaunchKernelGGL(pack_planes_fmt_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, w, ldw, (char *)out,
                       N, K, rows_pad, fmt);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

}  // namespace capdec
