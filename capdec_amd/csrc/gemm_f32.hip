// fp32 MFMA GEMM for gfx950:  C[M,N] = epi( A[M,K] . Bt[N,K]^T )
//
// Every dense projection of the caption path runs here (GPT-2 Conv1D = addmm with W [in,out],
// transposed once at load time to [out,in]; nn.Linear weights are already [out,in]; the tied
// lm_head is wte [vocab, d]).  v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain (no TF32 on
// CDNA4), which is what keeps greedy token ids identical to the fp32 reference.
//
// Tiling: 128x128 block tile, BK = 32, 256 threads = 4 wavefronts (2x2), each wavefront owns a
// 64x64 sub-tile = 2x2 MFMA 32x32 accumulators (64 accumulator registers).  Both operands are
// k-contiguous, staged through LDS as [128][36] fp32 (row stride 36 dwords: 16-byte aligned
// and conflict-free for ds_read_b128 by 16-lane groups).  The two half-waves of an MFMA
// 32x32x2 supply k and k+1; we let half h own k in [16h, 16h+16) of the BK tile (a k
// permutation applied to A and B alike), so each lane fetches its 16 k-values of a row with
// four ds_read_b128.  Register-staged double buffering: global loads of tile t+1 are issued
// before the 64 MFMAs of tile t and written to the other LDS buffer after them, one barrier
// per tile.
#include <cstdlib>

#include "gemm_epilogue.h"

namespace capdec {

// Shared main loop: accumulates the 128x128 tile at (m0, n0) into per-wave accumulators.
// BK = k-depth of one LDS stage (32: 72 KB LDS, 2 blocks/CU; 16: 40 KB, 3 blocks/CU).
template <int BK>
struct Tile {
    static constexpr int LD = BK + 4;                  // padded row stride (floats), 16-byte aligned
    static constexpr int TILE = GEMM_BM * LD;          // floats per operand stage
    static constexpr int KQ = BK / 4;                  // float4 per row
    static constexpr int ROWS_PER_PASS = 256 / KQ;
    static constexpr int NP = GEMM_BM / ROWS_PER_PASS; // staging passes per operand
    static constexpr int NQ = BK / 8;                  // ds_read_b128 per row per half-wave
    static constexpr int SMEM_FLOATS = 4 * TILE;
};

template <int BK>
__device__ __forceinline__ void gemm_mainloop(const float *__restrict__ A, int lda, const float *__restrict__ Bt,
                                              int ldb, int M, int N, int K, int m0, int n0, float *smem,
                                              f32x16 (&acc)[2][2]) {
    using TL = Tile<BK>;
    constexpr int LD = TL::LD, TILE = TL::TILE, NP = TL::NP, NQ = TL::NQ;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l32 = lane & 31;

    float *As = smem;               // [2][128][LD]
    float *Bs = smem + 2 * TILE;    // [2][128][LD]

    // global staging: NP float4 per thread per operand.  Rows past M (N) are clamped to the last
    // valid row: they only feed C rows (columns) that are never stored, so no predication is
    // needed and every load is a plain global_load_dwordx4.
    const int srow = t / TL::KQ, skq = (t % TL::KQ) * 4;
    // (named scalars, not arrays: arrays assigned under `if (more)` end up in scratch memory)
    constexpr int RP = TL::ROWS_PER_PASS;
    const float *ap0 = A + (size_t)min(m0 + srow, M - 1) * lda + skq;
    const float *ap1 = A + (size_t)min(m0 + srow + RP, M - 1) * lda + skq;
    const float *ap2 = A + (size_t)min(m0 + srow + 2 * RP, M - 1) * lda + skq;
    const float *ap3 = A + (size_t)min(m0 + srow + 3 * RP, M - 1) * lda + skq;
    const float *bp0 = Bt + (size_t)min(n0 + srow, N - 1) * ldb + skq;
    const float *bp1 = Bt + (size_t)min(n0 + srow + RP, N - 1) * ldb + skq;
    const float *bp2 = Bt + (size_t)min(n0 + srow + 2 * RP, N - 1) * ldb + skq;
    const float *bp3 = Bt + (size_t)min(n0 + srow + 3 * RP, N - 1) * ldb + skq;
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    ra2 = ra3 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
#define GLOAD(k0)                                                          \
    ra0 = *reinterpret_cast<const float4 *>(ap0 + (k0));                   \
    rb0 = *reinterpret_cast<const float4 *>(bp0 + (k0));                   \
    ra1 = *reinterpret_cast<const float4 *>(ap1 + (k0));                   \
    rb1 = *reinterpret_cast<const float4 *>(bp1 + (k0));                   \
    if constexpr (NP > 2) {                                                \
        ra2 = *reinterpret_cast<const float4 *>(ap2 + (k0));               \
        rb2 = *reinterpret_cast<const float4 *>(bp2 + (k0));               \
        ra3 = *reinterpret_cast<const float4 *>(ap3 + (k0));               \
        rb3 = *reinterpret_cast<const float4 *>(bp3 + (k0));               \
    }
    const int st_off = srow * LD + skq;
#define LSTORE(buf)                                                                           \
    *reinterpret_cast<float4 *>(As + (buf) * TILE + st_off) = ra0;                            \
    *reinterpret_cast<float4 *>(Bs + (buf) * TILE + st_off) = rb0;                            \
    *reinterpret_cast<float4 *>(As + (buf) * TILE + st_off + RP * LD) = ra1;                  \
    *reinterpret_cast<float4 *>(Bs + (buf) * TILE + st_off + RP * LD) = rb1;                  \
    if constexpr (NP > 2) {                                                                   \
        *reinterpret_cast<float4 *>(As + (buf) * TILE + st_off + 2 * RP * LD) = ra2;          \
        *reinterpret_cast<float4 *>(Bs + (buf) * TILE + st_off + 2 * RP * LD) = rb2;          \
        *reinterpret_cast<float4 *>(As + (buf) * TILE + st_off + 3 * RP * LD) = ra3;          \
        *reinterpret_cast<float4 *>(Bs + (buf) * TILE + st_off + 3 * RP * LD) = rb3;          \
    }
    static_assert(NP == 2 || NP == 4, "staging passes");

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / BK;
    GLOAD(0)
    LSTORE(0)
    __syncthreads();

    const int a_off = (wm * 64 + l32) * LD + (BK / 2) * half;
    const int b_off = (wn * 64 + l32) * LD + (BK / 2) * half;

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) { GLOAD((kt + 1) * BK) }
        const float *Ab = As + buf * TILE + a_off;
        const float *Bb = Bs + buf * TILE + b_off;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float4 fa0 = *reinterpret_cast<const float4 *>(Ab + 4 * q);
            const float4 fa1 = *reinterpret_cast<const float4 *>(Ab + 32 * LD + 4 * q);
            const float4 fb0 = *reinterpret_cast<const float4 *>(Bb + 4 * q);
            const float4 fb1 = *reinterpret_cast<const float4 *>(Bb + 32 * LD + 4 * q);
#define MFMA4(c)                                                                          \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.c, fb0.c, acc[0][0], 0, 0, 0);    \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.c, fb1.c, acc[0][1], 0, 0, 0);    \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1.c, fb0.c, acc[1][0], 0, 0, 0);    \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1.c, fb1.c, acc[1][1], 0, 0, 0);
            MFMA4(x) MFMA4(y) MFMA4(z) MFMA4(w)
#undef MFMA4
        }
        if (more) { LSTORE(buf ^ 1) }
        __syncthreads();
    }
#undef GLOAD
#undef LSTORE
}

template <int BK, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_f32_kernel(const float *__restrict__ A, int lda,
                                                          const float *__restrict__ Bt, int ldb, float *C, int ldc,
                                                          int M, int N, int K, const float *__restrict__ bias,
                                                          const float *resid, int ldr, int act, int tiles_m,
                                                          int tiles_n) {
    __shared__ __attribute__((aligned(16))) float smem[Tile<BK>::SMEM_FLOATS];
    int tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn);
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
    f32x16 acc[2][2];
    gemm_mainloop<BK>(A, lda, Bt, ldb, M, N, K, m0, n0, smem, acc);

    epilogue_store(acc, C, ldc, M, N, m0, n0, bias, resid, ldr, act);
}

// lm_head variant: per (row, 128-col tile) max, sum exp(x - max) and top-k (value, column).
// The logits tile goes accumulators -> LDS (reusing the staging buffers) -> 16-lane groups,
// one row per group, 8 columns per lane; nothing but ~ (2 + 2k) words per (row, tile) reaches HBM.
template <int KSEL, int BK, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_f32_topk_kernel(const float *__restrict__ A, int lda,
                                                                 const float *__restrict__ Bt, int ldb, int M, int N,
                                                                 int K, float inv_temp, float *tile_max,
                                                                 float *tile_sum, float *cand_val, int *cand_idx,
                                                                 int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) float smem[Tile<BK>::SMEM_FLOATS];
    static_assert(64 * CT_LD <= Tile<BK>::SMEM_FLOATS, "half an epilogue tile must fit the staging buffers");
    int tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn);
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
    f32x16 acc[2][2];
    gemm_mainloop<BK>(A, lda, Bt, ldb, M, N, K, m0, n0, smem, acc);   // ends with a barrier

    epilogue_topk<KSEL>(acc, smem, M, N, m0, n0, tn, tiles_n, inv_temp, tile_max, tile_sum, cand_val, cand_idx);
}

int launch_gemm_f32(hipStream_t st, const float *A, int lda, const float *Bt, int ldb, float *C, int ldc, int M,
                    int N, int K, const GemmEpilogue &epi) {
    CAPDEC_CHECK(M > 0 && N > 0 && K > 0, "gemm: empty problem");
    CAPDEC_CHECK(K % GEMM_BK == 0, "gemm: K must be a multiple of 32");
    CAPDEC_CHECK(lda % 4 == 0 && ldb % 4 == 0, "gemm: lda/ldb must be multiples of 4 (16-byte rows)");
    CAPDEC_CHECK((((uintptr_t)A | (uintptr_t)Bt) & 15) == 0, "gemm: operands must be 16-byte aligned");
    const int tiles_m = (M + GEMM_BM - 1) / GEMM_BM, tiles_n = (N + GEMM_BN - 1) / GEMM_BN;
    // BK 16 (40 KB LDS, 3-4 blocks per CU) measured faster on the wide / deep projections, BK 32 on the
    // N = K = 768 ones (measurement builds: CAPDEC_GEMM_BK overrides)
    const int bk_env = tuning_of(epi).f32_bk;
    const int bk = bk_env ? bk_env : ((N >= 3072 || K >= 2048) ? 16 : 32);
    if (bk == 16)
        hipLaunchKernelGGL((gemm_f32_kernel<16, 3>), dim3(tiles_m * tiles_n), dim3(256), 0, st, A, lda, Bt, ldb, C, ldc, M,
                           N, K, epi.bias, epi.resid, epi.ldr, epi.act, tiles_m, tiles_n);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<32, 2>), dim3(tiles_m * tiles_n), dim3(256), 0, st, A, lda, Bt, ldb, C, ldc, M,
                           N, K, epi.bias, epi.resid, epi.ldr, epi.act, tiles_m, tiles_n);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

int launch_gemm_f32_topk(hipStream_t st, const float *A, int lda, const float *Bt, int ldb, int M, int N, int K,
                         int k, float inv_temp, float *tile_max, float *tile_sum, float *cand_val, int *cand_idx,
                         const Tuning *tune) {
    CAPDEC_CHECK(M > 0 && N > 0 && K > 0, "gemm_topk: empty problem");
    CAPDEC_CHECK(K % GEMM_BK == 0, "gemm_topk: K must be a multiple of 32");
    CAPDEC_CHECK(lda % 4 == 0 && ldb % 4 == 0, "gemm_topk: lda/ldb must be multiples of 4");
    const int tiles_m = (M + GEMM_BM - 1) / GEMM_BM, tiles_n = (N + GEMM_BN - 1) / GEMM_BN;
    dim3 grid(tiles_m * tiles_n), block(256);
    const int bk = (tune ? *tune : default_tuning()).f32_lmhead_bk;
#define LAUNCH_TOPK(KS)                                                                                          \
    if (bk == 16)                                                                                                \
        hipLaunchKernelGGL((gemm_f32_topk_kernel<KS, 16, 3>), grid, block, 0, st, A, lda, Bt, ldb, M, N, K, inv_temp, \
                           tile_max, tile_sum, cand_val, cand_idx, tiles_m, tiles_n);                            \
    else                                                                                                         \
        hipLaunchKernelGGL((gemm_f32_topk_kernel<KS, 32, 2>), grid, block, 0, st, A, lda, Bt, ldb, M, N, K, inv_temp, \
                           tile_max, tile_sum, cand_val, cand_idx, tiles_m, tiles_n)
    switch (k) {
        case 1: LAUNCH_TOPK(1); break;
        case 2: LAUNCH_TOPK(2); break;
        case 3: LAUNCH_TOPK(3); break;
        case 4: LAUNCH_TOPK(4); break;
        case 5: LAUNCH_TOPK(5); break;
        case 6: LAUNCH_TOPK(6); break;
        case 7: LAUNCH_TOPK(7); break;
        case 8: LAUNCH_TOPK(8); break;
        default: CAPDEC_CHECK(false, "gemm_topk: k must be in 1..8");
    }
#undef LAUNCH_TOPK
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

}  // namespace capdec
