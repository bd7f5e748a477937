// fp32-accurate GEMM on the bf16 matrix cores:  C[M,N] = epi( A[M,K] . Bt[N,K]^T )
//
// gfx950's fp32-input MFMA runs at 1/16 of the bf16 MFMA rate (157 vs 2500 TFLOP/s).  Every fp32
// value is exactly a sum of three bf16 values  a = a1 + a2 + a3  (8 + 8 + 8 mantissa bits), so
//     a . b  =  a1 b1 + (a1 b2 + a2 b1) + (a2 b2 + a1 b3 + a3 b1) + O(2^-27 |a b|)
// and six v_mfma_f32_32x32x16_bf16 (products exact, fp32 accumulate) reproduce an fp32 dot product
// to fp32 round-off at 16/6 = 2.7x the throughput of the native fp32 MFMA.  The dropped terms
// (a2 b3 + a3 b2 + a3 b3) are below half an fp32 ulp of the product.
//
// Weights are split once at load time into three bf16 planes, packed tile-major; activations stay fp32 in
// HBM and are split on the fly while they are staged into LDS (6 VALU ops per element, hidden
// under 24 MFMAs per k-step).  Tiling as the fp32 kernel: 128x128 block, 4 wavefronts x (2x2)
// 32x32 accumulators; BK = 16 (one MFMA k-step) per LDS stage, three-stage ring (72 KB, 2 blocks/CU).
#include <cstdlib>

#include "bf16x3.h"
#include "gemm_epilogue.h"

namespace capdec {

constexpr int X3_BPLANE_B = X3_PLANE_B;      // 4096: one B plane of a stage (128 columns)
constexpr int X3_STAGES = 3;
// WMG = 2: 128x128 tile (72 KB LDS, 2 blocks/CU); WMG = 1: 64x128 tile for small M (54 KB, twice the blocks)
template <int WMG> struct X3Geo {
    static constexpr int BM = 64 * WMG;
    static constexpr int APLANE_B = BM * X3_ROW_B;
    static constexpr int AOPER_B = 3 * APLANE_B;
    static constexpr int STAGE_B = AOPER_B + 3 * X3_BPLANE_B;
    static constexpr int SMEM_B = X3_STAGES * STAGE_B > 64 * CT_LD * 4 ? X3_STAGES * STAGE_B : 64 * CT_LD * 4;
};

// Weight pre-pack: fp32 W [N, K] -> bf16 planes in TILE-MAJOR order
//     Bpk[tile_n][k_step][plane][row 0..127][16 bf16]      (rows past N are zero)
// i.e. exactly the LDS image of one B stage (3 planes x 128 rows x 32 B, 16-B halves swapped when bit
// 3 of the row is set), so a block's B tile of one k-step is 12 KB of CONTIGUOUS memory: every
// global load uses whole 128-B lines and the LDS write is a linear copy.
__global__ void pack_planes_kernel(const float *__restrict__ w, __bf16 *__restrict__ out, int N, int K, int nk) {
    // one thread per (tile, k_step, row, half): 8 consecutive k
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int tiles = (N + GEMM_BN - 1) / GEMM_BN;
    if (i >= (size_t)tiles * nk * 256) return;
    const int hk = i & 1, r = (i >> 1) & 127;
    const size_t tk = i >> 8;                       // tile * nk + k_step
    const int ks = (int)(tk % nk), tile = (int)(tk / nk);
    const int n = tile * GEMM_BN + r;
    bf16x4 h0, m0, l0, h1, m1, l1;
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (n < N) {
        const float *src = w + (size_t)n * K + ks * 16 + hk * 8;
        v0 = reinterpret_cast<const float4 *>(src)[0];
        v1 = reinterpret_cast<const float4 *>(src)[1];
    }
    split3(v0, h0, m0, l0);
    split3(v1, h1, m1, l1);
    __bf16 *dst = out + tk * (3 * 128 * 16) + r * 16 + ((hk ^ ((r >> 3) & 1)) << 3);
    reinterpret_cast<bf16x4 *>(dst)[0] = h0;
    reinterpret_cast<bf16x4 *>(dst)[1] = h1;
    reinterpret_cast<bf16x4 *>(dst + 2048)[0] = m0;
    reinterpret_cast<bf16x4 *>(dst + 2048)[1] = m1;
    reinterpret_cast<bf16x4 *>(dst + 4096)[0] = l0;
    reinterpret_cast<bf16x4 *>(dst + 4096)[1] = l1;
}

size_t packed_planes_bytes(int N, int K) { return x3_packed_bytes(N, K); }

int launch_pack_planes(hipStream_t st, const float *w, int N, int K, void *out) {
    CAPDEC_CHECK(K % 64 == 0, "pack_planes: K must be a multiple of 64");
    const int nk = K / X3_BK;
    const size_t tot = (size_t)((N + GEMM_BN - 1) / GEMM_BN) * nk * 256;
    hipLaunchKernelGGL(pack_planes_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, w, (__bf16 *)out, N, K, nk);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// Main loop.  LDS: 3-stage ring, rows of 32 B (16 bf16) whose two 16-B halves are swapped when bit 3
// of the row index is set (conflict-free ds_read_b128 without padding; 3 x 24 KB = 72 KB, 2 blocks/CU).
// Global loads run TWO k-steps ahead of their LDS store (L2 misses served by the Infinity Cache take
// longer than one 24-MFMA k-step), through alternating register sets; the loop is unrolled by 4 so
// every register set has a static name.  In k-step kt a wavefront
//   reads the fragments of tile kt from stage kt%3,
//   issues its 24 MFMAs with the split + LDS store of tile kt+2 (stage (kt+2)%3) interleaved between
//   the six MFMA groups (VALU / LDS work overlaps the matrix pipe inside the wavefront),
//   re-issues the global loads of tile kt+4 into the registers it just drained (unconditionally, with a
//   clamped tile index past the end: a load under a branch would force the compiler to wait vmcnt(0)
//   instead of a counted wait); one barrier per k-step.
// A: thread (row = t>>2 [+64], quad = t&3) owns 4 k of each k-step; an (even, odd) k-step pair is loaded
//    together so the 4 threads of a row fetch one whole 128-B line.  B: packed tile, 12 KB contiguous.
template <int WMG>
__device__ __forceinline__ void x3_mainloop(const float *__restrict__ A, int lda, const __bf16 *__restrict__ Bpk,
                                            int M, int K, int m0, int tn, char *smem,
                                            f32x16 (&acc)[2][WaveGrid<WMG>::NJ]) {
    using G = X3Geo<WMG>;
    constexpr int NJ = WaveGrid<WMG>::NJ;
    constexpr int X3_PLANE_B = G::APLANE_B, X3_OPER_B = G::AOPER_B, X3_STAGE_B = G::STAGE_B;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = WaveGrid<WMG>::wm(wave), wn = WaveGrid<WMG>::wn(wave);
    const int half = lane >> 5, l32 = lane & 31;
    const int nk = K / X3_BK;                     // multiple of 4 (K % 64 == 0)

    const int arow = t >> 2, akq = t & 3;
    const float *ap0 = A + (size_t)min(m0 + arow, M - 1) * lda + akq * 4;
    const float *ap1 = A + (size_t)min(m0 + arow + (WMG == 2 ? 64 : 0), M - 1) * lda + akq * 4;   // second 64 rows (WMG 2)
    const int a_st = arow * X3_ROW_B + (((akq >> 1) ^ ((arow >> 3) & 1)) << 4) + ((akq & 1) << 3);
    const __bf16 *bp = Bpk + (size_t)tn * nk * 6144 + t * 8;
    const int b_st = X3_OPER_B + t * 16;

    // staging registers: A pairs P0 / P1 (even + odd k-step of a pair), B sets B0 / B1
    float4 p0_0e, p0_1e, p0_0o, p0_1o, p1_0e, p1_1e, p1_0o, p1_1o;
    p0_1e = p0_1o = p1_1e = p1_1o = make_float4(0.f, 0.f, 0.f, 0.f);
    uint4 b0_0, b0_1, b0_2, b1_0, b1_1, b1_2;
#define X3_GLOAD_A(P, kp)   /* kp = even k-step of the pair */                       \
    P##_0e = *reinterpret_cast<const float4 *>(ap0 + (kp) * X3_BK);                  \
    P##_0o = *reinterpret_cast<const float4 *>(ap0 + (kp) * X3_BK + X3_BK);          \
    if constexpr (WMG == 2) {                                                        \
        P##_1e = *reinterpret_cast<const float4 *>(ap1 + (kp) * X3_BK);              \
        P##_1o = *reinterpret_cast<const float4 *>(ap1 + (kp) * X3_BK + X3_BK);      \
    }
#define X3_GLOAD_B(S, ks)                                                            \
    S##_0 = *reinterpret_cast<const uint4 *>(bp + (size_t)(ks) * 6144);              \
    S##_1 = *reinterpret_cast<const uint4 *>(bp + (size_t)(ks) * 6144 + 2048);       \
    S##_2 = *reinterpret_cast<const uint4 *>(bp + (size_t)(ks) * 6144 + 4096);
#define X3_STORE_A(sb, RA, off)                                                                   \
    {                                                                                             \
        bf16x4 h, m, l;                                                                           \
        split3(RA, h, m, l);                                                                      \
        *reinterpret_cast<bf16x4 *>((sb) + a_st + (off)) = h;                                     \
        *reinterpret_cast<bf16x4 *>((sb) + X3_PLANE_B + a_st + (off)) = m;                        \
        *reinterpret_cast<bf16x4 *>((sb) + 2 * X3_PLANE_B + a_st + (off)) = l;                    \
    }
#define X3_STORE_B(sb, S)                                                                         \
    *reinterpret_cast<uint4 *>((sb) + b_st) = S##_0;                                              \
    *reinterpret_cast<uint4 *>((sb) + X3_BPLANE_B + b_st) = S##_1;                                \
    *reinterpret_cast<uint4 *>((sb) + 2 * X3_BPLANE_B + b_st) = S##_2;

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int swz = ((half ^ ((l32 >> 3) & 1)) << 4);
    const int a_rd = (wm * 64 + l32) * X3_ROW_B + swz;
    const int b_rd = X3_OPER_B + (wn * 32 * NJ + l32) * X3_ROW_B + swz;

    bf16x8 fa0[3], fa1[3], fb0[3], fb1[3];
    // smallest terms first: (a3 b1), (a1 b3), (a2 b2), (a2 b1), (a1 b2), (a1 b1)
#define X3_TERM(pa, pb)                                                                                \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[pa], fb0[pb], acc[0][0], 0, 0, 0);         \
    if constexpr (NJ == 2) acc[0][NJ - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[pa], fb1[pb], acc[0][NJ - 1], 0, 0, 0); \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[pa], fb0[pb], acc[1][0], 0, 0, 0);         \
    if constexpr (NJ == 2) acc[1][NJ - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[pa], fb1[pb], acc[1][NJ - 1], 0, 0, 0);
    // one k-step: RA0/RA1 = A registers of tile kt+2 (two row passes), SB = its B set, LOADS = re-issue
#define X3_STEP(kt, RA0, RA1, SB, LOADS)                                                           \
    {                                                                                              \
        const char *rs = smem + ((kt) % X3_STAGES) * X3_STAGE_B;                                   \
        char *ws = smem + (((kt) + 2) % X3_STAGES) * X3_STAGE_B;                                   \
        const bool st = (kt) + 2 < nk;                                                             \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                            \
            fa0[p] = *reinterpret_cast<const bf16x8 *>(rs + p * X3_PLANE_B + a_rd);                \
            fa1[p] = *reinterpret_cast<const bf16x8 *>(rs + p * X3_PLANE_B + a_rd + 32 * X3_ROW_B); \
            fb0[p] = *reinterpret_cast<const bf16x8 *>(rs + p * X3_BPLANE_B + b_rd);               \
            if constexpr (NJ == 2) fb1[p] = *reinterpret_cast<const bf16x8 *>(rs + p * X3_BPLANE_B + b_rd + 32 * X3_ROW_B); \
        }                                                                                          \
        X3_TERM(2, 0)                                                                              \
        if (st) X3_STORE_A(ws, RA0, 0)                                                             \
        X3_TERM(0, 2)                                                                              \
        X3_TERM(1, 1)                                                                              \
        if constexpr (WMG == 2) { if (st) X3_STORE_A(ws, RA1, 64 * X3_ROW_B) }                     \
        X3_TERM(1, 0)                                                                              \
        X3_TERM(0, 1)                                                                              \
        if (st) { X3_STORE_B(ws, SB) }                                                             \
        X3_TERM(0, 0)                                                                              \
        LOADS                                                                                      \
        __syncthreads();                                                                           \
    }

    // prologue: tiles 0, 1 staged synchronously; tiles 2, 3 (pair P1, B0, B1) put in flight
    X3_GLOAD_A(p0, 0)
    X3_GLOAD_B(b0, 0)
    X3_GLOAD_B(b1, 1)
    X3_STORE_A(smem, p0_0e, 0)
    if constexpr (WMG == 2) X3_STORE_A(smem, p0_1e, 64 * X3_ROW_B)
    X3_STORE_B(smem, b0)
    X3_STORE_A(smem + X3_STAGE_B, p0_0o, 0)
    if constexpr (WMG == 2) X3_STORE_A(smem + X3_STAGE_B, p0_1o, 64 * X3_ROW_B)
    X3_STORE_B(smem + X3_STAGE_B, b1)
    X3_GLOAD_A(p1, 2)
    X3_GLOAD_B(b0, 2)
    X3_GLOAD_B(b1, 3)
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 4) {
        // tile kt+2 = even half of pair P1; afterwards B0 <- tile kt+4, P0 <- pair (kt+4, kt+5)
        X3_STEP(kt, p1_0e, p1_1e, b0, X3_GLOAD_B(b0, min(kt + 4, nk - 1)) X3_GLOAD_A(p0, min(kt + 4, nk - 2)))
        // tile kt+3 = odd half of P1; B1 <- tile kt+5
        X3_STEP(kt + 1, p1_0o, p1_1o, b1, X3_GLOAD_B(b1, min(kt + 5, nk - 1)))
        // tile kt+4 = even half of P0; B0 <- tile kt+6, P1 <- pair (kt+6, kt+7)
        X3_STEP(kt + 2, p0_0e, p0_1e, b0, X3_GLOAD_B(b0, min(kt + 6, nk - 1)) X3_GLOAD_A(p1, min(kt + 6, nk - 2)))
        // tile kt+5 = odd half of P0; B1 <- tile kt+7
        X3_STEP(kt + 3, p0_0o, p0_1o, b1, X3_GLOAD_B(b1, min(kt + 7, nk - 1)))
    }
#undef X3_GLOAD_A
#undef X3_GLOAD_B
#undef X3_STORE_A
#undef X3_STORE_B
#undef X3_TERM
#undef X3_STEP
}

template <int WMG>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(const float *__restrict__ A, int lda,
                                                             const __bf16 *__restrict__ Bpk, float *C, int ldc,
                                                             int M, int N, int K,
                                                             const float *__restrict__ bias, const float *resid,
                                                             int ldr, int act, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) char smem[X3Geo<WMG>::SMEM_B];
    int tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn);
    const int m0 = tm * X3Geo<WMG>::BM, n0 = tn * GEMM_BN;
    f32x16 acc[2][WaveGrid<WMG>::NJ];
    x3_mainloop<WMG>(A, lda, Bpk, M, K, m0, tn, smem, acc);
    epilogue_store<WMG>(acc, C, ldc, M, N, m0, n0, bias, resid, ldr, act);
}

template <int KSEL, int WMG>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_topk_kernel(const float *__restrict__ A, int lda,
                                                                  const __bf16 *__restrict__ Bpk, int M, int N,
                                                                  int K, float inv_temp,
                                                                  float *tile_max, float *tile_sum, float *cand_val,
                                                                  int *cand_idx, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) char smem[X3Geo<WMG>::SMEM_B];
    int tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn);
    const int m0 = tm * X3Geo<WMG>::BM, n0 = tn * GEMM_BN;
    f32x16 acc[2][WaveGrid<WMG>::NJ];
    x3_mainloop<WMG>(A, lda, Bpk, M, K, m0, tn, smem, acc);   // ends with a barrier
    epilogue_topk<KSEL, WMG>(acc, reinterpret_cast<float *>(smem), M, N, m0, n0, tn, tiles_n, inv_temp, tile_max, tile_sum,
                        cand_val, cand_idx);
}

// ---------------------------------------------------------------------------------------------------------
// Packed-A variant: when the producer of A (LayerNorm, ...) already emits the split, tile-major planes, both
// operands of a k-step are 12 KB contiguous LDS images and the main loop moves them with LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip, no ds_write, no split VALU): per k-step a wavefront issues
// 12 ds_read_b128, 24 MFMAs and 6 DMA pieces (tile kt+2 -> stage (kt+2)%3), and one counted s_waitcnt
// vmcnt(6) in front of a raw s_barrier retires the pieces of tile kt+1.  Every vector-memory operation of the
// loop is a DMA, so the in-order vmcnt count is exact.
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// TR: the MFMA operands are swapped (D^T = B^T-fragment x A-fragment), so a lane ends up with FOUR CONSECUTIVE COLUMNS
// of one output row (row = lane & 31, columns 8 (r >> 2) + 4 (lane >> 5) + (r & 3)) instead of four consecutive
// rows of one column: the epilogue stores float4 / whole split quads straight from the accumulators.
template <bool TR>
__device__ __forceinline__ void x3p_mainloop(const __bf16 *__restrict__ Apk, const __bf16 *__restrict__ Bpk, int K,
                                             int tm, int tn, char *smem, f32x16 (&acc)[2][2], int dbg = 0,
                                             int ks0 = 0, int nks = -1) {
    constexpr int STAGE_B = 2 * X3_BLOCK_B;                  // A block + B block = 24 KB
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l32 = lane & 31;
    const int nkf = K / X3_BK;                                                 // k-steps of the whole K (panel stride)
    const int nk = nks < 0 ? nkf : nks;                                        // k-steps of THIS block: [ks0, ks0 + nk), even
    // dbg (CAPDEC_ABL_DMA, measurement only -- results are wrong): 1 = every block streams the panels of tile (0, 0)
    // (L2-resident: what the kernel does without fabric traffic), 2 = every k-step re-reads k-step 0 of its own panels
    const __bf16 *ap = Apk + ((size_t)(dbg == 1 ? 0 : tm) * nkf + ks0) * (X3_BLOCK_B / 2) + t * 8;   // this thread's 16-B piece
    const __bf16 *bp = Bpk + ((size_t)(dbg == 1 ? 0 : tn) * nkf + ks0) * (X3_BLOCK_B / 2) + t * 8;
    const int ksmask = dbg == 2 ? 0 : -1;
    char *dst0 = smem + wave * 1024;                                           // wave-uniform LDS base of its pieces
#define X3P_DMA(stage, ks_)                                                                                    \
    {                                                                                                          \
        const int ks = (ks_) & ksmask;                                                                         \
        const __bf16 *sa = ap + (size_t)(ks) * (X3_BLOCK_B / 2), *sb = bp + (size_t)(ks) * (X3_BLOCK_B / 2);   \
        char *d = dst0 + (stage) * STAGE_B;                                                                    \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                        \
            __builtin_amdgcn_global_load_lds((glb_void *)(sa + p * 2048), (lds_void *)(d + p * X3_PLANE_B), 16, 0, 0); \
            __builtin_amdgcn_global_load_lds((glb_void *)(sb + p * 2048), (lds_void *)(d + X3_BLOCK_B + p * X3_PLANE_B), 16, 0, 0); \
        }                                                                                                      \
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int swz = ((half ^ ((l32 >> 3) & 1)) << 4);
    const int a_rd = (wm * 64 + l32) * X3_ROW_B + swz;
    const int b_rd = X3_BLOCK_B + (wn * 64 + l32) * X3_ROW_B + swz;

    // Fragment sets F0 / F1: the 12 ds_read_b128 of tile kt+1 are issued at the top of k-step kt and land under its
    // 24 MFMAs (which run from the other set), so no LDS latency sits in front of an MFMA.  Ring of 3 stages:
    // in k-step kt  stage kt%3 (its fragments were read during kt-1) receives tile kt+3 by DMA, stage (kt+1)%3 is
    // read, tile kt+2 is in flight.  End of k-step: vmcnt(6) = tile kt+2 landed (only tile kt+3's 6 pieces pending),
    // lgkmcnt(0) = this wave's fragment reads are complete, then the barrier frees stage (kt+1)%3 for the next DMA.
    bf16x8 f0a0[3], f0a1[3], f0b0[3], f0b1[3], f1a0[3], f1a1[3], f1b0[3], f1b1[3];
#define X3P_READ(F, stage)                                                                          \
    {                                                                                               \
        const char *rs = smem + (stage) * STAGE_B;                                                  \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                             \
            F##a0[p] = *reinterpret_cast<const bf16x8 *>(rs + p * X3_PLANE_B + a_rd);                \
            F##a1[p] = *reinterpret_cast<const bf16x8 *>(rs + p * X3_PLANE_B + a_rd + 32 * X3_ROW_B); \
            F##b0[p] = *reinterpret_cast<const bf16x8 *>(rs + p * X3_PLANE_B + b_rd);                \
            F##b1[p] = *reinterpret_cast<const bf16x8 *>(rs + p * X3_PLANE_B + b_rd + 32 * X3_ROW_B); \
        }                                                                                           \
    }
#define X3P_MM(x, y, c) (TR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, c, 0, 0, 0)                \
                            : __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0))
#define X3P_TERM(F, pa, pb)                                      \
    acc[0][0] = X3P_MM(F##a0[pa], F##b0[pb], acc[0][0]);         \
    acc[0][1] = X3P_MM(F##a0[pa], F##b1[pb], acc[0][1]);         \
    acc[1][0] = X3P_MM(F##a1[pa], F##b0[pb], acc[1][0]);         \
    acc[1][1] = X3P_MM(F##a1[pa], F##b1[pb], acc[1][1]);
#define X3P_MFMAS(F) X3P_TERM(F, 2, 0) X3P_TERM(F, 0, 2) X3P_TERM(F, 1, 1) X3P_TERM(F, 1, 0) X3P_TERM(F, 0, 1) X3P_TERM(F, 0, 0)
    // (the s_waitcnt builtin, not inline asm: the compiler's own waitcnt pass must see that the fragment reads
    //  have completed, or it puts an lgkmcnt(0) in front of the next MFMAs -- behind the freshly issued reads)
#define X3P_SYNC()                                                                      \
    asm volatile("" ::: "memory");                                                      \
    __builtin_amdgcn_s_waitcnt(0x0076); /* vmcnt(6) expcnt(7 = none) lgkmcnt(0) */      \
    __builtin_amdgcn_s_barrier();                                                       \
    asm volatile("" ::: "memory");

    // issue order inside a k-step: one memory operation in the shadow of each MFMA (12 fragment reads, then the 6
    // DMA pieces), the last 6 MFMAs bare; without this hipcc sinks the reads next to their uses
#define X3P_INTERLEAVE()                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) {                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   /* 1 MFMA    */               \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   /* 1 DS read */               \
    }                                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {                                     \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                 \
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   /* 1 VMEM (LDS-DMA piece) */  \
    }                                                                                      \
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                                     \
    __builtin_amdgcn_sched_barrier(0);
    X3P_DMA(0, 0)
    X3P_DMA(1, min(1, nk - 1))
    X3P_DMA(2, min(2, nk - 1))
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");            // tile 0 landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    X3P_READ(f0, 0)
    X3P_SYNC()                                                   // tile 1 landed, stage 0 free
    int s0 = 0;                                                  // kt % 3
    for (int kt = 0; kt < nk; kt += 2) {
        const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s1 == 2 ? 0 : s1 + 1;
        X3P_READ(f1, s1)                                         // tile kt+1
        X3P_DMA(s0, min(kt + 3, nk - 1))                         // unconditional (clamped) so the vmcnt count is exact
        X3P_MFMAS(f0)                                            // tile kt
        X3P_INTERLEAVE()
        X3P_SYNC()
        X3P_READ(f0, s2)                                         // tile kt+2
        X3P_DMA(s1, min(kt + 4, nk - 1))
        X3P_MFMAS(f1)                                            // tile kt+1
        X3P_INTERLEAVE()
        X3P_SYNC()
        s0 = s2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // clamped tail pieces must land before LDS is reused
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#undef X3P_DMA
#undef X3P_READ
#undef X3P_TERM
#undef X3P_MM
#undef X3P_MFMAS
#undef X3P_SYNC
#undef X3P_INTERLEAVE
}

template <bool VEC4>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3p_kernel(const __bf16 *__restrict__ Apk,
                                                              const __bf16 *__restrict__ Bpk, float *C, int ldc, int M,
                                                              int N, int K, const float *__restrict__ bias,
                                                              const float *resid, int ldr, int act, int tiles_m,
                                                              int tiles_n, char *packed_out, int dbg) {
    __shared__ __attribute__((aligned(16))) char smem[X3_STAGES * 2 * X3_BLOCK_B];
    int tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn);
    f32x16 acc[2][2];
    x3p_mainloop<true>(Apk, Bpk, K, tm, tn, smem, acc, dbg);
    if (packed_out)
        epilogue_store_packed_t(acc, packed_out, N >> 4, M, N, tm * GEMM_BM, tn * GEMM_BN, bias, act);
    else
        epilogue_store_t<VEC4>(acc, C, ldc, M, N, tm * GEMM_BM, tn * GEMM_BN, bias, resid, ldr, act);
}

template <int KSEL>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3p_topk_kernel(const __bf16 *__restrict__ Apk,
                                                                   const __bf16 *__restrict__ Bpk, int M, int N, int K,
                                                                   float inv_temp, float *tile_max, float *tile_sum,
                                                                   float *cand_val, int *cand_idx, int tiles_m,
                                                                   int tiles_n) {
    __shared__ __attribute__((aligned(16))) char smem[X3_STAGES * 2 * X3_BLOCK_B];
    static_assert(128 * CT_LD * 4 <= X3_STAGES * 2 * X3_BLOCK_B, "epilogue tile must fit the staging ring");
    int tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn);
    f32x16 acc[2][2];
    x3p_mainloop<false>(Apk, Bpk, K, tm, tn, smem, acc);      // ends with a barrier
    epilogue_topk<KSEL, 2, true>(acc, reinterpret_cast<float *>(smem), M, N, tm * GEMM_BM, tn * GEMM_BN, tn, tiles_n, inv_temp,
                           tile_max, tile_sum, cand_val, cand_idx);
}

// ---- split-K for under-filled grids (small M: a handful of 128x128 tiles, each walking all of K serially, is
// latency-bound -- 59 us per GEMM at 40 rows).  The K range is cut into S slices (grid = tiles x S); every block
// stores its raw fp32 partial tile to a workspace [S][M][N], and a second kernel adds the slices IN A FIXED ORDER
// (deterministic, unlike atomics) and applies the epilogue: bias, activation, residual, fp32 or packed output.
__global__ __launch_bounds__(256, 2) void gemm_bf16x3p_splitk_kernel(const __bf16 *__restrict__ Apk,
                                                                     const __bf16 *__restrict__ Bpk, float *part, int M,
                                                                     int N, int K, int tiles_m, int tiles_n, int S) {
    __shared__ __attribute__((aligned(16))) char smem[X3_STAGES * 2 * X3_BLOCK_B];
    const int ntiles = tiles_m * tiles_n;
    const int slice = blockIdx.x / ntiles;
    int tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn, blockIdx.x - slice * ntiles);
    const int nks = K / X3_BK / S;
    f32x16 acc[2][2];
    x3p_mainloop<true>(Apk, Bpk, K, tm, tn, smem, acc, 0, slice * nks, nks);
    epilogue_store_t<true>(acc, part + (size_t)slice * M * N, N, M, N, tm * GEMM_BM, tn * GEMM_BN, nullptr, nullptr, 0,
                           CAPDEC_ACT_NONE);
}

__global__ void splitk_reduce_kernel(const float *__restrict__ part, int S, int M, int N, const float *__restrict__ bias,
                                     int act, const float *resid, int ldr, float *C, int ldc, char *packed_out,
                                     int fmt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // one float4 of the [M, N] result
    const int nq = N >> 2;
    if (i >= (size_t)M * nq) return;
    const int row = (int)(i / nq), col = (int)(i - (size_t)row * nq) * 4;
    const size_t sl = (size_t)M * N;
    float4 v = *reinterpret_cast<const float4 *>(part + (size_t)row * N + col);
    for (int s = 1; s < S; ++s) {
        const float4 p = *reinterpret_cast<const float4 *>(part + s * sl + (size_t)row * N + col);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    if (bias) {
        const float4 b = *reinterpret_cast<const float4 *>(bias + col);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    v.x = act_apply(v.x, act); v.y = act_apply(v.y, act); v.z = act_apply(v.z, act); v.w = act_apply(v.w, act);
    if (packed_out) {
        x3_store_quad(packed_out, N >> 4, row, col >> 4, (col >> 2) & 3, v, fmt);
        return;
    }
    if (resid) {
        const float4 r4 = *reinterpret_cast<const float4 *>(resid + (size_t)row * ldr + col);
        v.x = post_resid(v.x + r4.x, act); v.y = post_resid(v.y + r4.y, act);
        v.z = post_resid(v.z + r4.z, act); v.w = post_resid(v.w + r4.w, act);
    }
    *reinterpret_cast<float4 *>(C + (size_t)row * ldc + col) = v;
}

// split-K reduce + bias + residual (no activation) of a full-width result (N = row length <= 1024) with the LayerNorm
// that FOLLOWS it in the block stack fused in: one wavefront per row adds the S partial rows in slice order, writes the
// new residual-stream row C (fp32, may alias resid), then normalises it in registers and writes the packed operand of
// the next GEMM -- one launch instead of reduce + LayerNorm, and the row is read once instead of twice.
__global__ __launch_bounds__(256) void splitk_reduce_ln_kernel(const float *__restrict__ part, int S, int M, int N,
                                                               const float *__restrict__ bias, const float *resid, int ldr,
                                                               float *C, int ldc, const float *__restrict__ lnw,
                                                               const float *__restrict__ lnb, float eps,
                                                               char *__restrict__ ln_packed, int fmt) {
    constexpr int MAXV = 4;                                   // float4s per lane: N <= 1024
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = N >> 2, nk = N / X3_BK;
    const size_t sl = (size_t)M * N;
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + 64 * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < nv) {
            float4 a = reinterpret_cast<const float4 *>(part + (size_t)row * N)[idx];
            int k = 1;
            for (; k + 3 < S; k += 4) {          // four slices in flight, added in slice order
                const float4 p0 = reinterpret_cast<const float4 *>(part + (k + 0) * sl + (size_t)row * N)[idx];
                const float4 p1 = reinterpret_cast<const float4 *>(part + (k + 1) * sl + (size_t)row * N)[idx];
                const float4 p2 = reinterpret_cast<const float4 *>(part + (k + 2) * sl + (size_t)row * N)[idx];
                const float4 p3 = reinterpret_cast<const float4 *>(part + (k + 3) * sl + (size_t)row * N)[idx];
                a.x += p0.x; a.y += p0.y; a.z += p0.z; a.w += p0.w;
                a.x += p1.x; a.y += p1.y; a.z += p1.z; a.w += p1.w;
                a.x += p2.x; a.y += p2.y; a.z += p2.z; a.w += p2.w;
                a.x += p3.x; a.y += p3.y; a.z += p3.z; a.w += p3.w;
            }
            for (; k < S; ++k) {
                const float4 p = reinterpret_cast<const float4 *>(part + k * sl + (size_t)row * N)[idx];
                a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
            }
            if (bias) {
                const float4 b = reinterpret_cast<const float4 *>(bias)[idx];
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
            if (resid) {
                const float4 r4 = reinterpret_cast<const float4 *>(resid + (size_t)row * ldr)[idx];
                a.x += r4.x; a.y += r4.y; a.z += r4.z; a.w += r4.w;
            }
            reinterpret_cast<float4 *>(C + (size_t)row * ldc)[idx] = a;
            v[i] = a;
            s += (a.x + a.y) + (a.z + a.w);
        }
    }
    // LayerNorm exactly as layernorm_packed_kernel (elementwise.hip): two-pass in registers, biased variance
    const float mean = wave_sum(s) / (float)N;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float a = v[i].x - mean, b2 = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
            q += (a * a + b2 * b2) + (c * c + e * e);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)N + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float4 ww = reinterpret_cast<const float4 *>(lnw)[idx];
            const float4 bb = reinterpret_cast<const float4 *>(lnb)[idx];
            float4 o;
            o.x = (v[i].x - mean) * rstd * ww.x + bb.x;
            o.y = (v[i].y - mean) * rstd * ww.y + bb.y;
            o.z = (v[i].z - mean) * rstd * ww.z + bb.z;
            o.w = (v[i].w - mean) * rstd * ww.w + bb.w;
            x3_store_quad(ln_packed, nk, row, idx >> 2, idx & 3, o, fmt);
        }
    }
}

int launch_splitk_reduce(hipStream_t st, const float *part, int S, int M, int N, const GemmEpilogue &epi, float *C,
                         int ldc, int fmt) {
    const bool fuse_off = !tuning_of(epi).fuse_ln;
    // (below ~1000 rows the element-parallel reduce + a separate LayerNorm keep more loads in flight than one wavefront
    //  per row: 8 captions x beam 5 measured 74 vs 81 ms per pass)
    if (!fuse_off && M >= 1024 && epi.ln_out && epi.ln_w && epi.ln_b && epi.packed_out == nullptr && epi.act == CAPDEC_ACT_NONE &&
        N % 16 == 0 && N <= 1024 && ldc % 4 == 0 && ((uintptr_t)epi.ln_w & 15) == 0 && ((uintptr_t)epi.ln_b & 15) == 0) {
        hipLaunchKernelGGL(splitk_reduce_ln_kernel, dim3((M + 3) / 4), dim3(256), 0, st, part, S, M, N, epi.bias, epi.resid,
                           epi.ldr, C, ldc, epi.ln_w, epi.ln_b, epi.ln_eps, (char *)epi.ln_out, fmt);
        CAPDEC_HIP(hipGetLastError());
        if (epi.ln_done) *epi.ln_done = 1;
        return 0;
    }
    const size_t nq = (size_t)M * (N / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, part, S, M, N,
                       epi.bias, epi.act, epi.resid, epi.ldr, C, ldc, (char *)epi.packed_out, fmt);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// slices for a [M, N, K] problem (1 = no split).  Two under-filled regimes (the chip runs 512 blocks at a time):
//  (a) M <= 512 rows (at most four tile rows: the decode of small batches): S depends on K ONLY -- the largest divisor of
//      the k-step count that leaves >= 8 (an even number of) k-steps per slice -- so a row's result is bit-identical for
//      every batch size in this regime;
//  (b) larger M whose grid is still at most HALF of the chip (<= 256 tiles: the N = 768 projections at a few thousand
//      rows, e.g. 625 captions x beam 5 = 3125 rows -> 150 tiles): the largest S with tiles x S <= 512 and >= 16 (even)
//      k-steps per slice.  mlp.c_proj (K = 3072, 150 tiles walking 192 k-steps each): 58 -> ~35 us.
// Across a regime / S boundary the fp32 summation order of a row changes (round-off level; the unsplit regime above is
// again batch-size independent).  CAPDEC_SPLITK=0 disables both, CAPDEC_SPLITK_MID=0 only (b).
int gemm_splitk_slices(int M, int N, int K, const Tuning &t) {
    const bool off = !t.splitk, mid_off = !t.splitk_mid;
    const int nk = K / X3_BK;
    if (off || N % 4 != 0) return 1;
    int best = 1;
    if (M <= 4 * GEMM_BM) {
        for (int s = 2; s <= nk / 8; ++s)
            if (nk % s == 0 && (nk / s) % 2 == 0) best = s;
        return best;
    }
    if (mid_off) return 1;
    const int tiles = ((M + GEMM_BM - 1) / GEMM_BM) * ((N + GEMM_BN - 1) / GEMM_BN);
    if (tiles > 256) return 1;
    for (int s = 2; s <= nk / 16 && tiles * s <= 512; ++s)
        if (nk % s == 0 && (nk / s) % 2 == 0) best = s;
    return best;
}
// (Tried and removed: cutting only the tiles of the last, partly filled round of a > 512-tile grid into K-slices inside
//  the same launch, summed by the last piece to arrive at a per-tile counter.  Device-scope fences cost an L2 write-back +
//  invalidate per piece; with sc1 write-through dumps instead, the dump + arrival + re-read still cost more than the
//  round they save: 3125 x 3072 x 768: 55 -> 68 us, 25000 x 768 x 3072: 393 -> 440 us.  DESIGN.md section 5.)
size_t gemm_splitk_ws_bytes(int M, int N, int K, const Tuning &t) {
    const int s = gemm_splitk_slices(M, N, K, t);
    size_t b = s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
    for (int which : {10, 14}) b = std::max(b, pp_splitk_ws_bytes(which, M, N, K));         // (the ping-pong kernels' own split)
    return b;
}

int launch_gemm_bf16x3p(hipStream_t st, const void *Apacked, const void *Bpacked, float *C, int ldc, int M, int N, int K,
                        const GemmEpilogue &epi) {
    CAPDEC_CHECK(M > 0 && N > 0 && K > 0 && K % 64 == 0, "gemm_bf16x3p: K must be a multiple of 64");
    CAPDEC_CHECK(epi.packed_out == nullptr ||
                     (N % 64 == 0 && epi.resid == nullptr && ((uintptr_t)epi.bias & 15) == 0),
                 "gemm_bf16x3p: packed output needs N % 64 == 0, a 16-byte aligned bias and no residual");
    const int tiles_m = (M + GEMM_BM - 1) / GEMM_BM, tiles_n = (N + GEMM_BN - 1) / GEMM_BN;
    // float4 epilogue when every row segment is 16-byte aligned (always the case on the decode path)
    const bool vec4 = N % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)C & 15) == 0 &&
                      (epi.bias == nullptr || ((uintptr_t)epi.bias & 15) == 0) &&
                      (epi.resid == nullptr || (epi.ldr % 4 == 0 && ((uintptr_t)epi.resid & 15) == 0));
    const int S = (vec4 && epi.splitk_ws) ? gemm_splitk_slices(M, N, K, tuning_of(epi)) : 1;
    if (S > 1 && epi.splitk_ws_bytes >= (size_t)S * M * N * sizeof(float)) {
        float *part = (float *)epi.splitk_ws;
        hipLaunchKernelGGL(gemm_bf16x3p_splitk_kernel, dim3(tiles_m * tiles_n * S), dim3(256), 0, st,
                           (const __bf16 *)Apacked, (const __bf16 *)Bpacked, part, M, N, K, tiles_m, tiles_n, S);
        CAPDEC_HIP(hipGetLastError());
        return launch_splitk_reduce(st, part, S, M, N, epi, C, ldc, PK_BF16X3);
    }
    const int dbg = tuning_of(epi).x3_abl_dma;      // (measurement builds only: 0 otherwise)
    if (vec4)
        hipLaunchKernelGGL(gemm_bf16x3p_kernel<true>, dim3(tiles_m * tiles_n), dim3(256), 0, st, (const __bf16 *)Apacked,
                           (const __bf16 *)Bpacked, C, ldc, M, N, K, epi.bias, epi.resid, epi.ldr, epi.act, tiles_m,
                           tiles_n, (char *)epi.packed_out, dbg);
    else
        hipLaunchKernelGGL(gemm_bf16x3p_kernel<false>, dim3(tiles_m * tiles_n), dim3(256), 0, st, (const __bf16 *)Apacked,
                           (const __bf16 *)Bpacked, C, ldc, M, N, K, epi.bias, epi.resid, epi.ldr, epi.act, tiles_m,
                           tiles_n, (char *)epi.packed_out, dbg);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

int launch_gemm_bf16x3p_topk(hipStream_t st, const void *Apacked, const void *Bpacked, int M, int N, int K, int k,
                             float inv_temp, float *tile_max, float *tile_sum, float *cand_val, int *cand_idx) {
    CAPDEC_CHECK(M > 0 && N > 0 && K > 0 && K % 64 == 0, "gemm_bf16x3p_topk: K must be a multiple of 64");
    const int tiles_m = (M + GEMM_BM - 1) / GEMM_BM, tiles_n = (N + GEMM_BN - 1) / GEMM_BN;
    dim3 grid(tiles_m * tiles_n), block(256);
#define LAUNCH_TOPKP(KS)                                                                                          \
    hipLaunchKernelGGL(gemm_bf16x3p_topk_kernel<KS>, grid, block, 0, st, (const __bf16 *)Apacked,                  \
                       (const __bf16 *)Bpacked, M, N, K, inv_temp, tile_max, tile_sum, cand_val, cand_idx, tiles_m, tiles_n)
    switch (k) {
        case 1: LAUNCH_TOPKP(1); break;
        case 2: LAUNCH_TOPKP(2); break;
        case 3: LAUNCH_TOPKP(3); break;
        case 4: LAUNCH_TOPKP(4); break;
        case 5: LAUNCH_TOPKP(5); break;
        case 6: LAUNCH_TOPKP(6); break;
        case 7: LAUNCH_TOPKP(7); break;
        case 8: LAUNCH_TOPKP(8); break;
        default: CAPDEC_CHECK(false, "gemm_topk: k must be in 1..8");
    }
#undef LAUNCH_TOPKP
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

static bool x3_use_small_tile(int M, int tiles_n, const Tuning &t) {
    const int force = t.x3_tile_m;
    // Measured on MI355X (M = 3125 / 5000 decode rows): the 64-row tile doubles the blocks of an under-filled
    // grid but also doubles the B bytes per FLOP, and comes out 6-10 % SLOWER than 128-row tiles even at
    // 150 blocks on 256 CUs -- so it is opt-in only (CAPDEC_X3_TILE_M=64), e.g. for M <= 64.
    (void)M; (void)tiles_n;
    return force == 64;
}

int launch_gemm_bf16x3(hipStream_t st, const float *A, int lda, const void *Bpacked, float *C, int ldc, int M, int N,
                       int K, const GemmEpilogue &epi) {
    CAPDEC_CHECK(M > 0 && N > 0 && K > 0, "gemm: empty problem");
    CAPDEC_CHECK(K % 64 == 0 && lda % 4 == 0, "gemm_bf16x3: K must be a multiple of 64");
    CAPDEC_CHECK((((uintptr_t)A | (uintptr_t)Bpacked) & 15) == 0, "gemm_bf16x3: operands must be 16-byte aligned");
    // 64-row tiles when the 128-row grid would leave the chip (256 CUs x 2 blocks) under-filled
    const int tiles_n = (N + GEMM_BN - 1) / GEMM_BN;
    const bool small = x3_use_small_tile(M, tiles_n, tuning_of(epi));
    const int bm = small ? 64 : 128, tiles_m = (M + bm - 1) / bm;
    if (small)
        hipLaunchKernelGGL(gemm_bf16x3_kernel<1>, dim3(tiles_m * tiles_n), dim3(256), 0, st, A, lda, (const __bf16 *)Bpacked,
                           C, ldc, M, N, K, epi.bias, epi.resid, epi.ldr, epi.act, tiles_m, tiles_n);
    else
        hipLaunchKernelGGL(gemm_bf16x3_kernel<2>, dim3(tiles_m * tiles_n), dim3(256), 0, st, A, lda, (const __bf16 *)Bpacked,
                           C, ldc, M, N, K, epi.bias, epi.resid, epi.ldr, epi.act, tiles_m, tiles_n);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

int launch_gemm_bf16x3_topk(hipStream_t st, const float *A, int lda, const void *Bpacked, int M, int N, int K, int k,
                            float inv_temp, float *tile_max, float *tile_sum, float *cand_val, int *cand_idx) {
    CAPDEC_CHECK(M > 0 && N > 0 && K > 0, "gemm_topk: empty problem");
    CAPDEC_CHECK(K % 64 == 0 && lda % 4 == 0, "gemm_bf16x3_topk: K must be a multiple of 64");
    const int tiles_n = (N + GEMM_BN - 1) / GEMM_BN;
    const bool small = x3_use_small_tile(M, tiles_n, default_tuning());
    const int bm = small ? 64 : 128, tiles_m = (M + bm - 1) / bm;
    dim3 grid(tiles_m * tiles_n), block(256);
#define LAUNCH_TOPK(KS)                                                                                               \
    if (small)                                                                                                        \
        hipLaunchKernelGGL((gemm_bf16x3_topk_kernel<KS, 1>), grid, block, 0, st, A, lda, (const __bf16 *)Bpacked, M, N, K, \
                           inv_temp, tile_max, tile_sum, cand_val, cand_idx, tiles_m, tiles_n);                       \
    else                                                                                                              \
        hipLaunchKernelGGL((gemm_bf16x3_topk_kernel<KS, 2>), grid, block, 0, st, A, lda, (const __bf16 *)Bpacked, M, N, K, \
                           inv_temp, tile_max, tile_sum, cand_val, cand_idx, tiles_m, tiles_n)
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

CAPDEC_SAT_ACCESSOR(sat_count_gemm_bf16x3)

}  // namespace capdec
