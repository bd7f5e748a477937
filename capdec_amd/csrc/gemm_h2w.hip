// Round 3: the fp32-accurate three-product GEMM (format PK_F16X2, see gemm_f16x2.hip) with ONE accumulator set and wide
// wave tiles:  C[M,N] = epi( A[M,K] . Bt[N,K]^T ).
//
// Why: the 128x128 / 64x64-per-wave kernel of round 2 spends one s_barrier, 8 ds_read_b128 and 4 LDS-DMA pieces per
// 12 MFMAs, and its SQ counters (profiles/r2_pmc_sq_gemm_f16x2p.txt) show the matrix pipe busy only 58-64 % with the
// waves parked 27 % of their time: the loop is bound by what surrounds the MFMAs.  A wider wave tile amortises all of
// it, but the second accumulator set (the 2^-11-weighted cross terms) leaves no registers for one.
// (What it bought, profiles/r3_gemm_geometries.txt / r3_pmc_sq_gemm.txt: -4..8 % cycles, pipe busy 51 -> 54 %, +6 % at the
//  package power cap where the grid is whole rounds -- so a planner uses these kernels only where they remove a round.)
//
// How: a b = hi_a hi_b + 2^-11 (hi_a lo'_b + lo'_a hi_b) with lo' = 2^11 lo the stored low plane.  Multiply the hi_b
// FRAGMENT by 2^11 in registers after the ds_read (4 v_pk_mul_f16 per fragment: an exponent shift, exact while
// |b| < 32 -- B is always a weight matrix here; max |w| is measured once when its planes are packed and a matrix with
// max |w| >= 16 simply keeps the two-accumulator kernel, GemmEpilogue::wide_ok):
//     2^11 a b = hi_a (2^11 hi_b) + hi_a lo'_b + lo'_a hi_b
// -- three MFMAs into the SAME accumulator, result scaled by 2^-11 in the epilogue.  No change to the operand
// format: the activations' packers (LayerNorm, attention, fc epilogue) and the weight planes are those of round 2.
// The 64 freed registers go to a 128x64 wave tile (TI x TJ = 4 x 2 blocks of 32x32): 12 ds_read_b128, 8 v_pk_mul_f16
// and one barrier per 24 MFMAs, fragments of k-step kt+1 re-loaded into the registers of kt as soon as a row group's
// MFMAs have issued (one register set + the B fragments double-buffered).
//
// One template serves the geometries (waves WM x WN, wave tile TI x TJ, ring depth NS):
//   W256x128: 2x2 waves, 4x2 tiles, 256x128 block tile, 24 KB stages, NS = 3 (72 KB): two blocks per CU
//   W256x256: 2x4 waves (512 threads), 256x256 block tile, 32 KB stages, NS = 4 (128 KB): one block per CU
//   W128x192: 2x2 waves, 2x3 tiles, 20 KB stages, NS = 4 (80 KB): two blocks per CU -- column tiles of 192
//   W256x256q: 2x2 waves, 4x4 tiles (128x128 per wavefront, AGPR accumulators), one wavefront per SIMD (measurement)
#include <algorithm>
#include <cstdlib>

#include "bf16x3.h"
#include "gemm_epilogue_lds.h"
#include "gemm_epilogue_w.h"

namespace capdec {

typedef __attribute__((address_space(3))) void lds_void_w;
typedef const __attribute__((address_space(1))) void glb_void_w;

__host__ __device__ constexpr int waitcnt_imm_w(int vm, int lgkm) {
    return (vm & 15) | (7 << 4) | ((lgkm & 15) << 8) | ((vm >> 4) << 14);
}

template <int WM_, int WN_, int TI_, int TJ_, int NS_, int MINW_, bool ACCMAJOR_ = false>
struct WGeo {
    static constexpr bool ACCMAJOR = ACCMAJOR_;   // the three MFMAs of an accumulator back to back (measurement)
    static constexpr int WM = WM_, WN = WN_, TI = TI_, TJ = TJ_, NS = NS_, MINW = MINW_;
    static constexpr int NW = WM * WN, THREADS = 64 * NW;
    static constexpr int BM = WM * TI * 32, BN = WN * TJ * 32;
    // operands move in CHUNKS of 32 rows (one 1 KB LDS-DMA piece per plane): a block tile may start anywhere on a
    // 32-row boundary of the 128-row packed tiles, so BN = 192 is as good as 128 or 256
    static constexpr int CA = BM / 32, CB = BN / 32;
    static constexpr int PIECES = 2 * (CA + CB), PPW = PIECES / NW;      // pieces per stage / per wavefront
    static constexpr int STAGE_B = PIECES * 1024;                        // one k-step of both operands
    static constexpr int SMEM_B = NS * STAGE_B;
    static_assert(PIECES % NW == 0, "pieces divide evenly over the wavefronts");
    static_assert(PPW * (NS - 2) < 64, "vmcnt range");
};

// Issue schedule of one k-step.  The k-step is cut into TI REGIONS (one per row group of the wave tile, closed by a
// sched_barrier so the compiler cannot merge the groups: left alone it re-sorts the MFMAs column-major and the fragment
// re-loads land behind the last MFMA).  Slot s (1-based) of a k-step = its s-th MFMA, optionally followed by one memory
// operation:
//   region 0 (slots 1 .. 3 TJ): the B reads of the next tile and the A reads of its last row group (2 TJ + 2 reads)
//   region g > 0: first the A reads of group g-1 of the next tile (their registers were freed by region g-1)
//   the LDS-DMA pieces take the first PPW slots still free after region 0
// kind(s): 0 = MFMA alone, 1 = + a DS read, 2 = + an LDS-DMA piece
template <class G> struct WSched {
    static constexpr int GM = 3 * G::TJ, NSLOT = G::TI * GM, HEAD = 2 * G::TJ + 2;
    static_assert(HEAD <= GM, "region 0 holds the B reads");
    static constexpr int kind(int s) {
        int k[NSLOT + 2] = {};
        for (int i = 1; i <= HEAD; ++i) k[i] = 1;
        for (int g = 1; g < G::TI; ++g) k[g * GM + 1] = k[g * GM + 2] = 1;
        int dma = G::PPW;
        for (int i = GM + 1; i <= NSLOT && dma; ++i)
            if (!k[i]) { k[i] = 2; --dma; }
        return k[s];
    }
    static constexpr int dma_before(int g) { int n = 0; for (int i = 1; i <= g * GM; ++i) n += kind(i) == 2; return n; }
    static_assert(dma_before(G::TI) == G::PPW, "every LDS-DMA piece of a k-step has a slot");
};
template <class G, int S, int END>
__device__ __forceinline__ void w_sched_emit() {
    if constexpr (S <= END) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (WSched<G>::kind(S) == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        else if constexpr (WSched<G>::kind(S) == 2) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        w_sched_emit<G, S + 1, END>();
    }
}

// Main loop.  acc[i][j] (TR layout, see gemm_epilogue.h): C[m0 + wm TI 32 + i 32 + (lane & 31)]
//                                                          [n0 + wn TJ 32 + j 32 + 8 (r >> 2) + 4 (lane >> 5) + (r & 3)]
// scaled by 2^11.  chunksA / chunksB = number of 32-row chunks the packed operands hold (rows padded to 128: block tiles
// past the end re-read the last chunk; the epilogue drops those rows / columns).
template <class G, bool TR>
__device__ __forceinline__ void h2w_mainloop(const _Float16 *__restrict__ Apk, const _Float16 *__restrict__ Bpk, int K,
                                             int tm, int tn, int chunksA, int chunksB, char *smem,
                                             f32x16 (&acc)[G::TI][G::TJ]) {
    constexpr int TI = G::TI, TJ = G::TJ, NS = G::NS, PPW = G::PPW, SB = G::STAGE_B;
    using SC = WSched<G>;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / G::WN, wn = wave % G::WN;
    const int half = lane >> 5, l32 = lane & 31;
    const int nk = K / X3_BK;
    // ---- LDS-DMA: piece p = wave PPW + e of a stage = plane (p & 1) of chunk (p >> 1); chunks 0 .. CA-1 are the A rows
    // of the block tile, CA .. CA+CB-1 its B rows.  Global chunk g of a packed operand = piece (g & 3) of each plane of
    // its 128-row tile g >> 2.  The LDS image is chunk-major: [chunk][plane][32 rows x 32 B].
    const char *src[PPW];
#pragma unroll
    for (int e = 0; e < PPW; ++e) {
        const int p = wave * PPW + e, c = p >> 1, pl = p & 1;
        const bool isA = c < G::CA;
        const int g = isA ? min(tm * G::CA + c, chunksA - 1) : min(tn * G::CB + (c - G::CA), chunksB - 1);
        const char *base = reinterpret_cast<const char *>(isA ? Apk : Bpk);
        src[e] = base + (size_t)(g >> 2) * nk * H2_BLOCK_B + pl * X3_PLANE_B + (g & 3) * 1024 + lane * 16;
    }
    char *dst0 = smem + wave * (PPW * 1024);
#define W_DMA(stage, ks_, e0, e1)                                                                               \
    {                                                                                                           \
        _Pragma("unroll") for (int e = (e0); e < (e1); ++e)                                                     \
            __builtin_amdgcn_global_load_lds((glb_void_w *)(src[e] + (size_t)(ks_) * H2_BLOCK_B),                \
                                             (lds_void_w *)(dst0 + (stage) * SB + e * 1024), 16, 0, 0);         \
    }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int swz = ((half ^ ((l32 >> 3) & 1)) << 4);
    const int a_rd = (2 * wm * TI) * 1024 + l32 * X3_ROW_B + swz;                  // chunk wm TI + i, plane p: + (2 i + p) KB
    const int b_rd = (2 * (G::CA + wn * TJ)) * 1024 + l32 * X3_ROW_B + swz;

    // fragment sets: a[i][plane], b[j][plane]
    f16x8 f0a[TI][2], f0b[TJ][2], f1a[TI][2], f1b[TJ][2];
#define W_READ_A(F, stage, i)                                                                                    \
    {                                                                                                            \
        F##a[i][0] = *reinterpret_cast<const f16x8 *>(smem + (stage) * SB + a_rd + (2 * (i)) * 1024);       \
        F##a[i][1] = *reinterpret_cast<const f16x8 *>(smem + (stage) * SB + a_rd + (2 * (i) + 1) * 1024);   \
    }
#define W_READ_B(F, stage)                                                                                         \
    {                                                                                                              \
        _Pragma("unroll") for (int j = 0; j < TJ; ++j) {                                                           \
            F##b[j][0] = *reinterpret_cast<const f16x8 *>(smem + (stage) * SB + b_rd + (2 * j) * 1024);       \
            F##b[j][1] = *reinterpret_cast<const f16x8 *>(smem + (stage) * SB + b_rd + (2 * j + 1) * 1024);   \
        }                                                                                                          \
    }
#define W_MM1(x, y, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0)
#define W_MM(x, y, c) (TR ? W_MM1(y, x, c) : W_MM1(x, y, c))
    // the MFMAs of row group i of the tile held in fragment set F (bs[j] = 2^11 hi_b[j])
#define W_GROUP(F, i)                                                                                      \
    if constexpr (G::ACCMAJOR) {                                                                           \
        _Pragma("unroll") for (int j = 0; j < TJ; ++j) {                                                   \
            acc[i][j] = W_MM(F##a[i][1], F##b[j][0], acc[i][j]);                                           \
            acc[i][j] = W_MM(F##a[i][0], bs[j], acc[i][j]);                                                \
            acc[i][j] = W_MM(F##a[i][0], F##b[j][1], acc[i][j]);                                           \
        }                                                                                                  \
    } else {   /* term-major: consecutive MFMAs never share an accumulator */                              \
        _Pragma("unroll") for (int j = 0; j < TJ; ++j) acc[i][j] = W_MM(F##a[i][1], F##b[j][0], acc[i][j]); \
        _Pragma("unroll") for (int j = 0; j < TJ; ++j) acc[i][j] = W_MM(F##a[i][0], bs[j], acc[i][j]);      \
        _Pragma("unroll") for (int j = 0; j < TJ; ++j) acc[i][j] = W_MM(F##a[i][0], F##b[j][1], acc[i][j]); \
    }
#define W_SYNC()                                                                       \
    asm volatile("" ::: "memory");                                                     \
    __builtin_amdgcn_s_waitcnt(waitcnt_imm_w(PPW * (NS - 2), 0));                      \
    __builtin_amdgcn_s_barrier();                                                      \
    asm volatile("" ::: "memory");                                                     \
    __builtin_amdgcn_sched_barrier(0);     /* (the next k-step's MFMAs only read registers: keep them behind the barrier) */
#define W_REGION(CUR, NXT, s_next, s_dma, tile_dma, g)                                                      \
    if constexpr ((g) < TI) {                                                                               \
        if constexpr ((g) == 0) { W_READ_B(NXT, s_next) W_READ_A(NXT, s_next, TI - 1) }                     \
        else { W_READ_A(NXT, s_next, (g) - 1) }                                                             \
        constexpr int e0_ = SC::dma_before(g), e1_ = SC::dma_before((g) + 1);   /* (forces compile-time evaluation) */ \
        W_DMA(s_dma, tile_dma, e0_, e1_)                                                                    \
        W_GROUP(CUR, g)                                                                                     \
        w_sched_emit<G, (g) * SC::GM + 1, ((g) + 1) * SC::GM>();                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    }
#define W_STEP(CUR, NXT, s_next, s_dma, tile_dma)                                      \
    {                                                                                  \
        f16x8 bs[TJ];                                                                  \
        _Pragma("unroll") for (int j = 0; j < TJ; ++j) bs[j] = CUR##b[j][0] * (_Float16)H2_LO_SCALE; \
        W_REGION(CUR, NXT, s_next, s_dma, tile_dma, 0)                                 \
        W_REGION(CUR, NXT, s_next, s_dma, tile_dma, 1)                                 \
        W_REGION(CUR, NXT, s_next, s_dma, tile_dma, 2)                                 \
        W_REGION(CUR, NXT, s_next, s_dma, tile_dma, 3)                                 \
        W_SYNC()                                                                       \
    }

    // prologue: tiles 0 .. NS-1 in flight (stage s <- tile s); tile 0 landed -> fragment set f0; tile 1 landed
#pragma unroll
    for (int s = 0; s < NS; ++s) W_DMA(s, min(s, nk - 1), 0, PPW)
    __builtin_amdgcn_s_waitcnt(waitcnt_imm_w(PPW * (NS - 1), 15));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    W_READ_B(f0, 0)
#pragma unroll
    for (int i = 0; i < TI; ++i) W_READ_A(f0, 0, i)
    W_SYNC()
    // k-step kt: MFMAs of tile kt from registers, fragments of tile kt+1 read from stage (kt+1) % NS, tile kt+NS sent
    // to stage kt % NS (its fragments were all read during k-step kt-1, before the barrier that ended it).  End of the
    // k-step: all but the NS-2 newest tiles have landed => tile kt+2 is in LDS.
    int s0 = 0;
    for (int kt = 0; kt < nk; kt += 2) {
        const int s1 = s0 + 1 == NS ? 0 : s0 + 1, s2 = s1 + 1 == NS ? 0 : s1 + 1;
        W_STEP(f0, f1, s1, s0, min(kt + NS, nk - 1))
        W_STEP(f1, f0, s2, s1, min(kt + 1 + NS, nk - 1))
        s0 = s2;
    }
    __builtin_amdgcn_s_waitcnt(waitcnt_imm_w(0, 15));            // clamped tail pieces must land before LDS is reused
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#undef W_DMA
#undef W_READ_A
#undef W_READ_B
#undef W_MM1
#undef W_MM
#undef W_GROUP
#undef W_SYNC
#undef W_REGION
#undef W_STEP
}

// persistent form: grid = min(tiles, slots) blocks, block b walks tiles b, b + grid, ... (see gemm_f16x2p_kernel)
template <class G>
__global__ __launch_bounds__(G::THREADS, G::MINW) void gemm_h2w_kernel(const _Float16 *__restrict__ Apk,
                                                                      const _Float16 *__restrict__ Bpk, float *C, int ldc,
                                                                      int M, int N, int K, const float *__restrict__ bias,
                                                                      const float *resid, int ldr, int act, int tiles_m,
                                                                      int tiles_n, char *packed_out, float scale) {
    __shared__ __attribute__((aligned(16))) char smem[G::SMEM_B];
    const int ntiles = tiles_m * tiles_n;
    const int chunksA = ((M + 127) >> 7) * 4, chunksB = ((N + 127) >> 7) * 4;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int tm, tn;
        tile_coords(tiles_m, tiles_n, tm, tn, tile);
        f32x16 acc[G::TI][G::TJ];
        h2w_mainloop<G, true>(Apk, Bpk, K, tm, tn, chunksA, chunksB, smem, acc);     // ends with a barrier: the ring is free
        EpiArgs ea;       // coalesced epilogue through the (idle) ring: gemm_epilogue_lds.h
        ea.C = C; ea.ldc = ldc; ea.M = M; ea.N = N; ea.m0 = tm * G::BM; ea.n0 = tn * G::BN;
        ea.bias = bias; ea.act = act; ea.scale = scale; ea.ldr = ldr;
        ea.packed = packed_out;
        if (packed_out) ea.resid_pk = reinterpret_cast<const char *>(resid);      // (with packed_out, `resid` is PACKED)
        else ea.resid = resid;
        epilogue_lds<G>(acc, smem, ea);
        if (tile + (int)gridDim.x < ntiles) {       // the next tile's DMA pieces land in the slabs
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(waitcnt_imm_w(63, 0));
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
}

template <class G, int KSEL>
__global__ __launch_bounds__(G::THREADS, G::MINW) void gemm_h2w_topk_kernel(const _Float16 *__restrict__ Apk,
                                                                           const _Float16 *__restrict__ Bpk, int M, int N,
                                                                           int K, float scale, float *tile_max,
                                                                           float *tile_sum, float *cand_val, int *cand_idx,
                                                                           int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) char smem[G::SMEM_B];
    int tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn);
    f32x16 acc[G::TI][G::TJ];
    h2w_mainloop<G, false>(Apk, Bpk, K, tm, tn, ((M + 127) >> 7) * 4, ((N + 127) >> 7) * 4, smem, acc);   // ends with a barrier
    epilogue_topk_w<G, KSEL>(acc, scale, reinterpret_cast<float *>(smem), M, N, tm * G::BM, tn * G::BN, tn, tiles_n, tile_max,
                             tile_sum, cand_val, cand_idx);
}

// The same kernel over a row count only the DEVICE knows (*m_dev rows of a compacted A operand: the exact second pass of
// the fused lm_head, decode.hip): a fixed grid of persistent blocks walks the tiles there are; with *m_dev == 0 every block
// leaves at once.
template <class G, int KSEL>
__global__ __launch_bounds__(G::THREADS, G::MINW) void gemm_h2w_topk_dev_kernel(const _Float16 *__restrict__ Apk,
                                                                               const _Float16 *__restrict__ Bpk,
                                                                               const int *__restrict__ m_dev, int N, int K,
                                                                               float scale, float *tile_max, float *tile_sum,
                                                                               float *cand_val, int *cand_idx, int tiles_n) {
    __shared__ __attribute__((aligned(16))) char smem[G::SMEM_B];
    const int M = *m_dev;
    const int tiles_m = (M + G::BM - 1) / G::BM, ntiles = tiles_m * tiles_n;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int tm, tn;
        tile_coords(tiles_m, tiles_n, tm, tn, tile);
        f32x16 acc[G::TI][G::TJ];
        h2w_mainloop<G, false>(Apk, Bpk, K, tm, tn, ((M + 127) >> 7) * 4, ((N + 127) >> 7) * 4, smem, acc);   // ends with a barrier
        epilogue_topk_w<G, KSEL>(acc, scale, reinterpret_cast<float *>(smem), M, N, tm * G::BM, tn * G::BN, tn, tiles_n,
                                 tile_max, tile_sum, cand_val, cand_idx);      // (ends with a barrier: the ring is free again)
    }
}

using W256x128 = WGeo<2, 2, 4, 2, 3, 2>;      // 4 waves, 72 KB, two blocks per CU
#ifdef CAPDEC_MEASURE
using W256x256 = WGeo<2, 4, 4, 2, 4, 2>;      // 8 waves, 128 KB, one block per CU (measured: -4 .. -27 %)
// (measured and removed from the build, profiles/r3_gemm_geometries.txt: WGeo<2,2,2,2,3,3> = 128x128 at three blocks per CU
//  and WGeo<2,2,2,2,4,2> at two: +-0 against the round-2 kernel; WGeo<2,2,4,2,3,2,true>, accumulator-major MFMA order: -4 %)
using W256x256q = WGeo<2, 2, 4, 4, 4, 1>;     // 4 waves x (128 x 128), 128 KB, ONE wavefront per SIMD (accumulators in AGPRs: -40 %)
#endif
using W128x192 = WGeo<2, 2, 2, 3, 4, 2>;      // 4 waves x (64 x 96), 80 KB, two blocks per CU: column tiles of 192

// max |w| of a device matrix as the bit pattern of a non-negative float (atomicMax on the bits is order preserving)
__global__ void absmax_bits_kernel(const float *__restrict__ w, size_t n, unsigned *__restrict__ out) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float a = __builtin_fabsf(w[i]);
        m = (a > m || a != a) ? a : m;                          // a NaN weight poisons the maximum: not "wide ok"
    }
    m = wave_max(m == m ? m : __builtin_inff());
    if ((threadIdx.x & 63) == 0) atomicMax(out, __builtin_bit_cast(unsigned, m));
}
int launch_absmax_bits(hipStream_t st, const float *w, size_t n, unsigned *d_out) {
    CAPDEC_HIP(hipMemsetAsync(d_out, 0, sizeof(unsigned), st));
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(absmax_bits_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, w, n, d_out);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// Which kernel for an f16x2 GEMM [M, N, K] whose grid is not split along K: 0 = the round-2 128x128 two-accumulator
// kernel, else a WGeo id of launch_gemm_h2w.  Estimated time = (full rounds of 512 blocks + what the partly filled last
// round costs) x tile area / relative loop efficiency; a block alone on its CU runs ~1.64x faster than one of a pair
// (measured: one block per CU = 0.82 of two), so a last round of <= 256 tiles costs 0.61.  The wide tiles only win
// where they remove a round: at 25 000 rows the estimates tie (measured: within +-3 %, profiles/r3_gemm_geometries.txt)
// and the round-2 kernel stays; at a few thousand rows mlp.c_fc (600 tiles of 128x128 = 1.17 rounds) takes the
// 128x192 tile (400 tiles, one round).
int h2w_plan(int M, int N, int K) {
    (void)K;
    auto cost = [&](int bm, int bn, double eff) {
        const long tiles = (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
        const long full = tiles / 512, rem = tiles % 512;
        const double rounds = (double)full + (rem == 0 ? 0.0 : rem <= 256 ? 0.61 : 1.0);
        return rounds * bm * bn / eff;
    };
    const double c0 = cost(128, 128, 1.0), c8 = cost(128, 192, 1.04), c2 = cost(256, 128, 1.05);
    int best = 0;
    double cb = c0 * 0.97;                       // the wide kernel must be worth >= 3 %
    if (c8 < cb) { best = 8; cb = c8; }
    if (c2 < cb) { best = 2; cb = c2; }
    return best;
}

template <class G>
static int launch_h2w(hipStream_t st, const void *Apacked, const void *Bpacked, float *C, int ldc, int M, int N, int K,
                      const GemmEpilogue &epi, float scale, int slots) {
    const int tiles_m = (M + G::BM - 1) / G::BM, tiles_n = (N + G::BN - 1) / G::BN;
    const int ntiles = tiles_m * tiles_n;
    const int grid = ntiles <= 4 * slots ? std::min(ntiles, slots) : ntiles;
    const float *resid_arg = epi.packed_out ? (const float *)epi.resid_packed : epi.resid;
    hipLaunchKernelGGL((gemm_h2w_kernel<G>), dim3(grid), dim3(G::THREADS), 0, st, (const _Float16 *)Apacked,
                       (const _Float16 *)Bpacked, C, ldc, M, N, K, epi.bias, resid_arg, epi.ldr, epi.act, tiles_m, tiles_n,
                       (char *)epi.packed_out, scale);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// scale = 2^-11 (x 2^t for weights packed with a pre-scale 2^-t; t = 0 today: weights with |w| >= 16 keep the
// two-accumulator kernel, see planes_of in capi.hip).  Requires the float4 epilogue (caller checks).
int launch_gemm_h2w(hipStream_t st, int which, const void *Apacked, const void *Bpacked, float *C, int ldc, int M, int N,
                    int K, const GemmEpilogue &epi, float scale) {
    switch (which) {
        case 2: return launch_h2w<W256x128>(st, Apacked, Bpacked, C, ldc, M, N, K, epi, scale, 512);
#ifdef CAPDEC_MEASURE
        case 3: return launch_h2w<W256x256>(st, Apacked, Bpacked, C, ldc, M, N, K, epi, scale, 256);
        case 6: return launch_h2w<W256x256q>(st, Apacked, Bpacked, C, ldc, M, N, K, epi, scale, 256);
#endif
        case 8: return launch_h2w<W128x192>(st, Apacked, Bpacked, C, ldc, M, N, K, epi, scale, 512);
        default: CAPDEC_CHECK(false, "gemm_h2w: unknown geometry");
    }
    return 0;
}

// fused lm_head on the 256 x 128 tile: same partial lists per (row, 128-column tile) as launch_gemm_f16x2p_topk
int launch_gemm_h2w_topk(hipStream_t st, const void *Apacked, const void *Bpacked, int M, int N, int K, int k,
                         float inv_temp, float *tile_max, float *tile_sum, float *cand_val, int *cand_idx,
                         const Tuning *tune) {
    CAPDEC_CHECK(M > 0 && N > 0 && K > 0 && K % 64 == 0, "gemm_h2w_topk: K must be a multiple of 64");
    using G = W256x128;
    const int tiles_m = (M + G::BM - 1) / G::BM, tiles_n = (N + G::BN - 1) / G::BN;
    dim3 grid(tiles_m * tiles_n), block(G::THREADS);
    const float scale = inv_temp / H2_LO_SCALE;
#ifdef CAPDEC_MEASURE
    // CAPDEC_LMHEAD_K1=1 (WRONG results): run the k = 1 epilogue whatever k is -- what the top-k selection rounds cost
    if (tune && tune->lmhead_k1) k = 1;
#else
    (void)tune;
#endif
#define LAUNCH_TOPKW(KS)                                                                                           \
    hipLaunchKernelGGL((gemm_h2w_topk_kernel<G, KS>), grid, block, 0, st, (const _Float16 *)Apacked,                \
                       (const _Float16 *)Bpacked, M, N, K, scale, tile_max, tile_sum, cand_val, cand_idx, tiles_m, tiles_n)
    switch (k) {
        case 1: LAUNCH_TOPKW(1); break;
        case 2: LAUNCH_TOPKW(2); break;
        case 3: LAUNCH_TOPKW(3); break;
        case 4: LAUNCH_TOPKW(4); break;
        case 5: LAUNCH_TOPKW(5); break;
        case 6: LAUNCH_TOPKW(6); break;
        case 7: LAUNCH_TOPKW(7); break;
        case 8: LAUNCH_TOPKW(8); break;
        default: CAPDEC_CHECK(false, "gemm_topk: k must be in 1..8");
    }
#undef LAUNCH_TOPKW
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// k = 5 on *m_dev rows (m_cap = the capacity the output lists were sized for)
int launch_gemm_h2w_topk_dev(hipStream_t st, const void *Apacked, const void *Bpacked, const int *m_dev, int N, int K,
                             float inv_temp, float *tile_max, float *tile_sum, float *cand_val, int *cand_idx) {
    CAPDEC_CHECK(m_dev && N > 0 && K > 0 && K % 64 == 0, "gemm_h2w_topk_dev: bad argument");
    using G = W256x128;
    const int tiles_n = (N + G::BN - 1) / G::BN;
    hipLaunchKernelGGL((gemm_h2w_topk_dev_kernel<G, 5>), dim3(512), dim3(G::THREADS), 0, st, (const _Float16 *)Apacked,
                       (const _Float16 *)Bpacked, m_dev, N, K, inv_temp / H2_LO_SCALE, tile_max, tile_sum, cand_val, cand_idx,
                       tiles_n);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

CAPDEC_SAT_ACCESSOR(sat_count_gemm_h2w)

}  // namespace capdec
