// Round 4: the fp32-accurate three-product GEMM (format PK_F16X2, see gemm_f16x2.hip) as ONE 8-wavefront block per CU
// whose two wavefronts per SIMD alternate between a load phase and a matrix phase ("ping-pong"):
//     C[M,N] = epi( A[M,K] . Bt[N,K]^T )
//
// Why: the round-2 / round-3 kernels put two INDEPENDENT 4-wavefront blocks on a CU.  Each wavefront interleaves its own
// fragment reads and LDS-DMA pieces with its MFMAs and meets the other three wavefronts of its block at an s_barrier every
// k-step, while its SIMD partner -- a wavefront of the other block, at an arbitrary phase -- competes for the same matrix
// pipe: the SQ counters show that pipe busy 50-54 % of the cycles and the waves parked or issue-stalled the rest of the
// time, with random and with zero operands, capped and uncapped clocks alike (profiles/r3_pmc_sq_gemm.txt).  Here the two
// wavefronts of a SIMD belong to the SAME block and are kept half a k-step apart by the block's own barriers: while one
// group of four wavefronts issues nothing but MFMAs (at raised priority), the other group issues the ds_reads of its next
// fragments and its share of the LDS-DMA pieces; at the barrier they swap roles.  The matrix pipe of every SIMD is handed
// from one wavefront to the other without either of them having to hide its own memory operations under its own MFMAs.
// (The CDNA4 guide's plain-HIP bf16 GEMM template is built the same way; this is that structure for the two-plane
//  operands and the three products of the f16x2 scheme, with the packed tile-major operands of bf16x3.h.)
//
// Geometry (PGeo): WM x WN = 8 wavefronts, wave tile TI x TJ blocks of 32 x 32; wavefronts 0-3 form group 0, 4-7 group 1
// (the hardware places wavefront w and w + 4 of a block on the same SIMD).  A stage of the LDS ring holds one k-step
// (16 k, both planes) of the block tile's A and B rows in 1 KB pieces (32 rows x 32 B of one plane), chunk-major like
// gemm_h2w.hip; the ring has NS stages and the DMA runs D = NS - 2 k-steps ahead.
//
// Timeline (half-phases h, one s_barrier between consecutive ones; group 0 is at LOAD(kt) when h = 2 kt, group 1 when
// h = 2 kt + 1, each group's MMA(kt) is the half-phase after its LOAD(kt)):
//   LOAD(kt): ds_read the fragments of tile kt from stage kt % NS; send this wavefront's pieces of tile kt + D to stage
//             (kt + D) % NS; s_waitcnt vmcnt(PPW (D - 1))  -- this wavefront's pieces of tile kt + 1 have landed.
//   MMA(kt):  s_waitcnt lgkmcnt(0); s_setprio 1; the 3 TI TJ MFMAs of tile kt; s_setprio 0.
// Read-after-DMA: a fragment read of tile kt + 1 (group 0: h = 2 kt + 2, group 1: h = 2 kt + 3) comes after every
// wavefront's counted wait for that tile (group 0: end of h = 2 kt, group 1: end of h = 2 kt + 1) AND a barrier.
// DMA-after-read: stage (kt + D) % NS last held tile kt + D - NS, whose last reads (group 1, h = 2 (kt + D - NS) + 1) were
// retired by the lgkmcnt(0) at the start of h = 2 (kt + D - NS) + 2; the earliest overwrite is group 0's at h = 2 kt, and
// 2 (kt + D - NS) + 2 < 2 kt  <=>  D <= NS - 2.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "bf16x3.h"
#include "gemm_epilogue_lds.h"
#include "gemm_epilogue_w.h"

namespace capdec {

typedef __attribute__((address_space(3))) void lds_void_p;
typedef const __attribute__((address_space(1))) void glb_void_p;

__host__ __device__ constexpr int waitcnt_imm_p(int vm, int lgkm) {
    return (vm & 15) | (7 << 4) | ((lgkm & 15) << 8) | ((vm >> 4) << 14);
}

template <int WM_, int WN_, int TI_, int TJ_, int NS_, bool TWOACC_>
struct PGeo {
    static constexpr int WM = WM_, WN = WN_, TI = TI_, TJ = TJ_, NS = NS_;
    static constexpr bool TWOACC = TWOACC_;       // two accumulator sets (any weights) / one (weights with max |w| < 16)
    static constexpr int NW = WM * WN, THREADS = 64 * NW, MINW = 2;
    static constexpr int BM = WM * TI * 32, BN = WN * TJ * 32;
    static constexpr int CA = BM / 32, CB = BN / 32;
    static constexpr int PIECES = 2 * (CA + CB), PPW = (PIECES + NW - 1) / NW;   // (a remainder is padded with duplicates)
    static constexpr int STAGE_B = PIECES * 1024;
    static constexpr int SMEM_B = NS * STAGE_B;
    static constexpr int D = NS - 2;
    static_assert(NW == 8, "two groups of four wavefronts");
    static_assert(D >= 1 && PPW * D < 64, "ring depth / vmcnt range");
    static_assert(SMEM_B <= 160 * 1024, "LDS");
};

// acc (and ac for TWOACC) in the TR layout of gemm_epilogue.h when TR, the plain MFMA layout otherwise
// (ks0, nks: the k-steps [ks0, ks0 + nks) of the product -- split-K; nks < 0: all of them)
template <class G, bool TR, int ABL = 0>
__device__ __forceinline__ void pp_mainloop(const _Float16 *__restrict__ Apk, const _Float16 *__restrict__ Bpk, int K,
                                            int tm, int tn, int chunksA, int chunksB, char *smem,
                                            f32x16 (&acc)[G::TI][G::TJ], f32x16 (&ac)[G::TWOACC ? G::TI : 1][G::TWOACC ? G::TJ : 1],
                                            int ks0 = 0, int nks = -1, long long *stamp1 = nullptr) {
    constexpr int TI = G::TI, TJ = G::TJ, NS = G::NS, PPW = G::PPW, SB = G::STAGE_B, D = G::D;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / G::WN, wn = wave % G::WN, grp = wave >> 2;
    const int half = lane >> 5, l32 = lane & 31;
    const int nkf = K / X3_BK;                     // k-steps of the whole K (panel stride)
    const int nk = nks < 0 ? nkf : nks;            // k-steps of THIS block
    // ---- LDS-DMA: piece p of a stage = plane (p & 1) of chunk (p >> 1); chunks 0 .. CA-1 = the A rows of the block tile,
    // CA .. CA+CB-1 its B rows.  Global chunk g of a packed operand = piece (g & 3) of each plane of its 128-row tile g >> 2.
    const char *src[PPW];
    int dofs[PPW];
#pragma unroll
    for (int e = 0; e < PPW; ++e) {
        const int p = min(wave * PPW + e, G::PIECES - 1), c = p >> 1, pl = p & 1;
        const bool isA = c < G::CA;
        const int g = isA ? min(tm * G::CA + c, chunksA - 1) : min(tn * G::CB + (c - G::CA), chunksB - 1);
        const char *base = reinterpret_cast<const char *>(isA ? Apk : Bpk);
        src[e] = base + ((size_t)(g >> 2) * nkf + ks0) * H2_BLOCK_B + pl * X3_PLANE_B + (g & 3) * 1024 + lane * 16;
        dofs[e] = p * 1024;
    }
#define P_DMA(stage, ks_)                                                                                        \
    {                                                                                                            \
        _Pragma("unroll") for (int e = 0; e < PPW; ++e)                                                          \
            __builtin_amdgcn_global_load_lds((glb_void_p *)(src[e] + (size_t)(ks_) * H2_BLOCK_B),                 \
                                             (lds_void_p *)(smem + (stage) * SB + dofs[e]), 16, 0, 0);           \
    }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if constexpr (G::TWOACC) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) ac[i][j][r] = 0.f;
    }
    const int swz = ((half ^ ((l32 >> 3) & 1)) << 4);
    const int a_rd = (2 * wm * TI) * 1024 + l32 * X3_ROW_B + swz;                  // chunk wm TI + i, plane p: + (2 i + p) KB
    const int b_rd = (2 * (G::CA + wn * TJ)) * 1024 + l32 * X3_ROW_B + swz;
    f16x8 fa[TI][2], fb[TJ][2];
#define P_READ(stage)                                                                                            \
    {                                                                                                            \
        const char *rs = smem + (stage) * SB;                                                                    \
        _Pragma("unroll") for (int j = 0; j < TJ; ++j) {                                                         \
            fb[j][0] = *reinterpret_cast<const f16x8 *>(rs + b_rd + (2 * j) * 1024);                             \
            fb[j][1] = *reinterpret_cast<const f16x8 *>(rs + b_rd + (2 * j + 1) * 1024);                         \
        }                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < TI; ++i) {                                                         \
            fa[i][0] = *reinterpret_cast<const f16x8 *>(rs + a_rd + (2 * i) * 1024);                             \
            fa[i][1] = *reinterpret_cast<const f16x8 *>(rs + a_rd + (2 * i + 1) * 1024);                         \
        }                                                                                                        \
    }
#define P_MM1(x, y, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0)
#define P_MM(x, y, c) (TR ? P_MM1(y, x, c) : P_MM1(x, y, c))
#define P_BARRIER()                          \
    asm volatile("" ::: "memory");           \
    __builtin_amdgcn_s_barrier();            \
    asm volatile("" ::: "memory");           \
    __builtin_amdgcn_sched_barrier(0);

    // prologue: tiles 0 .. D-1 in flight, tile 0 landed everywhere; group 1 then waits one barrier more (half a k-step)
#pragma unroll
    for (int s = 0; s < D; ++s) P_DMA(s, min(s, nk - 1))
    __builtin_amdgcn_s_waitcnt(waitcnt_imm_p(PPW * (D - 1), 15));
    P_BARRIER()
#ifdef CAPDEC_MEASURE
    if (stamp1 && threadIdx.x == 0) *stamp1 = wall_clock64();          // first tile landed
#endif
    if constexpr (ABL == 2 || ABL == 5) { P_READ(0) }
    if (grp == 1) { P_BARRIER() }
    int s0 = 0, sd = D % NS;                     // kt % NS, (kt + D) % NS
    for (int kt = 0; kt < nk; ++kt) {
        // ---- LOAD(kt)
        // (ABL, measurement only -- wrong results: 1 = no LDS-DMA in the loop, 2 = no fragment reads, 3 = no MFMAs,
        //  4 = no barriers, 5 = neither DMA nor reads)
        if constexpr (ABL != 2 && ABL != 5) { P_READ(s0) }
        if constexpr (ABL != 1 && ABL != 5) {
            P_DMA(sd, min(kt + D, nk - 1))
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(waitcnt_imm_p(PPW * (D - 1), 15));
        }
        if constexpr (ABL != 4) { P_BARRIER() }
        // ---- MMA(kt)
        __builtin_amdgcn_s_waitcnt(waitcnt_imm_p(63, 0));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if constexpr (ABL == 3) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) asm volatile("" ::"v"(fa[i][0]), "v"(fa[i][1]), "v"(fb[j][0]), "v"(fb[j][1]));
        } else if constexpr (G::TWOACC) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) ac[i][j] = P_MM(fa[i][1], fb[j][0], ac[i][j]);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = P_MM(fa[i][0], fb[j][0], acc[i][j]);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) ac[i][j] = P_MM(fa[i][0], fb[j][1], ac[i][j]);
        } else {
            f16x8 bs[TJ];
#pragma unroll
            for (int j = 0; j < TJ; ++j) bs[j] = fb[j][0] * (_Float16)H2_LO_SCALE;
            // term-major over the whole wave tile: an accumulator comes up again only after TI TJ other MFMAs
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = P_MM(fa[i][1], fb[j][0], acc[i][j]);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = P_MM(fa[i][0], bs[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = P_MM(fa[i][0], fb[j][1], acc[i][j]);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ABL != 4) { P_BARRIER() }
        s0 = s0 + 1 == NS ? 0 : s0 + 1;
        sd = sd + 1 == NS ? 0 : sd + 1;
    }
    if (grp == 0) { P_BARRIER() }
    __builtin_amdgcn_s_waitcnt(waitcnt_imm_p(0, 15));            // clamped tail pieces must land before the ring is reused
    P_BARRIER()
#undef P_DMA
#undef P_READ
#undef P_MM1
#undef P_MM
#undef P_BARRIER
}

template <class G>
__device__ __forceinline__ void pp_join(f32x16 (&am)[G::TI][G::TJ], const f32x16 (&ac)[G::TWOACC ? G::TI : 1][G::TWOACC ? G::TJ : 1]) {
    if constexpr (G::TWOACC) {
#pragma unroll
        for (int i = 0; i < G::TI; ++i)
#pragma unroll
            for (int j = 0; j < G::TJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) am[i][j][r] = fmaf(ac[i][j][r], 1.0f / H2_LO_SCALE, am[i][j][r]);
    }
}

// persistent form: grid = min(tiles, CUs) blocks, block b walks tiles b, b + grid, ...
// (ABL / stamps: measurement builds only, -DCAPDEC_MEASURE; ABL 8 = the direct, uncoalesced epilogues of gemm_epilogue_w.h)
template <class G, int ABL = 0>
__global__ __launch_bounds__(G::THREADS, G::MINW) void gemm_pp_kernel(const _Float16 *__restrict__ Apk,
                                                                     const _Float16 *__restrict__ Bpk, float *C, int ldc,
                                                                     int M, int N, int K, const float *__restrict__ bias,
                                                                     const float *resid, int ldr, int act, int tiles_m,
                                                                     int tiles_n, char *packed_out, float scale, QkvScatter sc,
                                                                     long long *stamps) {
    __shared__ __attribute__((aligned(16))) char smem[G::SMEM_B];
    const int ntiles = tiles_m * tiles_n;
    const int chunksA = ((M + 127) >> 7) * 4, chunksB = ((N + 127) >> 7) * 4;
#ifdef CAPDEC_MEASURE
    if (stamps && threadIdx.x == 0) stamps[blockIdx.x * 4 + 0] = wall_clock64();
#endif
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int tm, tn;
        tile_coords(tiles_m, tiles_n, tm, tn, tile);
        f32x16 acc[G::TI][G::TJ], ac[G::TWOACC ? G::TI : 1][G::TWOACC ? G::TJ : 1];
#ifdef CAPDEC_MEASURE
        const bool st1 = stamps && tile == (int)blockIdx.x;
        pp_mainloop<G, true, (ABL >= 6 ? 0 : ABL)>(Apk, Bpk, K, tm, tn, chunksA, chunksB, smem, acc, ac, 0, -1,
                                                    st1 ? stamps + blockIdx.x * 4 + 1 : nullptr);
        if (st1 && threadIdx.x == 0) stamps[blockIdx.x * 4 + 2] = wall_clock64();
#else
        pp_mainloop<G, true>(Apk, Bpk, K, tm, tn, chunksA, chunksB, smem, acc, ac);
#endif
        pp_join<G>(acc, ac);
        EpiArgs ea;
        ea.C = C; ea.ldc = ldc; ea.M = M; ea.N = N; ea.m0 = tm * G::BM; ea.n0 = tn * G::BN;
        ea.bias = bias; ea.act = act; ea.scale = scale; ea.ldr = ldr;
        ea.packed = packed_out;
        if (packed_out) ea.resid_pk = reinterpret_cast<const char *>(resid);      // (with packed_out, `resid` is PACKED)
        else ea.resid = resid;
        ea.sc = &sc;
#ifdef CAPDEC_MEASURE
        if constexpr (ABL == 6) {          // no stores at all (a never-taken store keeps the accumulators alive)
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < G::TI; ++i)
#pragma unroll
                for (int j = 0; j < G::TJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
            if (sum == 123456.789f) C[threadIdx.x] = sum;
        } else if constexpr (ABL == 7) {   // fp32 C through the slabs with NON-TEMPORAL stores
            ea.nt = true;
            epilogue_lds<G>(acc, smem, ea);
        } else if constexpr (ABL == 8) {
            if (packed_out)
                epilogue_store_packed_tw<G>(acc, scale, packed_out, N >> 4, M, N, tm * G::BM, tn * G::BN, bias, act,
                                            reinterpret_cast<const char *>(resid));
            else
                epilogue_store_tw<G>(acc, scale, C, ldc, M, N, tm * G::BM, tn * G::BN, bias, resid, ldr, act);
        } else
#endif
        {
            epilogue_lds<G>(acc, smem, ea);
            if (tile + (int)gridDim.x < ntiles) {       // the next tile's DMA pieces land in the slabs: every wavefront must be done reading
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_waitcnt(waitcnt_imm_p(63, 0));
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
#ifdef CAPDEC_MEASURE
        if (st1) {
            __builtin_amdgcn_s_waitcnt(waitcnt_imm_p(0, 0));            // the tile's stores have left the CU
            if (threadIdx.x == 0) stamps[blockIdx.x * 4 + 3] = wall_clock64();
        }
#endif
    }
}

// split-K for grids that leave most CUs idle (the N = 768 projections of a few thousand rows): block b = (tile, slice)
// walks k-steps [slice nks, (slice + 1) nks) and writes its raw fp32 partial tile to part[slice][M][N]; the slices are
// summed in a fixed order by launch_splitk_reduce (gemm_bf16x3.hip), which applies the epilogue -- and the LayerNorm that
// follows, when there is one
template <class G>
__global__ __launch_bounds__(G::THREADS, G::MINW) void gemm_pp_splitk_kernel(const _Float16 *__restrict__ Apk,
                                                                            const _Float16 *__restrict__ Bpk, float *part,
                                                                            int M, int N, int K, int tiles_m, int tiles_n,
                                                                            int S, float scale) {
    __shared__ __attribute__((aligned(16))) char smem[G::SMEM_B];
    const int ntiles = tiles_m * tiles_n;
    const int slice = blockIdx.x / ntiles;
    int tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn, blockIdx.x - slice * ntiles);
    const int nks = K / X3_BK / S;
    f32x16 acc[G::TI][G::TJ], ac[G::TWOACC ? G::TI : 1][G::TWOACC ? G::TJ : 1];
    pp_mainloop<G, true>(Apk, Bpk, K, tm, tn, ((M + 127) >> 7) * 4, ((N + 127) >> 7) * 4, smem, acc, ac, slice * nks, nks);
    pp_join<G>(acc, ac);
    EpiArgs ea;
    ea.C = part + (size_t)slice * M * N; ea.ldc = N; ea.M = M; ea.N = N; ea.m0 = tm * G::BM; ea.n0 = tn * G::BN;
    ea.scale = scale;
    epilogue_lds<G>(acc, smem, ea);
}

using P256x128 = PGeo<4, 2, 2, 2, 5, true>;       // 8 waves x (64 x 64), two accumulator sets, 24 KB stages, 120 KB
using P256x192s = PGeo<4, 2, 2, 3, 4, false>;     // 8 waves x (64 x 96), ONE accumulator set (two would spill), 28 KB stages, 112 KB
#ifdef CAPDEC_MEASURE
// 256 x 256: wins the isolated micro-benchmark (+12 % on mlp.c_fc at 25 000 rows), loses 2 % inside the decode loop, and the
// remedy the stamps suggest -- storing tile i while tile i + 1's MFMAs run -- needs a second set of 128 accumulator registers
// (2 x 128 + fragments > the 256 a wavefront has at two per SIMD) or 256 KB of LDS to park them: out of the product build
// since round 5 (profiles/r4_gemm_pp.txt, docs/rounds.md)
using P256x256s = PGeo<2, 4, 4, 2, 4, false>;     // 8 waves x (128 x 64), ONE accumulator set (weights with max |w| < 16), 128 KB
// (round 6: the one-set 256 x 128 and the 128 x 256 forms -- measured no faster than the two-set 256 x 128 tile in round 4,
//  never planned -- are gone from the measurement build too)
#endif

int pp_splitk_slices(int which, int M, int N, int K);

// measurement builds: the main-loop ablations exist for geometry 10 (256 x 128), the direct epilogue (8) also for the
// 256 x 256 tile; anything else runs the product kernel
template <class G> constexpr int pp_abl_for(int a) {
    return (G::BM == 256 && G::BN == 128 && G::TWOACC) ? a : (a == 8 && G::BN == 256) ? 8 : 0;
}
template <class G>
static int launch_pp(hipStream_t st, const void *Apacked, const void *Bpacked, float *C, int ldc, int M, int N, int K,
                     const GemmEpilogue &epi, float scale, int S) {
    const int tiles_m = (M + G::BM - 1) / G::BM, tiles_n = (N + G::BN - 1) / G::BN;
    const int ntiles = tiles_m * tiles_n;
    const float kscale = G::TWOACC ? 1.0f : scale;
    if (S > 1) {
        float *part = (float *)epi.splitk_ws;
        hipLaunchKernelGGL((gemm_pp_splitk_kernel<G>), dim3(ntiles * S), dim3(G::THREADS), 0, st, (const _Float16 *)Apacked,
                           (const _Float16 *)Bpacked, part, M, N, K, tiles_m, tiles_n, S, kscale);
        CAPDEC_HIP(hipGetLastError());
        return launch_splitk_reduce(st, part, S, M, N, epi, C, ldc, PK_F16X2);
    }
    const int grid = ntiles <= 4 * 256 ? std::min(ntiles, 256) : ntiles;
    const float *resid_arg = epi.packed_out ? (const float *)epi.resid_packed : epi.resid;
    const QkvScatter sc = epi.qkv_scatter ? *epi.qkv_scatter : QkvScatter();
    long long *stamps = nullptr;
#define LAUNCH_PP(A)                                                                                                  \
    hipLaunchKernelGGL((gemm_pp_kernel<G, pp_abl_for<G>(A)>), dim3(grid), dim3(G::THREADS), 0, st, (const _Float16 *)Apacked, \
                       (const _Float16 *)Bpacked, C, ldc, M, N, K, epi.bias, resid_arg, epi.ldr, epi.act, tiles_m, tiles_n, \
                       (char *)epi.packed_out, kscale, sc, stamps)
#ifdef CAPDEC_MEASURE
    // CAPDEC_PP_ABL (WRONG results for 1..5): 1 = no LDS-DMA in the loop, 2 = no fragment reads, 3 = no MFMAs, 4 = no
    // barriers, 5 = neither DMA nor reads, 8 = direct epilogue; CAPDEC_PP_STAMPS=<file>: per-block phase stamps, appended
    const Tuning &tn = tuning_of(epi);
    const int abl = tn.pp_abl;
    const char *stamp_path = tn.pp_stamps.empty() ? nullptr : tn.pp_stamps.c_str();
    static long long *d_stamps = nullptr;
    if (stamp_path) {
        if (!d_stamps) CAPDEC_HIP(hipMalloc(&d_stamps, 4096 * 4 * sizeof(long long)));
        if (grid <= 4096) { stamps = d_stamps; CAPDEC_HIP(hipMemsetAsync(d_stamps, 0, 4096 * 4 * sizeof(long long), st)); }
    }
    switch (abl) {
        case 1: LAUNCH_PP(1); break;
        case 2: LAUNCH_PP(2); break;
        case 3: LAUNCH_PP(3); break;
        case 4: LAUNCH_PP(4); break;
        case 5: LAUNCH_PP(5); break;
        case 6: LAUNCH_PP(6); break;
        case 7: LAUNCH_PP(7); break;
        case 8: LAUNCH_PP(8); break;
        default: LAUNCH_PP(0);
    }
#else
    LAUNCH_PP(0);
#endif
#undef LAUNCH_PP
    CAPDEC_HIP(hipGetLastError());
#ifdef CAPDEC_MEASURE
    if (stamps) {
        std::vector<long long> h((size_t)grid * 4);
        CAPDEC_HIP(hipStreamSynchronize(st));
        CAPDEC_HIP(hipMemcpy(h.data(), d_stamps, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        if (FILE *f = fopen(stamp_path, "a")) {
            fprintf(f, "launch M %d N %d K %d grid %d tiles %d\n", M, N, K, grid, ntiles);
            for (int b = 0; b < grid; ++b) fprintf(f, "%lld %lld %lld %lld\n", h[b * 4], h[b * 4 + 1], h[b * 4 + 2], h[b * 4 + 3]);
            fclose(f);
        }
    }
#endif
    return 0;
}

static void pp_tile(int which, int &bm, int &bn) {
    bm = 256;
    bn = which == 12 ? 256 : which == 14 ? 192 : 128;
}

// K slices for a ping-pong launch (1 = none): grids of at most a third of the CUs are cut along K -- the largest S with
// tiles x S <= 256 that divides the k-steps and leaves >= 8 per slice.  (625 captions x beam 5 = 3125 rows: the N = 768
// projections are 78 tiles of 256 x 128 -> S = 3, 234 blocks.)  Needs the fp32 workspace and a float4-able result.
int pp_splitk_slices(int which, int M, int N, int K) {
    int bm, bn;
    pp_tile(which, bm, bn);
    const int tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn), nk = K / X3_BK;
    // (M <= 512 -- the small-batch regime whose results must not depend on the batch size -- is never cut here: the planner
    //  does not use these kernels there, and a FORCED geometry, CAPDEC_H2W >= 10, then runs unsplit whatever M is)
    if (N % 4 != 0 || tiles * 3 > 256 || M <= 512) return 1;
    int best = 1;
    for (int s = 2; tiles * s <= 256 && s <= nk / 8; ++s)
        if (nk % s == 0) best = s;
    return best;
}
size_t pp_splitk_ws_bytes(int which, int M, int N, int K) {
    const int s = pp_splitk_slices(which, M, N, K);
    return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}

// Which kernel for an f16x2 GEMM [M, N, K] of more than 512 rows: 0 = the kernels of rounds 2-3 (launch_gemm_f16x2p's own
// planner), else a ping-pong geometry.  The chip runs these GEMMs at its power limit (sustained launches of the old and
// the new structure reach the same k-steps per second per CU: tools/pp_probe.sh, profiles/r4_gemm_pp.txt), so the
// ping-pong kernels are used only where they were MEASURED faster inside the decode loop:
//  * large launches (>= 8192 rows, several rounds): the 256 x 256 tile for the wide projections (N >= 2048: qkv, mlp.c_fc)
//    -- half the operand bytes per MFMA of a 128 x 128 tile (+3 % / +12 % at 25 000 rows); wide_ok weights only;
//  * mid-size launches (513 .. 8191 rows, about one round): the tile whose grid fills the 256 CUs best -- 256 x 192 for
//    mlp.c_fc (3125 rows: 208 blocks instead of 400 half-speed ones), 256 x 128 cut along K for the N = 768 projections.
// mode (CAPDEC_PP): 2 = mid-size launches only (DEFAULT: inside the 5000-caption decode loop the 256 x 256 tile came out
// 2 % slower than the round-2 kernels although it wins the isolated micro-benchmark), 1 = both regimes, 3 = large only
int pp_plan(int M, int N, int K, bool wide_ok, bool can_split, int mode) {
    if (M >= 8192) {
#ifdef CAPDEC_MEASURE
        if (mode != 2) return (wide_ok && N >= 2048) ? 12 : 0;      // (the 256 x 256 tile: measurement builds only)
#endif
        return 0;
    }
    if (mode == 3) return 0;
    const int nk = K / X3_BK;
    auto blocks_pp = [&](int which) {
        int bm, bn;
        pp_tile(which, bm, bn);
        const long tiles = (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
        return tiles * (can_split ? pp_splitk_slices(which, M, N, K) : 1);
    };
    (void)nk;
    // one round of blocks that fills >= 75 % of the CUs: prefer the larger tile
    if (wide_ok && N >= 2048) {
        const long b14 = blocks_pp(14);
        if (b14 <= 256 && b14 >= 192 && N % 192 == 0) return 14;
    }
    const long b10 = blocks_pp(10);
    if (b10 <= 256 && b10 >= 192 && N <= 1024) return 10;
    return 0;
}

// `which`: 10 = 256x128 (two accumulator sets), 14 = 256x192 (one set: wide_ok weights); measurement builds: 12 (256x256).
// scale = 2^-11 (single-set geometries; ignored by the two-set ones).  Requires the float4 epilogue (caller checks).
int launch_gemm_pp(hipStream_t st, int which, const void *Apacked, const void *Bpacked, float *C, int ldc, int M, int N,
                   int K, const GemmEpilogue &epi, float scale) {
    int S = 1;
    if (epi.splitk_ws && !epi.resid_packed && !epi.packed_out && !epi.qkv_scatter && (which == 10 || which == 14 || which == 12)) {
        S = pp_splitk_slices(which, M, N, K);
        if (S > 1 && epi.splitk_ws_bytes < (size_t)S * M * N * sizeof(float)) S = 1;
    }
    switch (which) {
        case 10: return launch_pp<P256x128>(st, Apacked, Bpacked, C, ldc, M, N, K, epi, scale, S);
        case 14: return launch_pp<P256x192s>(st, Apacked, Bpacked, C, ldc, M, N, K, epi, scale, S);
#ifdef CAPDEC_MEASURE
        case 12: return launch_pp<P256x256s>(st, Apacked, Bpacked, C, ldc, M, N, K, epi, scale, S);
#endif
        default: CAPDEC_CHECK(false, "gemm_pp: unknown geometry");
    }
    return 0;
}

CAPDEC_SAT_ACCESSOR(sat_count_gemm_pp)

}  // namespace capdec
