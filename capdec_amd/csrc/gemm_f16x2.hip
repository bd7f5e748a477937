// fp32-accurate GEMM on the fp16 matrix cores with THREE MFMAs per product:  C[M,N] = epi( A[M,K] . Bt[N,K]^T )
//
// Why: rocprofv3 SQ counters of the six-product split-bf16 kernel (profiles/r2_pmc_sq_gemm_bf16x3p.txt) show the
// matrix pipe busy 71-78 % of the kernel's cycles and the waves issue-stalled on it 61 % of their time: that kernel
// is matrix-pipe-bound in cycles (the rest of the distance to the datasheet peak is clock), so the lever is FEWER
// MFMAs per fp32 product, not a denser schedule of the same six.
//
// How (format PK_F16X2, bf16x3.h): every fp32 value is  a = hi + 2^-11 lo  with two fp16 planes (11 + 11 significand
// bits, the low plane stored scaled by 2^11 so it never leaves fp16's normal range), hence
//     a b = hi_a hi_b + 2^-11 (hi_a lo_b + lo_a hi_b) + [2^-22 lo_a lo_b, dropped]
// The dropped term is <= 2^-22 |a b| (rms 2^-24.6 |a b|): measured against fp64 the result sits BELOW the rounding
// noise of an ordinary fp32 GEMM (max error / sum |a||b|: 2e-8 on unit-scale data, 1.7e-7 with a 10^6 dynamic range
// across k; torch's own fp32 matmul: 1.5e-7 / 2.4e-7).  v_mfma_f32_32x32x16_f16 products are exact in fp32 and
// accumulate in fp32; the 2^-11-weighted cross terms have their own accumulator set, joined once in the epilogue.
// Ceiling = dense fp16 MFMA peak / 3 = 833 TFLOP/s fp32-equivalent (the six-product scheme: 417).
//
// Structure: the packed-A LDS-DMA main loop of gemm_bf16x3.hip with 2 planes: 128x128 tile, 4 wavefronts x (2x2)
// 32x32 tiles x 2 accumulator sets (128 accumulator registers), BK = 16 per stage, stage = A block + B block =
// 16 KB, ring of NS stages (4: 64 KB, two blocks per CU).  Per k-step a wavefront issues 8 ds_read_b128 (tile
// kt+1 into the other fragment set), 4 LDS-DMA pieces (tile kt+NS) and 12 MFMAs (tile kt, from registers); one
// counted s_waitcnt vmcnt(4 (NS-2)) lgkmcnt(0) + raw s_barrier per k-step.  MFMA time per k-step halves against
// the six-product kernel, so the ring is one stage deeper to keep the same DMA lead in cycles.
#include <algorithm>
#include <cstdlib>

#include "bf16x3.h"
#include "gemm_epilogue.h"
#include "gemm_epilogue_lds.h"

namespace capdec {

typedef __attribute__((address_space(3))) void lds_void_h;
typedef const __attribute__((address_space(1))) void glb_void_h;

constexpr int H2_STAGE_B = 2 * H2_BLOCK_B;                 // 16 KB: A block + B block of one k-step
constexpr int H2_TOPK_LDS = 128 * CT_LD * 4;               // 66 048 B: logits tile of the fused lm_head epilogue
template <int NS> struct H2Geo {
    static constexpr int SMEM_B = NS * H2_STAGE_B > H2_TOPK_LDS ? NS * H2_STAGE_B : H2_TOPK_LDS;
};

// fp32 [N, K] -> two fp16 planes, TILE-MAJOR  out[tile][k_step][plane][row 0..127][16 fp16]  (rows past N zero):
// byte-for-byte the LDS image of one operand of a stage (16-B halves of a row swapped when bit 3 of the row is set)
__global__ void pack_planes_h2_kernel(const float *__restrict__ w, int ldw, _Float16 *__restrict__ out, int N, int K,
                                      int nk) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (tile, k_step, row, half): 8 k
    const int tiles = (N + GEMM_BN - 1) / GEMM_BN;
    if (i >= (size_t)tiles * nk * 256) return;
    const int hk = i & 1, r = (i >> 1) & 127;
    const size_t tk = i >> 8;                       // tile * nk + k_step
    const int ks = (int)(tk % nk), tile = (int)(tk / nk);
    const int n = tile * GEMM_BN + r;
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (n < N) {
        const float *src = w + (size_t)n * ldw + ks * 16 + hk * 8;
        v0 = reinterpret_cast<const float4 *>(src)[0];
        v1 = reinterpret_cast<const float4 *>(src)[1];
    }
    f16x4 h0, l0, h1, l1;
    split2h(v0, h0, l0);
    split2h(v1, h1, l1);
    _Float16 *dst = out + tk * (2 * 128 * 16) + r * 16 + ((hk ^ ((r >> 3) & 1)) << 3);
    reinterpret_cast<f16x4 *>(dst)[0] = h0;
    reinterpret_cast<f16x4 *>(dst)[1] = h1;
    reinterpret_cast<f16x4 *>(dst + 2048)[0] = l0;
    reinterpret_cast<f16x4 *>(dst + 2048)[1] = l1;
}

int launch_pack_planes_h2(hipStream_t st, const float *w, int ldw, int N, int K, void *out) {
    CAPDEC_CHECK(K % 64 == 0 && ldw % 4 == 0, "pack_planes_h2: K must be a multiple of 64");
    const int nk = K / X3_BK;
    const size_t tot = (size_t)((N + GEMM_BN - 1) / GEMM_BN) * nk * 256;
    hipLaunchKernelGGL(pack_planes_h2_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, w, ldw,
                       (_Float16 *)out, N, K, nk);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// The TRANSPOSE of an fp32 matrix as a packed operand: src [rows][cols] row-major (row stride ld)  ->  the two fp16 planes of
// Mt [cols][Kp] (Mt[n][k] = src[k][n]; k >= rows zero-filled up to Kp, a multiple of 64; rows n >= cols of the last 128-row
// tile zero) -- the operands of a weight-gradient product dW = dY^T X, whose K runs over the ROWS of dY and X.  One block
// per 64 x 64 tile: rows read coalesced, transposed through LDS, every thread splits one (n, k-step) = 16 consecutive k.
// (Replaces a transpose_pad launch -- fp32 written and read back -- in front of the packer, train_ops.hip.)
__global__ __launch_bounds__(256) void pack_planes_h2_t_kernel(const float *__restrict__ src, int ld, int rows, int cols,
                                                               _Float16 *__restrict__ out, int nk) {
    __shared__ float tile[64][65];
    const int t = threadIdx.x, n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    for (int i = t; i < 64 * 64; i += 256) {
        const int kk = i >> 6, nn = i & 63, k = k0 + kk, n = n0 + nn;
        tile[kk][nn] = (k < rows && n < cols) ? src[(size_t)k * ld + n] : 0.f;
    }
    __syncthreads();
    const int nn = t & 63, kq = t >> 6, n = n0 + nn;
    const int tl = n >> 7, r = n & 127, ks = (k0 >> 4) + kq;
    _Float16 *dst = out + ((size_t)tl * nk + ks) * (2 * 128 * 16) + r * 16;
#pragma unroll
    for (int hk = 0; hk < 2; ++hk) {
        float4 v0, v1;
        v0.x = tile[kq * 16 + hk * 8 + 0][nn]; v0.y = tile[kq * 16 + hk * 8 + 1][nn];
        v0.z = tile[kq * 16 + hk * 8 + 2][nn]; v0.w = tile[kq * 16 + hk * 8 + 3][nn];
        v1.x = tile[kq * 16 + hk * 8 + 4][nn]; v1.y = tile[kq * 16 + hk * 8 + 5][nn];
        v1.z = tile[kq * 16 + hk * 8 + 6][nn]; v1.w = tile[kq * 16 + hk * 8 + 7][nn];
        f16x4 h0, l0, h1, l1;
        split2h(v0, h0, l0);
        split2h(v1, h1, l1);
        _Float16 *d = dst + ((hk ^ ((r >> 3) & 1)) << 3);
        reinterpret_cast<f16x4 *>(d)[0] = h0;
        reinterpret_cast<f16x4 *>(d)[1] = h1;
        reinterpret_cast<f16x4 *>(d + 2048)[0] = l0;
        reinterpret_cast<f16x4 *>(d + 2048)[1] = l1;
    }
}
// out: x3_packed_bytes(cols, Kp, PK_F16X2) bytes
int launch_pack_planes_h2_t(hipStream_t st, const float *src, int ld, int rows, int cols, int Kp, void *out) {
    CAPDEC_CHECK(Kp % 64 == 0 && Kp >= rows && rows > 0 && cols > 0 && ld >= cols, "pack_planes_h2_t: K (padded rows) must be a multiple of 64");
    const int npad = (cols + 127) / 128 * 128;
    hipLaunchKernelGGL(pack_planes_h2_t_kernel, dim3(npad / 64, Kp / 64), dim3(256), 0, st, src, ld, rows, cols, (_Float16 *)out, Kp / X3_BK);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// s_waitcnt immediate (gfx9 encoding): vmcnt(N) expcnt(none) lgkmcnt(L: 0 = wait for all, 15 = do not wait)
__host__ __device__ constexpr int waitcnt_imm(int vm, int lgkm) {
    return (vm & 15) | (7 << 4) | ((lgkm & 15) << 8) | ((vm >> 4) << 14);
}

// TR: MFMA operands swapped (D^T), a lane ends with four consecutive columns of one row (see gemm_bf16x3.hip).
// am = sum hi_a hi_b, ac = sum (hi_a lo_b + lo_a hi_b) (weight 2^-11, applied by h2_join).
// ABL (measurement only, CAPDEC_H2_ABL; results are WRONG for ABL != 0): 1 = no s_barrier in the loop, 2 = no LDS-DMA in
// the loop, 3 = no fragment reads in the loop, 4 = neither barrier nor DMA, 5 / 6 = all blocks load the same few panels
// KIND: 0 = f16x2 (the fp32-accurate scheme above).  1 / 2 = ONE-plane fp16 / bf16 operands (formats PK_F16X1 /
// PK_BF16X1: the reduced-precision modes, one MFMA per product): the same ring, DMA pieces and fragment reads, but a
// stage holds TWO consecutive k-steps of the single plane where f16x2 holds the two planes of one k-step (both are
// 2 x 4 KB contiguous per operand), so a stage is 32 deep and carries 8 MFMAs per wavefront instead of 12.
// CONV (implicit-GEMM 3x3 convolution, stride 1, padding 1; resnet.hip / capi.hip): the A operand is not an im2col
// matrix but the ACTIVATION itself in packed form -- rows = pixels of an NHWC tensor [M = N H W][C], the very layout a
// GEMM epilogue writes with packed_out -- and K runs over (tap, channel): K = 9 C, stage ks = tap * ncs + cs.  Row m of
// the product is output pixel m, whose operand row for tap (ky, kx) is input pixel m + (ky - 1) W + (kx - 1) (stride 1:
// output and input pixels share their index) or a row of zeros outside the image.  Each thread owns the same 16-byte
// piece of the stage as in the GEMM (row t / 2, half t & 1) and re-derives its source address when the tap changes --
// nine times per tile; a piece is a 16-byte half of a 32-byte row chunk whose position inside the chunk depends on bit 3
// of the row (the LDS swizzle), so a piece copied from source row rs to tile row rd sits at half ^ sw(rd) ^ sw(rs).
struct ConvGeo {
    int H = 0, W = 0, M = 0;      // image height / width (pixels), rows of the product (= N H W)
    int ncs = 0;                  // stages per tap: C / 16 (two-plane f16x2) or C / 32 (one-plane formats)
    const char *zero = nullptr;   // >= ncs * 8 KB + 8 KB of zeros: the operand rows of the padding
};

template <bool TR, int NS, int ABL = 0, int KIND = 0, bool CONV = false>
__device__ __forceinline__ void h2p_mainloop(const _Float16 *__restrict__ Apk, const _Float16 *__restrict__ Bpk, int K,
                                             int tm, int tn, char *smem, f32x16 (&am)[2][2], f32x16 (&ac)[2][2],
                                             int ks0 = 0, int nks = -1, const ConvGeo cg = ConvGeo()) {
    static_assert(NS >= 3 && NS <= 9, "ring depth");
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l32 = lane & 31;
    const int nkf = K / (KIND == 0 ? X3_BK : 2 * X3_BK);                       // stages of the whole K (panel stride)
    const int nk = nks < 0 ? nkf : nks;                                        // stages of THIS block, even
    // (ABL 5 / 6: every block loads one of 4 x 4 / 8 x 8 panels -- a 3 MB / 6 MB working set per launch: what the loop
    //  does when (nearly) every operand piece is an L2 hit)
    const int tml = ABL == 5 ? (tm & 3) : ABL == 6 ? (tm & 7) : tm, tnl = ABL == 5 ? (tn & 3) : ABL == 6 ? (tn & 7) : tn;
    const _Float16 *ap = Apk + ((size_t)tml * nkf + ks0) * (H2_BLOCK_B / 2) + t * 8;   // this thread's 16-B piece
    const _Float16 *bp = Bpk + ((size_t)tnl * nkf + ks0) * (H2_BLOCK_B / 2) + t * 8;
    char *dst0 = smem + wave * 1024;                                           // wave-uniform LDS base of its pieces
    // CONV: running (stage, tap, stage-in-tap) of the NEXT tile to send (the DMAs are issued in increasing stage order,
    // clamped at the last one) and this thread's source for the current tap
    int cv_ks = 0, cv_cs = 0, cv_tap = 0, cv_ox = 0, cv_oy = 0, cv_m = 0;
    bool cv_ok = false;
    const char *cv_cur = nullptr;
    auto cv_src = [&](int tap) -> const char * {
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int iy = cv_oy + ky - 1, ix = cv_ox + kx - 1;
        const bool ok = cv_ok && (unsigned)iy < (unsigned)cg.H && (unsigned)ix < (unsigned)cg.W;
        const int pin = cv_m + (ky - 1) * cg.W + (kx - 1), pr = pin & 127;
        const size_t off = (size_t)(pin >> 7) * cg.ncs * H2_BLOCK_B + pr * X3_ROW_B +
                           ((((t & 1) ^ ((t >> 4) & 1) ^ ((pr >> 3) & 1))) << 4);
        return ok ? reinterpret_cast<const char *>(Apk) + off : cg.zero + ((t & 1) << 4);
    };
    if constexpr (CONV) {
        cv_m = tm * GEMM_BM + (t >> 1);
        cv_ok = cv_m < cg.M;
        if (!cv_ok) cv_m = 0;
        cv_ox = cv_m % cg.W;
        cv_oy = (cv_m / cg.W) % cg.H;
        cv_cur = cv_src(0);
    }
#define H2_DMA(stage, ks_)                                                                                     \
    {                                                                                                          \
        char *d = dst0 + (stage) * H2_STAGE_B;                                                                 \
        if constexpr (CONV) {                                                                                  \
            const char *sa = cv_cur + (size_t)cv_cs * H2_BLOCK_B;                                              \
            const _Float16 *sb = bp + (size_t)cv_ks * (H2_BLOCK_B / 2);                                        \
            _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                                    \
                __builtin_amdgcn_global_load_lds((glb_void_h *)(sa + p * X3_PLANE_B), (lds_void_h *)(d + p * X3_PLANE_B), 16, 0, 0); \
                __builtin_amdgcn_global_load_lds((glb_void_h *)(sb + p * 2048), (lds_void_h *)(d + H2_BLOCK_B + p * X3_PLANE_B), 16, 0, 0); \
            }                                                                                                  \
            if (cv_ks < nk - 1) {                                                                              \
                ++cv_ks;                                                                                       \
                if (++cv_cs == cg.ncs) { cv_cs = 0; ++cv_tap; cv_cur = cv_src(cv_tap); }                       \
            }                                                                                                  \
        } else {                                                                                               \
            const _Float16 *sa = ap + (size_t)(ks_) * (H2_BLOCK_B / 2), *sb = bp + (size_t)(ks_) * (H2_BLOCK_B / 2); \
            _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                                    \
                __builtin_amdgcn_global_load_lds((glb_void_h *)(sa + p * 2048), (lds_void_h *)(d + p * X3_PLANE_B), 16, 0, 0); \
                __builtin_amdgcn_global_load_lds((glb_void_h *)(sb + p * 2048), (lds_void_h *)(d + H2_BLOCK_B + p * X3_PLANE_B), 16, 0, 0); \
            }                                                                                                  \
        }                                                                                                      \
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { am[i][j][r] = 0.f; ac[i][j][r] = 0.f; }

    const int swz = ((half ^ ((l32 >> 3) & 1)) << 4);
    const int a_rd = (wm * 64 + l32) * X3_ROW_B + swz;
    const int b_rd = H2_BLOCK_B + (wn * 64 + l32) * X3_ROW_B + swz;

    f16x8 f0a0[2], f0a1[2], f0b0[2], f0b1[2], f1a0[2], f1a1[2], f1b0[2], f1b1[2];
#define H2_READ(F, stage)                                                                           \
    {                                                                                               \
        const char *rs = smem + (stage) * H2_STAGE_B;                                               \
        _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                             \
            F##a0[p] = *reinterpret_cast<const f16x8 *>(rs + p * X3_PLANE_B + a_rd);                 \
            F##a1[p] = *reinterpret_cast<const f16x8 *>(rs + p * X3_PLANE_B + a_rd + 32 * X3_ROW_B); \
            F##b0[p] = *reinterpret_cast<const f16x8 *>(rs + p * X3_PLANE_B + b_rd);                 \
            F##b1[p] = *reinterpret_cast<const f16x8 *>(rs + p * X3_PLANE_B + b_rd + 32 * X3_ROW_B); \
        }                                                                                           \
    }
#define H2_MM1(x, y, c) (KIND == 2 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0) \
                                   : __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0))
#define H2_MM(x, y, c) (TR ? H2_MM1(y, x, c) : H2_MM1(x, y, c))
#define H2_TERM(ACC, F, pa, pb)                                \
    ACC[0][0] = H2_MM(F##a0[pa], F##b0[pb], ACC[0][0]);        \
    ACC[0][1] = H2_MM(F##a0[pa], F##b1[pb], ACC[0][1]);        \
    ACC[1][0] = H2_MM(F##a1[pa], F##b0[pb], ACC[1][0]);        \
    ACC[1][1] = H2_MM(F##a1[pa], F##b1[pb], ACC[1][1]);
#define H2_MFMAS(F)                                                                      \
    if constexpr (KIND == 0) { H2_TERM(ac, F, 1, 0) H2_TERM(am, F, 0, 0) H2_TERM(ac, F, 0, 1) } \
    else { H2_TERM(am, F, 0, 0) H2_TERM(am, F, 1, 1) }
    // (the s_waitcnt builtin, not inline asm: the compiler's own waitcnt pass must see that the fragment reads have
    //  completed, or it puts an lgkmcnt(0) in front of the next MFMAs -- behind the freshly issued reads)
#define H2_SYNC()                                                                        \
    asm volatile("" ::: "memory");                                                       \
    if constexpr (ABL == 2 || ABL == 4) __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));   \
    else __builtin_amdgcn_s_waitcnt(waitcnt_imm(4 * (NS - 2), 0));                       \
    if constexpr (ABL != 1 && ABL != 4) __builtin_amdgcn_s_barrier();                    \
    asm volatile("" ::: "memory");
    // one memory operation in the shadow of each MFMA: 8 fragment reads, then the 4 DMA pieces
#define H2_INTERLEAVE()                                                                    \
    if constexpr (KIND == 0) {                                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                 \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   /* 1 MFMA    */           \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   /* 1 DS read */           \
        }                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                 \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                             \
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   /* 1 VMEM (LDS-DMA piece) */ \
        }                                                                                  \
    } else {   /* 8 MFMAs: 4 x (MFMA, 2 reads), 4 x (MFMA, 1 DMA piece) */                 \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                 \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                             \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                             \
        }                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                 \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                             \
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);                             \
        }                                                                                  \
    }                                                                                      \
    __builtin_amdgcn_sched_barrier(0);

    // prologue: tiles 0 .. NS-1 in flight (stage s <- tile s); tile 0 landed -> fragments f0; tile 1 landed
#pragma unroll
    for (int s = 0; s < NS; ++s) H2_DMA(s, min(s, nk - 1))
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(4 * (NS - 1), 15));    // tile 0 landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    H2_READ(f0, 0)
    H2_SYNC()                                                    // tile 1 landed, this wave's reads of stage 0 done
    // k-step kt: fragments of tile kt+1 are read from stage (kt+1)%NS, tile kt+NS is sent to stage kt%NS (whose
    // fragments every wave finished reading before the barrier that ended k-step kt-1), the MFMAs of tile kt run from
    // registers.  End of k-step: all but the NS-2 newest tiles have landed => tile kt+2 is in LDS.
    int s0 = 0;                                                  // kt % NS
    for (int kt = 0; kt < nk; kt += 2) {
        const int s1 = s0 + 1 == NS ? 0 : s0 + 1, s2 = s1 + 1 == NS ? 0 : s1 + 1;
        if constexpr (ABL != 3) H2_READ(f1, s1)                  // tile kt+1
        if constexpr (ABL != 2 && ABL != 4) H2_DMA(s0, min(kt + NS, nk - 1))   // unconditional (clamped): exact vmcnt count
        H2_MFMAS(f0)                                             // tile kt
        if constexpr (ABL == 0) { H2_INTERLEAVE() }
        H2_SYNC()
        if constexpr (ABL != 3) H2_READ(f0, s2)                  // tile kt+2
        if constexpr (ABL != 2 && ABL != 4) H2_DMA(s1, min(kt + 1 + NS, nk - 1))
        H2_MFMAS(f1)                                             // tile kt+1
        if constexpr (ABL == 0) { H2_INTERLEAVE() }
        H2_SYNC()
        s0 = s2;
    }
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 15));              // clamped tail pieces must land before LDS is reused
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#undef H2_DMA
#undef H2_READ
#undef H2_TERM
#undef H2_MM
#undef H2_MM1
#undef H2_MFMAS
#undef H2_SYNC
#undef H2_INTERLEAVE
}

// acc = am + 2^-11 ac
__device__ __forceinline__ void h2_join(f32x16 (&am)[2][2], const f32x16 (&ac)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) am[i][j][r] = fmaf(ac[i][j][r], 1.0f / H2_LO_SCALE, am[i][j][r]);
}

constexpr int H2_NS = 4;
// wave grid of the 128 x 128 kernels for the coalesced epilogues (gemm_epilogue_lds.h): 2 x 2 wavefronts of 64 x 64
struct H2Tile { static constexpr int WN = 2, TI = 2, TJ = 2; };
// end of a persistent block's tile: the next tile's LDS-DMA pieces land in the epilogue's slabs
__device__ __forceinline__ void h2_slab_release(bool more) {
    if (more) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
}
__device__ __forceinline__ EpiArgs h2_epi_args(float *C, int ldc, int M, int N, int m0, int n0, const float *bias,
                                               const float *resid, int ldr, int act, char *packed_out, int fmt,
                                               const QkvScatter *sc) {
    EpiArgs ea;
    ea.C = C; ea.ldc = ldc; ea.M = M; ea.N = N; ea.m0 = m0; ea.n0 = n0;
    ea.bias = bias; ea.act = act; ea.ldr = ldr; ea.fmt = fmt;
    ea.packed = packed_out;
    if (packed_out) ea.resid_pk = reinterpret_cast<const char *>(resid);      // (with packed_out, `resid` is a PACKED residual)
    else ea.resid = resid;
    ea.sc = sc;
    return ea;
}

template <bool VEC4, int NS, int ABL = 0>
__global__ __launch_bounds__(256, 2) void gemm_f16x2p_kernel(const _Float16 *__restrict__ Apk,
                                                             const _Float16 *__restrict__ Bpk, float *C, int ldc, int M,
                                                             int N, int K, const float *__restrict__ bias,
                                                             const float *resid, int ldr, int act, int tiles_m,
                                                             int tiles_n, char *packed_out, QkvScatter sc) {
    __shared__ __attribute__((aligned(16))) char smem[NS * H2_STAGE_B];
    // Persistent form: the launch has min(tiles, 512) blocks (two per CU) and block b walks tiles b, b + grid, ...  The
    // hardware hands the blocks of a launch out breadth-first (one per CU, then the second slots), so the tiles of a
    // partly filled last round are taken one per CU by blocks that by then have the CU to themselves -- where one block
    // per tile lets the dispatcher put the leftovers two to a CU as slots free up (600 tiles: 2 x 37 us instead of
    // 37 + 15).  grid % 8 == 0 keeps a block's tiles on its own XCD's slice of the tile order.
    const int ntiles = tiles_m * tiles_n;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int tm, tn;
        tile_coords(tiles_m, tiles_n, tm, tn, tile);
        f32x16 am[2][2], ac[2][2];
        h2p_mainloop<true, NS, ABL>(Apk, Bpk, K, tm, tn, smem, am, ac);      // ends with a barrier: the ring is free again
        h2_join(am, ac);
        if constexpr (VEC4) {
            const EpiArgs ea = h2_epi_args(C, ldc, M, N, tm * GEMM_BM, tn * GEMM_BN, bias, resid, ldr, act, packed_out, PK_F16X2, &sc);
            epilogue_lds<H2Tile>(am, smem, ea);
            h2_slab_release(tile + (int)gridDim.x < ntiles);
        } else if (packed_out) {
            epilogue_store_packed_t(am, packed_out, N >> 4, M, N, tm * GEMM_BM, tn * GEMM_BN, bias, act, PK_F16X2,
                                    reinterpret_cast<const char *>(resid));  // (with packed_out, `resid` is a PACKED residual)
        } else {
            epilogue_store_t<false>(am, C, ldc, M, N, tm * GEMM_BM, tn * GEMM_BN, bias, resid, ldr, act);
        }
    }
}

template <int KSEL>
__global__ __launch_bounds__(256, 2) void gemm_f16x2p_topk_kernel(const _Float16 *__restrict__ Apk,
                                                                  const _Float16 *__restrict__ Bpk, int M, int N, int K,
                                                                  float inv_temp, float *tile_max, float *tile_sum,
                                                                  float *cand_val, int *cand_idx, int tiles_m,
                                                                  int tiles_n) {
    __shared__ __attribute__((aligned(16))) char smem[H2Geo<H2_NS>::SMEM_B];
    int tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn);
    f32x16 am[2][2], ac[2][2];
    h2p_mainloop<false, H2_NS>(Apk, Bpk, K, tm, tn, smem, am, ac);      // ends with a barrier
    h2_join(am, ac);
    epilogue_topk<KSEL, 2, true>(am, reinterpret_cast<float *>(smem), M, N, tm * GEMM_BM, tn * GEMM_BN, tn, tiles_n, inv_temp,
                                 tile_max, tile_sum, cand_val, cand_idx);
}

// split-K for under-filled grids: raw fp32 partial tiles to a workspace [S][M][N], summed in a fixed order by
// splitk_reduce_kernel (gemm_bf16x3.hip), which also applies the epilogue
__global__ __launch_bounds__(256, 2) void gemm_f16x2p_splitk_kernel(const _Float16 *__restrict__ Apk,
                                                                    const _Float16 *__restrict__ Bpk, float *part, int M,
                                                                    int N, int K, int tiles_m, int tiles_n, int S) {
    __shared__ __attribute__((aligned(16))) char smem[H2_NS * H2_STAGE_B];
    const int ntiles = tiles_m * tiles_n;
    const int slice = blockIdx.x / ntiles;
    int tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn, blockIdx.x - slice * ntiles);
    const int nks = K / X3_BK / S;
    f32x16 am[2][2], ac[2][2];
    h2p_mainloop<true, H2_NS>(Apk, Bpk, K, tm, tn, smem, am, ac, slice * nks, nks);
    h2_join(am, ac);
    const EpiArgs ea = h2_epi_args(part + (size_t)slice * M * N, N, M, N, tm * GEMM_BM, tn * GEMM_BN, nullptr, nullptr, 0,
                                   CAPDEC_ACT_NONE, nullptr, PK_F16X2, nullptr);
    epilogue_lds_wave<2, 2, 0, CAPDEC_ACT_NONE>(am, reinterpret_cast<float *>(smem + (threadIdx.x >> 6) * EpiSlab<2>::BYTES),
                                                ea.m0 + (threadIdx.x >> 7) * 64, ea.n0 + ((threadIdx.x >> 6) & 1) * 64, ea);
}

int launch_gemm_f16x2p(hipStream_t st, const void *Apacked, const void *Bpacked, float *C, int ldc, int M, int N, int K,
                       const GemmEpilogue &epi) {
    CAPDEC_CHECK(M > 0 && N > 0 && K > 0 && K % 64 == 0, "gemm_f16x2p: K must be a multiple of 64");
    CAPDEC_CHECK(epi.packed_out == nullptr ||
                     (N % 64 == 0 && epi.resid == nullptr && ((uintptr_t)epi.bias & 15) == 0),
                 "gemm_f16x2p: packed output needs N % 64 == 0, a 16-byte aligned bias and no residual");
    const int tiles_m = (M + GEMM_BM - 1) / GEMM_BM, tiles_n = (N + GEMM_BN - 1) / GEMM_BN;
    const bool vec4 = N % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)C & 15) == 0 &&
                      (epi.bias == nullptr || ((uintptr_t)epi.bias & 15) == 0) &&
                      (epi.resid == nullptr || (epi.ldr % 4 == 0 && ((uintptr_t)epi.resid & 15) == 0));
    CAPDEC_CHECK(epi.resid_packed == nullptr || epi.packed_out != nullptr, "gemm_f16x2p: a packed residual needs a packed output");
    const float *resid_arg = epi.packed_out ? (const float *)epi.resid_packed : epi.resid;
    const QkvScatter sc = epi.qkv_scatter ? *epi.qkv_scatter : QkvScatter();
    CAPDEC_CHECK(!sc.kc || (vec4 && epi.bias && !epi.resid && !epi.packed_out && epi.act == CAPDEC_ACT_NONE && N == 3 * sc.d &&
                            sc.d % GEMM_BN == 0 && !epi.splitk_ws),
                 "gemm_f16x2p: the qkv scatter epilogue needs an unsplit, biased, plain [M, 3d] projection");
    const Tuning &tn = tuning_of(epi);
    // round-4 ping-pong kernels (gemm_pp.hip): forced (CAPDEC_H2W >= 10) or where the planner expects a gain from them
    // (mid-size launches; the small-batch regime M <= 512 keeps its batch-size independent split-K)
    if (vec4 && !epi.invariant) {
        const bool can_split = epi.splitk_ws && !epi.resid_packed && !epi.packed_out && !sc.kc;
        int which = 0;
        if (tn.h2w >= 10) which = (tn.h2w == 10 || epi.wide_ok) ? tn.h2w : 0;
        else if (tn.h2w == 1 && tn.pp && M > 4 * GEMM_BM) which = pp_plan(M, N, K, epi.wide_ok, can_split, tn.pp);
        if (which) return launch_gemm_pp(st, which, Apacked, Bpacked, C, ldc, M, N, K, epi, 1.0f / H2_LO_SCALE);
    }
    const int S = (vec4 && epi.splitk_ws && !epi.resid_packed) ? gemm_splitk_slices(M, N, K, tn) : 1;
    if (S > 1 && epi.splitk_ws_bytes >= (size_t)S * M * N * sizeof(float)) {
        float *part = (float *)epi.splitk_ws;
        hipLaunchKernelGGL(gemm_f16x2p_splitk_kernel, dim3(tiles_m * tiles_n * S), dim3(256), 0, st,
                           (const _Float16 *)Apacked, (const _Float16 *)Bpacked, part, M, N, K, tiles_m, tiles_n, S);
        CAPDEC_HIP(hipGetLastError());
        return launch_splitk_reduce(st, part, S, M, N, epi, C, ldc, PK_F16X2);
    }
    if (vec4 && epi.wide_ok && tn.h2w >= 1 && tn.h2w < 10 && !sc.kc) {      // round-3 single-accumulator geometries where they remove a round
        const int which = tn.h2w >= 2 ? tn.h2w : h2w_plan(M, N, K);
        if (which) return launch_gemm_h2w(st, which, Apacked, Bpacked, C, ldc, M, N, K, epi, 1.0f / H2_LO_SCALE);
    }
    // persistent form for grids of up to four rounds, where the partly filled last round matters (625 captions: mlp.c_fc
    // 600 tiles, 80 -> 70 us inside the decode loop); larger grids keep one block per tile (the dispatcher balances them:
    // within noise either way at 25 000 rows).  CAPDEC_H2_PERSIST=<blocks> (0 = never)
    const int persist = tn.h2_persist;
    const int grid_h2 = (persist > 0 && tiles_m * tiles_n <= 4 * persist) ? std::min(tiles_m * tiles_n, persist) : tiles_m * tiles_n;
#define LAUNCH_H2(V4, NSV)                                                                                            \
    hipLaunchKernelGGL((gemm_f16x2p_kernel<V4, NSV>), dim3(grid_h2), dim3(256), 0, st, (const _Float16 *)Apacked, \
                       (const _Float16 *)Bpacked, C, ldc, M, N, K, epi.bias, resid_arg, epi.ldr, epi.act, tiles_m,      \
                       tiles_n, (char *)epi.packed_out, sc)
#ifdef CAPDEC_MEASURE
    // CAPDEC_H2_ABL 1..6: ablations of the main loop (WRONG results); CAPDEC_H2_NS: ring depth 3 / 5
    if (vec4 && tn.h2_abl >= 1 && tn.h2_abl <= 6) {
#define LAUNCH_H2A(A)                                                                                               \
    hipLaunchKernelGGL((gemm_f16x2p_kernel<true, H2_NS, A>), dim3(tiles_m * tiles_n), dim3(256), 0, st,               \
                       (const _Float16 *)Apacked, (const _Float16 *)Bpacked, C, ldc, M, N, K, epi.bias, epi.resid,   \
                       epi.ldr, epi.act, tiles_m, tiles_n, (char *)epi.packed_out, sc)
        switch (tn.h2_abl) {
            case 1: LAUNCH_H2A(1); break;
            case 2: LAUNCH_H2A(2); break;
            case 3: LAUNCH_H2A(3); break;
            case 4: LAUNCH_H2A(4); break;
            case 5: LAUNCH_H2A(5); break;
            default: LAUNCH_H2A(6);
        }
#undef LAUNCH_H2A
        CAPDEC_HIP(hipGetLastError());
        return 0;
    }
    if (vec4 && tn.h2_ns == 3) { LAUNCH_H2(true, 3); CAPDEC_HIP(hipGetLastError()); return 0; }
    if (vec4 && tn.h2_ns == 5) { LAUNCH_H2(true, 5); CAPDEC_HIP(hipGetLastError()); return 0; }
#endif
    if (vec4) LAUNCH_H2(true, H2_NS);
    else LAUNCH_H2(false, H2_NS);
#undef LAUNCH_H2
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

int launch_gemm_f16x2p_topk(hipStream_t st, const void *Apacked, const void *Bpacked, int M, int N, int K, int k,
                            float inv_temp, float *tile_max, float *tile_sum, float *cand_val, int *cand_idx) {
    CAPDEC_CHECK(M > 0 && N > 0 && K > 0 && K % 64 == 0, "gemm_f16x2p_topk: K must be a multiple of 64");
    const int tiles_m = (M + GEMM_BM - 1) / GEMM_BM, tiles_n = (N + GEMM_BN - 1) / GEMM_BN;
    dim3 grid(tiles_m * tiles_n), block(256);
#define LAUNCH_TOPKH(KS)                                                                                          \
    hipLaunchKernelGGL(gemm_f16x2p_topk_kernel<KS>, grid, block, 0, st, (const _Float16 *)Apacked,                 \
                       (const _Float16 *)Bpacked, M, N, K, inv_temp, tile_max, tile_sum, cand_val, cand_idx, tiles_m, tiles_n)
    switch (k) {
        case 1: LAUNCH_TOPKH(1); break;
        case 2: LAUNCH_TOPKH(2); break;
        case 3: LAUNCH_TOPKH(3); break;
        case 4: LAUNCH_TOPKH(4); break;
        case 5: LAUNCH_TOPKH(5); break;
        case 6: LAUNCH_TOPKH(6); break;
        case 7: LAUNCH_TOPKH(7); break;
        case 8: LAUNCH_TOPKH(8); break;
        default: CAPDEC_CHECK(false, "gemm_topk: k must be in 1..8");
    }
#undef LAUNCH_TOPKH
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// ---- one-plane (reduced-precision) kernels on the same main loop: KIND 1 = fp16 operands, 2 = bf16 operands
// (NS = 4: 64 KB, two blocks per CU; NS = 3: 48 KB and <= 168 registers, THREE blocks per CU -- CAPDEC_X1_NS picks)
template <bool VEC4, int KIND, int NS>
__global__ __launch_bounds__(256, (NS == 3 ? 3 : 2)) void gemm_x1_kernel(const _Float16 *__restrict__ Apk, const _Float16 *__restrict__ Bpk,
                                                         float *C, int ldc, int M, int N, int K,
                                                         const float *__restrict__ bias, const float *resid, int ldr,
                                                         int act, int tiles_m, int tiles_n, char *packed_out, int out_fmt,
                                                         QkvScatter sc) {
    __shared__ __attribute__((aligned(16))) char smem[NS * H2_STAGE_B];
    const int ntiles = tiles_m * tiles_n;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {          // persistent form: see gemm_f16x2p_kernel
        int tm, tn;
        tile_coords(tiles_m, tiles_n, tm, tn, tile);
        f32x16 am[2][2], ac[2][2];
        h2p_mainloop<true, NS, 0, KIND>(Apk, Bpk, K, tm, tn, smem, am, ac);
        if constexpr (VEC4) {
            const EpiArgs ea = h2_epi_args(C, ldc, M, N, tm * GEMM_BM, tn * GEMM_BN, bias, resid, ldr, act, packed_out, out_fmt, &sc);
            epilogue_lds<H2Tile>(am, smem, ea);
            h2_slab_release(tile + (int)gridDim.x < ntiles);
        } else if (packed_out) {
            epilogue_store_packed_t(am, packed_out, N >> 4, M, N, tm * GEMM_BM, tn * GEMM_BN, bias, act, out_fmt,
                                    reinterpret_cast<const char *>(resid));  // (with packed_out, `resid` is a PACKED residual)
        } else {
            epilogue_store_t<false>(am, C, ldc, M, N, tm * GEMM_BM, tn * GEMM_BN, bias, resid, ldr, act);
        }
    }
}

template <int KSEL, int KIND>
__global__ __launch_bounds__(256, 2) void gemm_x1_topk_kernel(const _Float16 *__restrict__ Apk,
                                                              const _Float16 *__restrict__ Bpk, int M, int N, int K,
                                                              float inv_temp, float *tile_max, float *tile_sum,
                                                              float *cand_val, int *cand_idx, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) char smem[H2Geo<H2_NS>::SMEM_B];
    int tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn);
    f32x16 am[2][2], ac[2][2];
    h2p_mainloop<false, H2_NS, 0, KIND>(Apk, Bpk, K, tm, tn, smem, am, ac);      // ends with a barrier
    epilogue_topk<KSEL, 2, true>(am, reinterpret_cast<float *>(smem), M, N, tm * GEMM_BM, tn * GEMM_BN, tn, tiles_n, inv_temp,
                                 tile_max, tile_sum, cand_val, cand_idx);
}

// the same kernel (k = 5) over *m_dev rows of a compacted A operand: the exact second pass of the fused lm_head in the
// one-plane modes (decode.hip: lm_head_select; two-plane form: gemm_h2w.hip: gemm_h2w_topk_dev_kernel)
template <int KIND>
__global__ __launch_bounds__(256, 2) void gemm_x1_topk_dev_kernel(const _Float16 *__restrict__ Apk,
                                                                  const _Float16 *__restrict__ Bpk,
                                                                  const int *__restrict__ m_dev, int N, int K, float inv_temp,
                                                                  float *tile_max, float *tile_sum, float *cand_val,
                                                                  int *cand_idx, int tiles_n) {
    __shared__ __attribute__((aligned(16))) char smem[H2Geo<H2_NS>::SMEM_B];
    const int M = *m_dev;
    const int tiles_m = (M + GEMM_BM - 1) / GEMM_BM, ntiles = tiles_m * tiles_n;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int tm, tn;
        tile_coords(tiles_m, tiles_n, tm, tn, tile);
        f32x16 am[2][2], ac[2][2];
        h2p_mainloop<false, H2_NS, 0, KIND>(Apk, Bpk, K, tm, tn, smem, am, ac);      // ends with a barrier
        epilogue_topk<5, 2, true>(am, reinterpret_cast<float *>(smem), M, N, tm * GEMM_BM, tn * GEMM_BN, tn, tiles_n, inv_temp,
                                  tile_max, tile_sum, cand_val, cand_idx);
        __syncthreads();                                                             // the ring is free again
    }
}

int launch_gemm_x1_topk_dev(hipStream_t st, const void *Apacked, const void *Bpacked, const int *m_dev, int N, int K,
                            float inv_temp, float *tile_max, float *tile_sum, float *cand_val, int *cand_idx, int fmt) {
    CAPDEC_CHECK(m_dev && N > 0 && K > 0 && K % 64 == 0, "gemm_x1_topk_dev: bad argument");
    CAPDEC_CHECK(fmt == PK_F16X1 || fmt == PK_BF16X1, "gemm_x1_topk_dev: one-plane operand formats only");
    const int tiles_n = (N + GEMM_BN - 1) / GEMM_BN;
    if (fmt == PK_F16X1)
        hipLaunchKernelGGL((gemm_x1_topk_dev_kernel<1>), dim3(512), dim3(256), 0, st, (const _Float16 *)Apacked,
                           (const _Float16 *)Bpacked, m_dev, N, K, inv_temp, tile_max, tile_sum, cand_val, cand_idx, tiles_n);
    else
        hipLaunchKernelGGL((gemm_x1_topk_dev_kernel<2>), dim3(512), dim3(256), 0, st, (const _Float16 *)Apacked,
                           (const _Float16 *)Bpacked, m_dev, N, K, inv_temp, tile_max, tile_sum, cand_val, cand_idx, tiles_n);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// split-K of the one-plane kernels (same regimes and reduce pass as the two-plane kernel: gemm_splitk_slices)
template <int KIND>
__global__ __launch_bounds__(256, 2) void gemm_x1_splitk_kernel(const _Float16 *__restrict__ Apk,
                                                                const _Float16 *__restrict__ Bpk, float *part, int M, int N,
                                                                int K, int tiles_m, int tiles_n, int S) {
    __shared__ __attribute__((aligned(16))) char smem[H2_NS * H2_STAGE_B];
    const int ntiles = tiles_m * tiles_n;
    const int slice = blockIdx.x / ntiles;
    int tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn, blockIdx.x - slice * ntiles);
    const int nks = K / (2 * X3_BK) / S;                 // stages (two k-steps each) of this slice
    f32x16 am[2][2], ac[2][2];
    h2p_mainloop<true, H2_NS, 0, KIND>(Apk, Bpk, K, tm, tn, smem, am, ac, slice * nks, nks);
    const EpiArgs ea = h2_epi_args(part + (size_t)slice * M * N, N, M, N, tm * GEMM_BM, tn * GEMM_BN, nullptr, nullptr, 0,
                                   CAPDEC_ACT_NONE, nullptr, PK_F16X2, nullptr);
    epilogue_lds_wave<2, 2, 0, CAPDEC_ACT_NONE>(am, reinterpret_cast<float *>(smem + (threadIdx.x >> 6) * EpiSlab<2>::BYTES),
                                                ea.m0 + (threadIdx.x >> 7) * 64, ea.n0 + ((threadIdx.x >> 6) & 1) * 64, ea);
}

// fmt = PK_F16X1 or PK_BF16X1 (both operands; a packed output is written in the same format)
int launch_gemm_x1(hipStream_t st, const void *Apacked, const void *Bpacked, float *C, int ldc, int M, int N, int K,
                   const GemmEpilogue &epi, int fmt) {
    CAPDEC_CHECK(M > 0 && N > 0 && K > 0 && K % 64 == 0, "gemm_x1: K must be a multiple of 64");
    CAPDEC_CHECK(fmt == PK_F16X1 || fmt == PK_BF16X1, "gemm_x1: one-plane operand formats only");
    CAPDEC_CHECK(epi.packed_out == nullptr ||
                     (N % 64 == 0 && epi.resid == nullptr && ((uintptr_t)epi.bias & 15) == 0),
                 "gemm_x1: packed output needs N % 64 == 0, a 16-byte aligned bias and no residual");
    const int tiles_m = (M + GEMM_BM - 1) / GEMM_BM, tiles_n = (N + GEMM_BN - 1) / GEMM_BN;
    const bool vec4 = N % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)C & 15) == 0 &&
                      (epi.bias == nullptr || ((uintptr_t)epi.bias & 15) == 0) &&
                      (epi.resid == nullptr || (epi.ldr % 4 == 0 && ((uintptr_t)epi.resid & 15) == 0));
    // (Round 5 measured the ping-pong tiles of gemm_pp.hip under one-plane operands -- 256 x 256 / 256 x 192 / 256 x 128,
    //  rings of 4-5 and of 5-6 stages -- against this kernel inside the decode loop: 12-15 % SLOWER per launch at 5000 rows,
    //  25 % at 25 000 (profiles/r5_x1_pingpong_ab.txt); the variant was removed again.)
    const QkvScatter sc = epi.qkv_scatter ? *epi.qkv_scatter : QkvScatter();
    CAPDEC_CHECK(!sc.kc || (vec4 && epi.bias && !epi.resid && !epi.packed_out && epi.act == CAPDEC_ACT_NONE && N == 3 * sc.d &&
                            sc.d % GEMM_BN == 0 && !epi.splitk_ws),
                 "gemm_x1: the qkv scatter epilogue needs an unsplit, biased, plain [M, 3d] projection");
    {
        int S = (vec4 && epi.splitk_ws && !epi.resid_packed) ? gemm_splitk_slices(M, N, K, tuning_of(epi)) : 1;
        if (S > 1 && ((K / X3_BK / S) % 4 != 0 || epi.splitk_ws_bytes < (size_t)S * M * N * sizeof(float))) S = 1;
        if (S > 1) {     // a slice is a whole, even number of two-k-step stages
            float *part = (float *)epi.splitk_ws;
            if (fmt == PK_F16X1)
                hipLaunchKernelGGL(gemm_x1_splitk_kernel<1>, dim3(tiles_m * tiles_n * S), dim3(256), 0, st,
                                   (const _Float16 *)Apacked, (const _Float16 *)Bpacked, part, M, N, K, tiles_m, tiles_n, S);
            else
                hipLaunchKernelGGL(gemm_x1_splitk_kernel<2>, dim3(tiles_m * tiles_n * S), dim3(256), 0, st,
                                   (const _Float16 *)Apacked, (const _Float16 *)Bpacked, part, M, N, K, tiles_m, tiles_n, S);
            CAPDEC_HIP(hipGetLastError());
            return launch_splitk_reduce(st, part, S, M, N, epi, C, ldc, fmt);
        }
    }
    // three blocks per CU measured +3 % at 25 000 rows, +1.3 % on the greedy bf16 workload; CAPDEC_X1_NS=4: two blocks
    const int ns3 = tuning_of(epi).x1_ns == 4 ? 0 : 1;
    // persistent blocks for grids of up to four rounds (768 slots with three blocks per CU, 512 with two)
    const int persist = tuning_of(epi).h2_persist;
    const int slots = persist > 0 ? ((vec4 && ns3) ? persist * 3 / 2 : persist) : 0;
    const int grid_x1 = (slots > 0 && tiles_m * tiles_n <= 4 * slots) ? std::min(tiles_m * tiles_n, slots) : tiles_m * tiles_n;
#define LAUNCH_X1V(V4, KD, NSV)                                                                                         \
    hipLaunchKernelGGL((gemm_x1_kernel<V4, KD, NSV>), dim3(grid_x1), dim3(256), 0, st, (const _Float16 *)Apacked, \
                       (const _Float16 *)Bpacked, C, ldc, M, N, K, epi.bias,                                           \
                       epi.packed_out ? (const float *)epi.resid_packed : epi.resid, epi.ldr, epi.act, tiles_m,       \
                       tiles_n, (char *)epi.packed_out, fmt, sc)
#define LAUNCH_X1K(KD)                                                                        \
    if (vec4) { if (ns3) LAUNCH_X1V(true, KD, 3); else LAUNCH_X1V(true, KD, 4); }             \
    else LAUNCH_X1V(false, KD, 4)
    if (fmt == PK_F16X1) { LAUNCH_X1K(1); } else { LAUNCH_X1K(2); }
#undef LAUNCH_X1K
#undef LAUNCH_X1V
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

int launch_gemm_x1_topk(hipStream_t st, const void *Apacked, const void *Bpacked, int M, int N, int K, int k,
                        float inv_temp, float *tile_max, float *tile_sum, float *cand_val, int *cand_idx, int fmt) {
    CAPDEC_CHECK(M > 0 && N > 0 && K > 0 && K % 64 == 0, "gemm_x1_topk: K must be a multiple of 64");
    CAPDEC_CHECK(fmt == PK_F16X1 || fmt == PK_BF16X1, "gemm_x1_topk: one-plane operand formats only");
    const int tiles_m = (M + GEMM_BM - 1) / GEMM_BM, tiles_n = (N + GEMM_BN - 1) / GEMM_BN;
    dim3 grid(tiles_m * tiles_n), block(256);
#define LAUNCH_TOPKX(KS)                                                                                              \
    if (fmt == PK_F16X1)                                                                                              \
        hipLaunchKernelGGL((gemm_x1_topk_kernel<KS, 1>), grid, block, 0, st, (const _Float16 *)Apacked,                \
                           (const _Float16 *)Bpacked, M, N, K, inv_temp, tile_max, tile_sum, cand_val, cand_idx, tiles_m, tiles_n); \
    else                                                                                                              \
        hipLaunchKernelGGL((gemm_x1_topk_kernel<KS, 2>), grid, block, 0, st, (const _Float16 *)Apacked,                \
                           (const _Float16 *)Bpacked, M, N, K, inv_temp, tile_max, tile_sum, cand_val, cand_idx, tiles_m, tiles_n)
    switch (k) {
        case 1: LAUNCH_TOPKX(1); break;
        case 2: LAUNCH_TOPKX(2); break;
        case 3: LAUNCH_TOPKX(3); break;
        case 4: LAUNCH_TOPKX(4); break;
        case 5: LAUNCH_TOPKX(5); break;
        case 6: LAUNCH_TOPKX(6); break;
        case 7: LAUNCH_TOPKX(7); break;
        case 8: LAUNCH_TOPKX(8); break;
        default: CAPDEC_CHECK(false, "gemm_topk: k must be in 1..8");
    }
#undef LAUNCH_TOPKX
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// ---- implicit-GEMM 3x3 convolution (stride 1, padding 1) on the same main loop (CONV above).  act_pk = the input
// activation [M = N H W pixels][Cin] as a packed operand (format fmt); Bpacked = the folded weights [Cout][9 Cin] in
// (ky, kx, c) order; output as for the GEMMs (fp32 C with bias / activation / residual, or packed_out)
template <int KIND>
__global__ __launch_bounds__(256, 2) void conv3x3_packed_kernel(const _Float16 *__restrict__ Apk,
                                                                const _Float16 *__restrict__ Bpk, float *C, int ldc, int M,
                                                                int N, int K, const float *__restrict__ bias,
                                                                const float *resid, int ldr, int act, int tiles_m,
                                                                int tiles_n, char *packed_out, int fmt, ConvGeo cg) {
    __shared__ __attribute__((aligned(16))) char smem[H2_NS * H2_STAGE_B];
    int tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn);
    f32x16 am[2][2], ac[2][2];
    h2p_mainloop<true, H2_NS, 0, KIND, true>(Apk, Bpk, K, tm, tn, smem, am, ac, 0, -1, cg);
    if constexpr (KIND == 0) h2_join(am, ac);
    const EpiArgs ea = h2_epi_args(C, ldc, M, N, tm * GEMM_BM, tn * GEMM_BN, bias, packed_out ? nullptr : resid, ldr, act, packed_out,
                                   fmt, nullptr);
    epilogue_lds<H2Tile>(am, smem, ea);
}

int launch_conv3x3_packed(hipStream_t st, const void *act_pk, const void *Bpacked, float *C, int ldc, int Nimg, int H,
                          int W, int Cin, int Cout, const GemmEpilogue &epi, int fmt, const void *zeros,
                          size_t zero_bytes) {
    CAPDEC_CHECK(fmt == PK_F16X2 || fmt == PK_F16X1 || fmt == PK_BF16X1, "conv3x3: operand format f16x2, f16 or bf16");
    CAPDEC_CHECK(Nimg > 0 && H > 0 && W > 0 && Cin % 64 == 0 && Cout % 4 == 0, "conv3x3: Cin % 64 == 0, Cout % 4 == 0");
    CAPDEC_CHECK((size_t)Nimg * H * W < ((size_t)1 << 31) - 256, "conv3x3: too many pixels for one launch");
    const int M = Nimg * H * W, N = Cout, K = 9 * Cin;
    ConvGeo cg;
    cg.H = H; cg.W = W; cg.M = M;
    cg.ncs = Cin / (fmt == PK_F16X2 ? 16 : 32);
    cg.zero = (const char *)zeros;
    CAPDEC_CHECK(zeros && zero_bytes >= (size_t)(cg.ncs + 1) * H2_BLOCK_B, "conv3x3: zero block too small");
    CAPDEC_CHECK(epi.packed_out == nullptr || (N % 64 == 0 && epi.resid == nullptr && ((uintptr_t)epi.bias & 15) == 0),
                 "conv3x3: packed output needs Cout % 64 == 0, a 16-byte aligned bias and no residual");
    CAPDEC_CHECK(epi.packed_out != nullptr ||
                     (ldc % 4 == 0 && ((uintptr_t)C & 15) == 0 && (epi.bias == nullptr || ((uintptr_t)epi.bias & 15) == 0) &&
                      (epi.resid == nullptr || (epi.ldr % 4 == 0 && ((uintptr_t)epi.resid & 15) == 0))),
                 "conv3x3: 16-byte aligned output rows, bias and residual");
    const int tiles_m = (M + GEMM_BM - 1) / GEMM_BM, tiles_n = (N + GEMM_BN - 1) / GEMM_BN;
#define LAUNCH_CONV(KD)                                                                                                  \
    hipLaunchKernelGGL((conv3x3_packed_kernel<KD>), dim3(tiles_m * tiles_n), dim3(256), 0, st, (const _Float16 *)act_pk,   \
                       (const _Float16 *)Bpacked, C, ldc, M, N, K, epi.bias, epi.resid, epi.ldr, epi.act, tiles_m, tiles_n, \
                       (char *)epi.packed_out, fmt, cg)
    if (fmt == PK_F16X2) LAUNCH_CONV(0); else if (fmt == PK_F16X1) LAUNCH_CONV(1); else LAUNCH_CONV(2);
#undef LAUNCH_CONV
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

CAPDEC_SAT_ACCESSOR(sat_count_gemm_f16x2)

}  // namespace capdec
