// Epilogues of the wide-tile GEMM kernels (gemm_h2w.hip, gemm_pp.hip): a wavefront owns TI x TJ blocks of 32 x 32 of
// the block tile; G supplies WN (wavefronts along N), TI, TJ (and BM / BN / NW / SMEM_B for the top-k epilogue).
// The two DIRECT store epilogues below (round 3: 16-byte / 8-byte pieces of 64 different rows per instruction) are what
// gemm_epilogue_lds.h replaced; they remain for the A/B of the measurement build (gemm_pp.hip, CAPDEC_PP_ABL=8).
#pragma once
#include "gemm_epilogue.h"

namespace capdec {

// ---- epilogues of the TR layout for a TI x TJ wave tile (generalised epilogue_store_t / epilogue_store_packed_t)
template <class G>
__device__ __forceinline__ void epilogue_store_tw(const f32x16 (&acc)[G::TI][G::TJ], float scale, float *C, int ldc, int M,
                                                  int N, int m0, int n0, const float *__restrict__ bias,
                                                  const float *resid, int ldr, int act) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / G::WN, wn = wave % G::WN, half = lane >> 5, l32 = lane & 31;
#pragma unroll
    for (int i = 0; i < G::TI; ++i) {
        const int row = m0 + (wm * G::TI + i) * 32 + l32;
        if (row >= M) continue;
#pragma unroll
        for (int j = 0; j < G::TJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = n0 + (wn * G::TJ + j) * 32 + 8 * g + 4 * half;
                if (col >= N) continue;
                float4 v = acc_quad(acc[i][j], g);
                v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
                if (bias) {
                    const float4 b = *reinterpret_cast<const float4 *>(bias + col);
                    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                }
                v.x = act_apply(v.x, act); v.y = act_apply(v.y, act);
                v.z = act_apply(v.z, act); v.w = act_apply(v.w, act);
                if (resid) {
                    const float4 r4 = *reinterpret_cast<const float4 *>(resid + (size_t)row * ldr + col);
                    v.x = post_resid(v.x + r4.x, act); v.y = post_resid(v.y + r4.y, act);
                    v.z = post_resid(v.z + r4.z, act); v.w = post_resid(v.w + r4.w, act);
                }
                *reinterpret_cast<float4 *>(C + (size_t)row * ldc + col) = v;
            }
    }
}

template <class G>
__device__ __forceinline__ void epilogue_store_packed_tw(const f32x16 (&acc)[G::TI][G::TJ], float scale, char *packed,
                                                         int nk_out, int M, int N, int m0, int n0,
                                                         const float *__restrict__ bias, int act,
                                                         const char *resid_pk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / G::WN, wn = wave % G::WN, half = lane >> 5, l32 = lane & 31;
#pragma unroll
    for (int i = 0; i < G::TI; ++i) {
        const int row = m0 + (wm * G::TI + i) * 32 + l32;
        if (row >= M) continue;
#pragma unroll
        for (int j = 0; j < G::TJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = n0 + (wn * G::TJ + j) * 32 + 8 * g + 4 * half;
                if (col >= N) continue;
                float4 v = acc_quad(acc[i][j], g);
                v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
                if (bias) {
                    const float4 b = *reinterpret_cast<const float4 *>(bias + col);
                    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                }
                v.x = act_apply(v.x, act); v.y = act_apply(v.y, act);
                v.z = act_apply(v.z, act); v.w = act_apply(v.w, act);
                if (resid_pk) {
                    const float4 rr = x3_load_quad(resid_pk, nk_out, row, col >> 4, (col >> 2) & 3, PK_F16X2);
                    v.x = post_resid(v.x + rr.x, act); v.y = post_resid(v.y + rr.y, act);
                    v.z = post_resid(v.z + rr.z, act); v.w = post_resid(v.w + rr.w, act);
                }
                x3_store_quad(packed, nk_out, row, col >> 4, (col >> 2) & 3, v, PK_F16X2);
            }
    }
}

// lm_head epilogue on a G::BM x 128 block tile (non-TR accumulator layout: acc[i][j][r] = C[wm TI 32 + i 32 + (r & 3) +
// 8 (r >> 2) + 4 (lane >> 5)][wn TJ 32 + j 32 + (lane & 31)]): the logits go through LDS in slabs of 128 rows (66 KB,
// re-using the ring) and leave as per-(row, 128-column tile) max, sum exp(x - max) and top-k (value, column) -- the
// arithmetic and the tie rules of epilogue_topk (gemm_epilogue.h), so the merge kernel sees the same partial lists
// whichever tile height produced them.
template <class G, int KSEL>
__device__ __forceinline__ void epilogue_topk_w(const f32x16 (&acc)[G::TI][G::TJ], float scale, float *Ct, int M, int N,
                                                int m0, int n0, int tn, int tiles_n, float *tile_max, float *tile_sum,
                                                float *cand_val, int *cand_idx) {
    static_assert(G::BN == 128 && G::NW == 4 && G::SMEM_B >= 128 * CT_LD * 4, "top-k epilogue: 128-column tiles, 4 waves");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / G::WN, wn = wave % G::WN, half = lane >> 5, l32 = lane & 31;
    const int grp = lane >> 4, sub = lane & 15;
    constexpr int SLABS = G::BM / 128, WROWS = G::TI * 32;       // rows of one wavefront's tile
    for (int hh = 0; hh < SLABS; ++hh) {
        if ((wm * WROWS) / 128 == hh) {
            const int r0 = (wm * WROWS) % 128;
#pragma unroll
            for (int i = 0; i < G::TI; ++i)
#pragma unroll
                for (int j = 0; j < G::TJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = r0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        Ct[row * CT_LD + (wn * G::TJ + j) * 32 + l32] = acc[i][j][r] * scale;
                    }
        }
        __syncthreads();
        for (int it = 0; it < 8; ++it) {
            const int rl = wave * 32 + it * 4 + grp;              // row within the slab
            const int row = m0 + hh * 128 + rl;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int cl = sub + 16 * j;
                v[j] = (n0 + cl < N) ? Ct[rl * CT_LD + cl] : -INFINITY;
            }
            float mx = v[0];
#pragma unroll
            for (int j = 1; j < 8; ++j) mx = fmaxf(mx, v[j]);
            mx = row16_max(mx);
            float se = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) se += __expf(v[j] - mx);
            se = row16_sum(se);
            const size_t tbase = (size_t)row * tiles_n + tn;
            if (row < M && sub == 0) {
                tile_max[tbase] = mx;
                tile_sum[tbase] = se;
            }
            row_tile_topk<KSEL>(v, mx, sub, n0, row < M, tbase, cand_val, cand_idx);
        }
        __syncthreads();
    }
}

}  // namespace capdec
