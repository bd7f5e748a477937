// Shared pieces of the MFMA GEMM kernels (native-f32 and split-bf16 main loops): tile order,
// activations, and the two epilogues operating on the 2x2 x (32x32) accumulator layout of one
// wavefront (C/D layout of v_mfma_*_32x32: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).
#pragma once
#include "bf16x3.h"
#include "common.h"

namespace capdec {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int CT_LD = 129;   // row stride (floats) of the LDS logits tile in the top-k epilogue

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case CAPDEC_ACT_TANH: return tanhf(v);
        case CAPDEC_ACT_RELU: return fmaxf(v, 0.f);
        case CAPDEC_ACT_GELU_NEW: {
            // transformers NewGELUActivation: 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3).  Evaluated through
            // the identity 0.5 (1 + tanh u) = sigmoid(2u) = 1 / (1 + exp(-2u)): one v_exp_f32 and one v_rcp_f32 (about ten
            // VALU instructions where ocml's tanhf takes ~50 -- 64 of them per thread sit in the epilogue of mlp.c_fc),
            // and no cancellation in 1 + tanh(u) for negative u: relative error a few ulp over the whole range
            const float c2 = 2.0f * 0.7978845608028654f;
            const float u2 = v * (c2 + (c2 * 0.044715f) * v * v);
            return __fdividef(v, 1.f + __expf(-u2));
        }
        case CAPDEC_ACT_QUICK_GELU: return __fdividef(v, 1.f + __expf(-1.702f * v));   // x * sigmoid(1.702 x)
        default: return v;
    }
}

// CAPDEC_ACT_RESID_RELU: the activation comes AFTER the residual add (act_apply leaves the value alone for this code)
__device__ __forceinline__ float post_resid(float v, int act) { return act == CAPDEC_ACT_RESID_RELU ? fmaxf(v, 0.f) : v; }

// XCD-aware, L2-friendly tile order: consecutive ids of one XCD walk 8 M-tiles per N-tile.
__device__ __forceinline__ void tile_coords(int tiles_m, int tiles_n, int &tm, int &tn, int block_id = -1) {
    constexpr int GM = 8;   // (2..32 measured within noise: neither panel set fits the 4 MB L2 of an XCD)
    const int nwg = tiles_m * tiles_n;
    int id = block_id < 0 ? (int)blockIdx.x : block_id;
    {   // bijective XCD remap: blocks b, b+8, b+16.. (same XCD) get contiguous ids
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, idx = id >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int per_group = GM * tiles_n;
    const int group = id / per_group;
    const int first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int in_g = id - group * per_group;
    tm = first_m + in_g % gsz;
    tn = in_g / gsz;
}

// Wave layouts of the 4-wavefront block: WMG = 2 -> 2x2 waves, 128x128 tile, 64x64 per wave (NJ = 2 column
// blocks); WMG = 1 -> 1x4 waves, 64x128 tile, 64x32 per wave (NJ = 1): twice the blocks for small M.
template <int WMG> struct WaveGrid {
    static constexpr int NJ = WMG;                 // 32-column blocks per wave
    static constexpr int BM = 64 * WMG;
    __device__ static int wm(int wave) { return WMG == 2 ? wave >> 1 : 0; }
    __device__ static int wn(int wave) { return WMG == 2 ? wave & 1 : wave; }
};

// C = act(acc + bias) + resid for the tile at (m0, n0)
template <int WMG = 2>
__device__ __forceinline__ void epilogue_store(const f32x16 (&acc)[2][WaveGrid<WMG>::NJ], float *C, int ldc, int M,
                                               int N, int m0, int n0, const float *__restrict__ bias,
                                               const float *resid, int ldr, int act) {
    constexpr int NJ = WaveGrid<WMG>::NJ;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = WaveGrid<WMG>::wm(wave), wn = WaveGrid<WMG>::wn(wave), half = lane >> 5, l32 = lane & 31;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int col = n0 + wn * 32 * NJ + j * 32 + l32;
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < M) {
                    float v = act_apply(acc[i][j][r] + bv, act);
                    if (resid) v = post_resid(v + resid[(size_t)row * ldr + col], act);
                    C[(size_t)row * ldc + col] = v;
                }
            }
        }
    }
}

// ---- epilogues of the TRANSPOSED accumulator layout (packed-A kernel: MFMA operands swapped, 128x128 tile, 2x2
// waves): acc[i][j][r] = C[m0 + wm 64 + i 32 + (lane & 31)][n0 + wn 64 + j 32 + 8 (r >> 2) + 4 (lane >> 5) + (r & 3)],
// i.e. four consecutive columns per (lane, r >> 2).
__device__ __forceinline__ float4 acc_quad(const f32x16 &a, int g) {
    return make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
}

// C = act(acc + bias) + resid, 16 float4 stores per lane (vec4: every row segment 16-byte aligned, N % 4 == 0)
template <bool VEC4>
__device__ __forceinline__ void epilogue_store_t(const f32x16 (&acc)[2][2], float *C, int ldc, int M, int N, int m0,
                                                 int n0, const float *__restrict__ bias, const float *resid, int ldr,
                                                 int act) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l32 = lane & 31;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = m0 + wm * 64 + i * 32 + l32;
        if (row >= M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = n0 + wn * 64 + j * 32 + 8 * g + 4 * half;
                if (col >= N) continue;
                float4 v = acc_quad(acc[i][j], g);
                if constexpr (VEC4) {
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
                } else {
                    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (col + q < N) {
                            float x = act_apply(e[q] + (bias ? bias[col + q] : 0.f), act);
                            if (resid) x = post_resid(x + resid[(size_t)row * ldr + col + q], act);
                            C[(size_t)row * ldc + col + q] = x;
                        }
                }
            }
    }
}

// (round 4: the K / V scatter of the decode-step qkv projection lives in gemm_epilogue_lds.h, MODE 2: whole 256-byte keys /
//  values per instruction instead of 16-byte pieces of 64 different rows)

// act(acc + bias) written as the packed split-bf16 A operand of the NEXT GEMM (its K = this GEMM's N, N % 16 == 0):
// a lane's four consecutive columns are exactly one quad of that operand -- split3 and three 8-byte stores, the fp32
// tile never reaches HBM (or LDS).
__device__ __forceinline__ void epilogue_store_packed_t(const f32x16 (&acc)[2][2], char *packed, int nk_out, int M,
                                                        int N, int m0, int n0, const float *__restrict__ bias,
                                                        int act, int fmt = PK_BF16X3, const char *resid_pk = nullptr) {
    // resid_pk: a residual [M, N] in the SAME packed format and geometry as the output (a residual stream that only ever
    // exists as a GEMM operand: the ResNet tower), added after the activation; CAPDEC_ACT_RESID_RELU then applies
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l32 = lane & 31;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = m0 + wm * 64 + i * 32 + l32;
        if (row >= M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = n0 + wn * 64 + j * 32 + 8 * g + 4 * half;
                if (col >= N) continue;
                float4 v = acc_quad(acc[i][j], g);
                if (bias) {
                    const float4 b = *reinterpret_cast<const float4 *>(bias + col);
                    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                }
                v.x = act_apply(v.x, act); v.y = act_apply(v.y, act);
                v.z = act_apply(v.z, act); v.w = act_apply(v.w, act);
                if (resid_pk) {
                    const float4 rr = x3_load_quad(resid_pk, nk_out, row, col >> 4, (col >> 2) & 3, fmt);
                    v.x = post_resid(v.x + rr.x, act); v.y = post_resid(v.y + rr.y, act);
                    v.z = post_resid(v.z + rr.z, act); v.w = post_resid(v.w + rr.w, act);
                }
                x3_store_quad(packed, nk_out, row, col >> 4, (col >> 2) & 3, v, fmt);
            }
    }
}

// top-KSEL (value, column) of one row's 128 logits held by a 16-lane group, 8 per lane (lane `sub` holds columns
// sub + 16 j): descending value, equal values in ascending column order -- the order topk_merge and the reference's
// torch.topk walk expect.  Round 4 form: per round the VALUE of the best remaining logit comes from a max3 tree and
// four v_max_f32 DPP steps, its lowest COLUMN from an equality scan and four v_min_u32 DPP steps, then the winner is
// retired -- ~45 VALU instructions where the former (value, column) pair tournament with its two-key comparisons took
// ~80; the selection rounds were a fifth of the fused lm_head's time (CAPDEC_LMHEAD_K1 measurement).
__device__ __forceinline__ int row16_min_i(int v) {
    v = min(v, dpp_i<DPP_XOR1>(v));
    v = min(v, dpp_i<DPP_XOR2>(v));
    v = min(v, dpp_i<DPP_HALF_MIRROR>(v));
    v = min(v, dpp_i<DPP_MIRROR>(v));
    return v;
}
template <int KSEL>
__device__ __forceinline__ void row_tile_topk(float (&v)[8], float row_max, int sub, int n0, bool write, size_t tbase,
                                              float *__restrict__ cand_val, int *__restrict__ cand_idx) {
#pragma unroll
    for (int kk = 0; kk < KSEL; ++kk) {
        float gv = row_max;                            // (round 0: the row maximum the caller already has)
        if (kk > 0) {
            const float lm = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
            gv = row16_max(lm);
        }
        int lc = 1 << 20;                              // lowest local column holding gv (columns ascend with j: scan downwards)
#pragma unroll
        for (int j = 7; j >= 0; --j) lc = v[j] == gv ? sub + 16 * j : lc;
        const int gc = row16_min_i(lc);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (sub + 16 * j == gc) ? -INFINITY : v[j];     // retire the winner (one lane, one j)
        if (write && sub == kk) {
            cand_val[tbase * KSEL + kk] = gv;
            cand_idx[tbase * KSEL + kk] = n0 + gc;
        }
    }
}

// lm_head epilogue: per (row, 128-col tile) max, sum exp(x - max) and top-k (value, column).  The
// logits tile goes accumulators -> LDS (`Ct`, >= 64 x CT_LD floats -- 128 x CT_LD with FULL --, reusing the staging buffers;
// the caller's main loop must have ended with a barrier) -> 16-lane groups, one row per group, 8
// columns per lane; only (2 + 2k) words per (row, tile) reach HBM.
template <int KSEL, int WMG = 2, bool FULL = false>
__device__ __forceinline__ void epilogue_topk(const f32x16 (&acc)[2][WaveGrid<WMG>::NJ], float *Ct, int M, int N,
                                              int m0, int n0, int tn, int tiles_n, float inv_temp, float *tile_max,
                                              float *tile_sum, float *cand_val, int *cand_idx) {
    constexpr int NJ = WaveGrid<WMG>::NJ;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = WaveGrid<WMG>::wm(wave), wn = WaveGrid<WMG>::wn(wave), half = lane >> 5, l32 = lane & 31;
    const int grp = lane >> 4, sub = lane & 15;
    // FULL (WMG == 2, Ct >= 128 x CT_LD floats = 66 KB): the whole 128-row tile goes through LDS at once -- every
    // wavefront stores its accumulators, one barrier, 32 rows per wavefront; otherwise 64 rows at a time.
    constexpr int SLABS = FULL ? 1 : WMG, ITERS = FULL ? 8 : 4, ROWS_PER_WAVE = FULL ? 32 : 16;
    for (int hh = 0; hh < SLABS; ++hh) {
        if (FULL || wm == hh) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (FULL ? wm * 64 : 0) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        Ct[row * CT_LD + wn * 32 * NJ + j * 32 + l32] = acc[i][j][r] * inv_temp;
                    }
        }
        __syncthreads();
        for (int it = 0; it < ITERS; ++it) {
            const int rl = wave * ROWS_PER_WAVE + it * 4 + grp;   // row within the slab
            const int row = m0 + hh * 64 + rl;
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
        if (!FULL) __syncthreads();
    }
}

}  // namespace capdec
