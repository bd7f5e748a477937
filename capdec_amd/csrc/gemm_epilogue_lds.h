// Coalesced GEMM epilogues: the accumulators of a wavefront go through an LDS transposition before they leave the CU.
//
// Why (round 4, tools/pp_stamps.py): in the TR accumulator layout a lane owns four consecutive columns of ONE row, so a
// float4 store instruction of the direct epilogues (gemm_epilogue.h, gemm_epilogue_w.h) touches 64 different rows -- 64
// partly written 128-byte lines per instruction, ~600 cycles each: a 256 x 256 tile took 17-20 us to store (a quarter of
// its whole time at 25 000 rows; most of a 625-caption launch's fixed cost), the packed outputs (8-byte pieces) more.
// Here every wavefront writes a 32-row block of its tile into its own LDS slab [32][W + 4] fp32 (16 ds_write_b128 at
// most), reads it back row-major -- consecutive lanes hold consecutive 16 bytes of a row -- applies bias / activation /
// residual there (bias is per column: loaded once per lane; the residual is read as coalesced as the result is written)
// and stores whole row segments: W = 64 columns -> 4 rows x 256 B per instruction.  The packed f16x2 output leaves as
// 16-byte pieces that tile 256-byte runs of a plane, the K / V third of the decode-step qkv projection as whole 256-byte
// keys / values.  No block barrier: a slab is private to its wavefront (the ring must be idle: the main loops end with
// vmcnt(0) + s_barrier).
#pragma once
#include "gemm_epilogue.h"

namespace capdec {

struct EpiArgs {
    float *C = nullptr;
    int ldc = 0, M = 0, N = 0, m0 = 0, n0 = 0;
    const float *bias = nullptr;
    const float *resid = nullptr;      // fp32 residual [M, ldr] (C path)
    int ldr = 0, act = CAPDEC_ACT_NONE;
    float scale = 1.0f;
    char *packed = nullptr;            // packed f16x2 / x1 output (K of the next GEMM = N)
    const char *resid_pk = nullptr;    // packed residual of the output's format
    int fmt = PK_F16X2;
    const QkvScatter *sc = nullptr;    // decode-step qkv projection: K / V thirds straight into the cache
    bool nt = false;                   // fp32 C: non-temporal stores (the result is not re-read by this kernel)
};

template <int TJ> struct EpiSlab {
    static constexpr int W = TJ * 32, LD = W + 4;                 // floats; LD % 32 == 4: conflict-free 16-byte row writes
    static constexpr int BYTES = 32 * LD * 4;
};

template <int ACT> __device__ __forceinline__ float act_const(float v) {
    if constexpr (ACT == CAPDEC_ACT_TANH) return tanhf(v);
    else if constexpr (ACT == CAPDEC_ACT_RELU) return fmaxf(v, 0.f);
    else if constexpr (ACT == CAPDEC_ACT_GELU_NEW) {
        const float c2 = 2.0f * 0.7978845608028654f;
        const float u2 = v * (c2 + (c2 * 0.044715f) * v * v);
        return __fdividef(v, 1.f + __expf(-u2));
    } else if constexpr (ACT == CAPDEC_ACT_QUICK_GELU) return __fdividef(v, 1.f + __expf(-1.702f * v));   // (rcp + mul like gelu_new above: an IEEE division is ~10 VALU instructions per element of a K = 512 tile whose MFMAs take less time than its epilogue)
    else return v;
}
template <int ACT> __device__ __forceinline__ float post_resid_const(float v) {
    if constexpr (ACT == CAPDEC_ACT_RESID_RELU) return fmaxf(v, 0.f);
    else return v;
}

// rows of the 32-row block per pass and passes per block when a lane owns Q consecutive quads (4 Q columns) of a row
template <int TJ, int Q> struct EpiWalk {
    static constexpr int LPR = TJ * 8 / Q;                        // lanes per row
    static constexpr int PASSES = (32 * LPR + 63) / 64;
};

// One wavefront: acc[i][j] (TR layout; WAVE_M / WAVE_N = the wavefront's block coordinates in units of its tile) ->
// slab -> out.  MODE 0: fp32 C (+ bias, activation, residual); 1: packed output (+ packed residual); 2: qkv scatter.
template <int TI, int TJ, int MODE, int ACT>
__device__ __forceinline__ void epilogue_lds_wave(const f32x16 (&acc)[TI][TJ], float *slab, int row0, int col0,
                                                  const EpiArgs &a) {
    using S = EpiSlab<TJ>;
    const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
    constexpr int Q = MODE == 1 ? 2 : 1;                           // packed output: 8 columns (one 16-byte piece) per lane
    using Wk = EpiWalk<TJ, Q>;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        // ---- accumulators -> slab (row l32, four consecutive columns per quad)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4 *>(slab + l32 * S::LD + j * 32 + 8 * g + 4 * half) = acc_quad(acc[i][j], g);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- slab -> memory, row-major
#pragma unroll
        for (int p = 0; p < Wk::PASSES; ++p) {
            const int idx = p * 64 + lane;
            const int rl = idx / Wk::LPR, cq = idx - rl * Wk::LPR;  // row in the block, piece in the row
            const int row = row0 + i * 32 + rl, col = col0 + cq * (4 * Q);
            const bool live = (Wk::PASSES * 64 == 32 * Wk::LPR || rl < 32) && row < a.M && col < a.N;
            if (!live) continue;
            float4 v[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                v[q] = *reinterpret_cast<const float4 *>(slab + rl * S::LD + cq * (4 * Q) + 4 * q);
                v[q].x *= a.scale; v[q].y *= a.scale; v[q].z *= a.scale; v[q].w *= a.scale;
                if (a.bias) {
                    const float4 b = *reinterpret_cast<const float4 *>(a.bias + col + 4 * q);
                    v[q].x += b.x; v[q].y += b.y; v[q].z += b.z; v[q].w += b.w;
                }
                v[q].x = act_const<ACT>(v[q].x); v[q].y = act_const<ACT>(v[q].y);
                v[q].z = act_const<ACT>(v[q].z); v[q].w = act_const<ACT>(v[q].w);
            }
            if constexpr (MODE == 0) {
                if (a.resid) {
                    const float4 r4 = *reinterpret_cast<const float4 *>(a.resid + (size_t)row * a.ldr + col);
                    v[0].x = post_resid_const<ACT>(v[0].x + r4.x); v[0].y = post_resid_const<ACT>(v[0].y + r4.y);
                    v[0].z = post_resid_const<ACT>(v[0].z + r4.z); v[0].w = post_resid_const<ACT>(v[0].w + r4.w);
                }
                if (a.nt) {
                    typedef float nt_f4 __attribute__((ext_vector_type(4)));
                    nt_f4 w; w[0] = v[0].x; w[1] = v[0].y; w[2] = v[0].z; w[3] = v[0].w;
                    __builtin_nontemporal_store(w, reinterpret_cast<nt_f4 *>(a.C + (size_t)row * a.ldc + col));
                }
                else *reinterpret_cast<float4 *>(a.C + (size_t)row * a.ldc + col) = v[0];
            } else if constexpr (MODE == 2) {
                const QkvScatter &sc = *a.sc;
                if (col < sc.d) {
                    *reinterpret_cast<float4 *>(a.C + (size_t)row * a.ldc + col) = v[0];
                } else {
                    const int cap = row / sc.beam, b = row - cap * sc.beam;
                    const size_t srow = (size_t)(sc.cmap ? sc.cmap[cap] : cap) * sc.beam + b;
                    float *cache = col >= 2 * sc.d ? sc.vc : sc.kc;
                    const int hc = col - (col >= 2 * sc.d ? 2 * sc.d : sc.d), head = hc >> 6;
                    const size_t el = ((srow * sc.heads + head) * sc.ctx + sc.pos) * 64 + (hc & 63);
                    if (sc.bf16) {
                        bf16x4 t;
                        t[0] = (__bf16)v[0].x; t[1] = (__bf16)v[0].y; t[2] = (__bf16)v[0].z; t[3] = (__bf16)v[0].w;
                        *reinterpret_cast<bf16x4 *>(reinterpret_cast<__bf16 *>(cache) + el) = t;
                    } else
                        *reinterpret_cast<float4 *>(cache + el) = v[0];
                }
            } else {
                // one 16-byte half of a (row, k-step) chunk per plane: k-step = col / 16, half = (col / 8) & 1, swizzled
                // by bit 3 of the row like x3_group_offset
                const int nk_out = a.N >> 4, ks = col >> 4, hf = (col >> 3) & 1, r7 = row & 127;
                const size_t blk = (size_t)(row >> 7) * nk_out + ks;
                const int inrow = r7 * X3_ROW_B + ((hf ^ ((r7 >> 3) & 1)) << 4);
                if (a.resid_pk) {
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        const float4 rr = x3_load_quad(a.resid_pk, nk_out, row, ks, 2 * hf + q, a.fmt);
                        v[q].x = post_resid_const<ACT>(v[q].x + rr.x); v[q].y = post_resid_const<ACT>(v[q].y + rr.y);
                        v[q].z = post_resid_const<ACT>(v[q].z + rr.z); v[q].w = post_resid_const<ACT>(v[q].w + rr.w);
                    }
                }
                if (a.fmt == PK_F16X2) {
                    f16x4 h0, l0, h1, l1;
                    split2h(v[0], h0, l0);
                    split2h(v[1], h1, l1);
                    char *pp = a.packed + blk * H2_BLOCK_B + inrow;
                    f16x8 hh, ll;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { hh[e] = h0[e]; hh[4 + e] = h1[e]; ll[e] = l0[e]; ll[4 + e] = l1[e]; }
                    *reinterpret_cast<f16x8 *>(pp) = hh;
                    *reinterpret_cast<f16x8 *>(pp + X3_PLANE_B) = ll;
                } else {      // one-plane formats
                    x3_store_quad(a.packed, nk_out, row, ks, 2 * hf, v[0], a.fmt);
                    x3_store_quad(a.packed, nk_out, row, ks, 2 * hf + 1, v[1], a.fmt);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// the activation code becomes a compile-time constant of the element loops (one switch per tile, not per element)
template <int TI, int TJ, int MODE>
__device__ __forceinline__ void epilogue_lds_dispatch(const f32x16 (&acc)[TI][TJ], float *slab, int row0, int col0,
                                                      const EpiArgs &a) {
    switch (a.act) {
        case CAPDEC_ACT_GELU_NEW: epilogue_lds_wave<TI, TJ, MODE, CAPDEC_ACT_GELU_NEW>(acc, slab, row0, col0, a); break;
        case CAPDEC_ACT_QUICK_GELU: epilogue_lds_wave<TI, TJ, MODE, CAPDEC_ACT_QUICK_GELU>(acc, slab, row0, col0, a); break;
        case CAPDEC_ACT_RELU: epilogue_lds_wave<TI, TJ, MODE, CAPDEC_ACT_RELU>(acc, slab, row0, col0, a); break;
        case CAPDEC_ACT_TANH: epilogue_lds_wave<TI, TJ, MODE, CAPDEC_ACT_TANH>(acc, slab, row0, col0, a); break;
        case CAPDEC_ACT_RESID_RELU: epilogue_lds_wave<TI, TJ, MODE, CAPDEC_ACT_RESID_RELU>(acc, slab, row0, col0, a); break;
        default: epilogue_lds_wave<TI, TJ, MODE, CAPDEC_ACT_NONE>(acc, slab, row0, col0, a); break;
    }
}

// block-level entry: G supplies WN, TI, TJ (wavefront w owns tile (w / WN, w % WN)); smem = the block's idle ring, at
// least NW * EpiSlab<TJ>::BYTES
template <class G>
__device__ __forceinline__ void epilogue_lds(const f32x16 (&acc)[G::TI][G::TJ], char *smem, const EpiArgs &a) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / G::WN, wn = wave % G::WN;
    float *slab = reinterpret_cast<float *>(smem + wave * EpiSlab<G::TJ>::BYTES);
    const int row0 = a.m0 + wm * G::TI * 32, col0 = a.n0 + wn * G::TJ * 32;
    if (a.packed) epilogue_lds_dispatch<G::TI, G::TJ, 1>(acc, slab, row0, col0, a);
    else if (a.sc && a.sc->kc) epilogue_lds_wave<G::TI, G::TJ, 2, CAPDEC_ACT_NONE>(acc, slab, row0, col0, a);
    else epilogue_lds_dispatch<G::TI, G::TJ, 0>(acc, slab, row0, col0, a);
}

}  // namespace capdec
