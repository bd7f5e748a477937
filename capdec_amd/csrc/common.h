// Shared declarations for libcapdec_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/capdec.h"
#include "config.h"

namespace capdec {

void set_error(const std::string &msg);

#define CAPDEC_HIP(expr)                                                                         \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            ::capdec::set_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" +       \
                                __FILE__ + ":" + std::to_string(__LINE__) + ")");               \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

#define CAPDEC_CHECK(cond, msg)                                                                  \
    do {                                                                                         \
        if (!(cond)) {                                                                           \
            ::capdec::set_error(std::string(msg) + " [" #cond "]");                              \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

#define CAPDEC_TRY(expr)                                                                         \
    do {                                                                                         \
        int _r = (expr);                                                                         \
        if (_r) return _r;                                                                       \
    } while (0)

constexpr int WAVE = 64;

// ---- wave-level helpers (64-wide wavefronts) ------------------------------------------
// Cross-lane traffic goes through DPP (data-parallel primitives: a VALU operand modifier), not
// through __shfl (which hipcc lowers to ds_bpermute_b32, an LDS-crossbar round trip per step).
// Within a row of 16 lanes: quad_perm [1,0,3,2] / [2,3,0,1] are the xor-1 / xor-2 butterflies, then
// row_half_mirror (lane i <- 7-i) and row_mirror (lane i <- 15-i) join the two quads and the two
// halves: after the four steps every lane of the row holds the reduction of the row's 16 lanes.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;

__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<DPP_XOR1>(v);
    v += dpp_f<DPP_XOR2>(v);
    v += dpp_f<DPP_HALF_MIRROR>(v);
    v += dpp_f<DPP_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<DPP_XOR1>(v));
    v = fmaxf(v, dpp_f<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f<DPP_MIRROR>(v));
    return v;
}
__device__ __forceinline__ float lane_bcast(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// full-wave reductions (all 64 lanes must be active): four row reductions + four scalar reads
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
}

// Philox4x32-10 counter RNG (Salmon et al.): key = seed, counter = (element index, stream).
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                           uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// ---- launch geometry of the f32 MFMA GEMM ----------------------------------------------
constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 32;
constexpr int TOPK_MAX = 8;

// Decode-step qkv projection (f16x2 mode, fp32 KV cache): the K and V thirds of the result go STRAIGHT into the KV cache
// at the position being decoded -- activation row r (caption cmap[r / beam] after compaction) -> cache row
// caption * beam + r % beam, [head][pos][64] -- instead of into the qkv activation buffer, so the attention kernel
// neither re-writes them (154 MB per layer-step at 25 000 rows) nor treats the current token apart: it is simply the
// last cached position.  The q third still goes to C.
struct QkvScatter {
    float *kc = nullptr, *vc = nullptr;     // this layer's K / V cache (fp32; bf16 elements when `bf16` is set)
    const int *cmap = nullptr;
    int beam = 1, heads = 0, ctx = 0, pos = 0, d = 0;
    bool bf16 = false;                      // bf16 mode: the cache holds bf16 -- K / V are rounded (RNE) as they are written
};

// gemm_f32.hip
struct GemmEpilogue {
    const float *bias = nullptr;   // [N]
    const float *resid = nullptr;  // [M, ldr] added after the activation
    int ldr = 0;
    int act = CAPDEC_ACT_NONE;
    void *splitk_ws = nullptr;     // bf16x3p only: workspace for split-K partial tiles (gemm_splitk_ws_bytes); without it
    size_t splitk_ws_bytes = 0;    // under-filled grids run unsplit
    void *packed_out = nullptr;    // bf16x3p only: write act(acc + bias) as the packed split-bf16 A operand (K = N)
                                   // of the next GEMM instead of fp32 C
    const Tuning *tune = nullptr;  // the context's environment knobs (config.h); nullptr = every default
    const QkvScatter *qkv_scatter = nullptr;   // f16x2p, unsplit grids only (see QkvScatter)
    bool invariant = false;        // batch-invariant mode: the unsplit 128 x 128 kernel whatever M is (no planner, no split-K)
    bool wide_ok = false;          // f16x2p only: the B operand is a weight with max |w| < 16, so its high plane can be scaled
                                   // by 2^11 in fp16 registers (the single-accumulator kernels of gemm_h2w.hip)
    const void *resid_packed = nullptr;   // f16x2p / x1 with packed_out only: residual [M, N] stored as a packed operand of
                                          // the output's format (added after the activation; no split-K then)
    // Optional LayerNorm of the RESULT rows, fused into the split-K reduce pass (only when the launch splits K, C has
    // N = ldc columns and N <= 1024): ln_out receives LayerNorm(C row) as a packed operand (format of the launch);
    // *ln_done is set to 1 when the fusion happened, left untouched otherwise (the caller then runs its own LayerNorm)
    const float *ln_w = nullptr, *ln_b = nullptr;
    float ln_eps = 1e-5f;
    void *ln_out = nullptr;
    int *ln_done = nullptr;
};
inline const Tuning &tuning_of(const GemmEpilogue &e) { return e.tune ? *e.tune : default_tuning(); }
int launch_gemm_f32(hipStream_t st, const float *A, int lda, const float *Bt, int ldb, float *C, int ldc,
                    int M, int N, int K, const GemmEpilogue &epi);
// lm_head: logits tile never leaves the CU; per (row, 128-column tile) emits max, sum exp(x - max)
// and the top-k (value, column) pairs.
int launch_gemm_f32_topk(hipStream_t st, const float *A, int lda, const float *Bt, int ldb, int M, int N, int K,
                         int k, float inv_temp, float *tile_max, float *tile_sum, float *cand_val, int *cand_idx,
                         const Tuning *tune = nullptr);
inline int gemm_tiles_n(int N) { return (N + GEMM_BN - 1) / GEMM_BN; }

// gemm_bf16x3.hip: the same two GEMMs on the bf16 matrix cores with operands split into three bf16
// planes (fp32-accurate, 6 MFMA terms).  Bpacked = tile-major planes made by launch_pack_planes.
size_t packed_planes_bytes(int N, int K);
inline size_t x3_packed_bytes_host(int rows, int K) { return (size_t)((rows + 127) / 128) * 128 * K * 3 * sizeof(uint16_t); }
int launch_pack_planes(hipStream_t st, const float *w, int N, int K, void *out);
int launch_gemm_bf16x3(hipStream_t st, const float *A, int lda, const void *Bpacked, float *C, int ldc, int M, int N,
                       int K, const GemmEpilogue &epi);
int launch_gemm_bf16x3_topk(hipStream_t st, const float *A, int lda, const void *Bpacked, int M, int N, int K, int k,
                            float inv_temp, float *tile_max, float *tile_sum, float *cand_val, int *cand_idx);
// packed-A variants: A already split / tile-major (written by launch_layernorm_packed, launch_pack_planes): both
// operands move by LDS-DMA
int launch_gemm_bf16x3p(hipStream_t st, const void *Apacked, const void *Bpacked, float *C, int ldc, int M, int N, int K,
                        const GemmEpilogue &epi);
int launch_gemm_bf16x3p_topk(hipStream_t st, const void *Apacked, const void *Bpacked, int M, int N, int K, int k,
                             float inv_temp, float *tile_max, float *tile_sum, float *cand_val, int *cand_idx);
// split-K of the packed-A kernel for under-filled grids: number of K slices (1 = none) and the workspace it needs
int gemm_splitk_slices(int M, int N, int K, const Tuning &t = default_tuning());
size_t gemm_splitk_ws_bytes(int M, int N, int K, const Tuning &t = default_tuning());
// adds the S partial tiles of a split-K launch in slice order and applies the epilogue (fmt: packed output format)
int launch_splitk_reduce(hipStream_t st, const float *part, int S, int M, int N, const GemmEpilogue &epi, float *C,
                         int ldc, int fmt);
// f16x2 mode (gemm_f16x2.hip): operands as TWO fp16 planes (format PK_F16X2 of bf16x3.h), three MFMAs per product,
// fp32-accurate.  launch_pack_planes_h2 packs an fp32 [N, K] matrix (row stride ldw) -- weights once, fp32 activations
// (mapper, patch embedding) per call.
int launch_pack_planes_h2(hipStream_t st, const float *w, int ldw, int N, int K, void *out);
// ... and the TRANSPOSE of src [rows][cols] (row stride ld) as the packed matrix [cols][Kp] (k = the rows of src, zero-padded to Kp)
int launch_pack_planes_h2_t(hipStream_t st, const float *src, int ld, int rows, int cols, int Kp, void *out);
int launch_gemm_f16x2p(hipStream_t st, const void *Apacked, const void *Bpacked, float *C, int ldc, int M, int N, int K,
                       const GemmEpilogue &epi);
int launch_gemm_f16x2p_topk(hipStream_t st, const void *Apacked, const void *Bpacked, int M, int N, int K, int k,
                            float inv_temp, float *tile_max, float *tile_sum, float *cand_val, int *cand_idx);
// round-3 wide-tile kernels with ONE accumulator set (gemm_h2w.hip); scale = 2^(t - 11), t = pack-time pre-scale
// exponent of the weights; `which`: 2 = 256x128 (two blocks per CU), 3 = 256x256 (8 waves), 6 = 256x256 (4 waves), 8 = 128x192
int h2w_plan(int M, int N, int K);
int launch_absmax_bits(hipStream_t st, const float *w, size_t n, unsigned *d_out);
int launch_gemm_h2w(hipStream_t st, int which, const void *Apacked, const void *Bpacked, float *C, int ldc, int M, int N,
                    int K, const GemmEpilogue &epi, float scale);
int launch_gemm_h2w_topk(hipStream_t st, const void *Apacked, const void *Bpacked, int M, int N, int K, int k,
                         float inv_temp, float *tile_max, float *tile_sum, float *cand_val, int *cand_idx,
                         const Tuning *tune = nullptr);
// round-4 ping-pong kernels (gemm_pp.hip): ONE 8-wavefront block per CU, the two wavefronts of a SIMD alternate between a
// load phase and a matrix phase; `which`: 10 = 256x128 (two accumulator sets), 11 = 256x128, 12 = 256x256, 13 = 128x256 (one set)
int launch_gemm_pp(hipStream_t st, int which, const void *Apacked, const void *Bpacked, float *C, int ldc, int M, int N,
                   int K, const GemmEpilogue &epi, float scale);
// planner: 0 = keep the kernels of rounds 2-3, else a ping-pong geometry (10 = 256x128, 14 = 256x192, 12 = 256x256)
int pp_plan(int M, int N, int K, bool wide_ok, bool can_split, int mode = 2);
size_t pp_splitk_ws_bytes(int which, int M, int N, int K);
// exact second pass of the fused lm_head (decode.hip): k = 5 lists for the *m_dev rows of a compacted packed A operand
int launch_gemm_h2w_topk_dev(hipStream_t st, const void *Apacked, const void *Bpacked, const int *m_dev, int N, int K,
                             float inv_temp, float *tile_max, float *tile_sum, float *cand_val, int *cand_idx);
int launch_gemm_x1_topk_dev(hipStream_t st, const void *Apacked, const void *Bpacked, const int *m_dev, int N, int K,
                            float inv_temp, float *tile_max, float *tile_sum, float *cand_val, int *cand_idx, int fmt);
// generic packer for any PackFmt (gemm_f16x2.hip); the one-plane formats use it
int launch_pack_planes_fmt(hipStream_t st, const float *w, int ldw, int N, int K, void *out, int fmt);
// round-2 one-plane kernels on the f16x2 main loop (gemm_f16x2.hip): 128x128 tile, two blocks per CU, 32-deep stages
int launch_gemm_x1(hipStream_t st, const void *Apacked, const void *Bpacked, float *C, int ldc, int M, int N, int K,
                   const GemmEpilogue &epi, int fmt);
int launch_gemm_x1_topk(hipStream_t st, const void *Apacked, const void *Bpacked, int M, int N, int K, int k,
                        float inv_temp, float *tile_max, float *tile_sum, float *cand_val, int *cand_idx, int fmt);
// LayerNorm whose output goes straight into the packed split-bf16 A format of the next GEMM (d % 16 == 0)
int launch_layernorm_packed(hipStream_t st, const float *x, int ldx, const float *w, const float *b, float eps,
                            void *packed, int rows, int d, int fmt = 0);

// elementwise.hip
int launch_layernorm(hipStream_t st, const float *x, int ldx, const float *w, const float *b, float eps, float *y,
                     int ldy, int rows, int d);
int launch_embed_tokens(hipStream_t st, const int *tok, const float *wte, const float *wpe_row, float *h, int rows,
                        int d, const int *cmap = nullptr, int beam = 1);
int launch_embed_prefix(hipStream_t st, const float *prefix, const float *wpe, float *h, int n, int P, int pos0,
                        int d);
int launch_gather_rows(hipStream_t st, const float *table, const int *ids, float *out, int rows, int d);
int launch_normalize_prefix(hipStream_t st, const float *x, int n, int dim, int normalize, const float *offset,
                            float *out);
int launch_noise_inject(hipStream_t st, const float *x, int n, int dim, float variance, const float *offset,
                        int uniform, int dont_norm, uint64_t seed, const float *noise, const float *u, float *out);
int launch_tmapper_concat(hipStream_t st, const float *lin, const float *prefix_const, float *seq, int n,
                          int clip_len, int P, int d);
int launch_tmapper_take(hipStream_t st, const float *seq, float *out, int n, int clip_len, int P, int d);
int launch_transpose(hipStream_t st, const float *in, float *out, int rows, int cols);  // out[c][r] = in[r][c]
// CLIP glue
int launch_clip_text_embed(hipStream_t st, const int *tokens, const float *tok_emb, const float *pos_emb, float *h,
                           int n, int L, int d, int P = 0, const int *perm = nullptr);
int launch_eot_index(hipStream_t st, const int *tokens, int *flat_idx, int n, int L, int flat = 1);   // (flat: n*L +) argmax_t tokens[n,t]
int launch_eot_rows(hipStream_t st, const int *pos, const int *perm, int *rows, int m, int P);
int launch_scatter_rows(hipStream_t st, const float *src, const int *perm, float *dst, int rows, int d);
int launch_im2col_patches(hipStream_t st, const float *pixels, float *patches, int n, int S, int patch);
int launch_vision_assemble(hipStream_t st, const float *patch_out, const float *cls, const float *pos, float *seq,
                           int n, int ntok, int d);

// gemm_f16x2.hip: implicit-GEMM 3x3 convolution (stride 1, padding 1) on the packed-operand main loop: act_pk = the input
// activation [Nimg H W pixels][Cin] as a packed operand of format fmt (f16x2 / f16 / bf16), Bpacked = weights
// [Cout][9 Cin] in (ky, kx, c) order; zeros = a zeroed device buffer of >= (Cin / 16 + 1) * 8 KB (the padding's rows)
int launch_conv3x3_packed(hipStream_t st, const void *act_pk, const void *Bpacked, float *C, int ldc, int Nimg, int H,
                          int W, int Cin, int Cout, const GemmEpilogue &epi, int fmt, const void *zeros, size_t zero_bytes);

// resnet.hip: glue of CLIP's ModifiedResNet tower (NHWC fp32, channels padded to multiples of 64)
int launch_im2col3x3(hipStream_t st, const float *in, float *out, int N, int H, int W, int C, int stride, bool nchw3,
                     int Kp);
// the same straight into the packed GEMM operand of format fmt (PK_*: bf16x3.h): no fp32 im2col matrix in HBM
int launch_im2col3x3_packed(hipStream_t st, const float *in, void *out, int N, int H, int W, int C, int stride, bool nchw3,
                            int Kp, int fmt);
int launch_avgpool2(hipStream_t st, const float *in, float *out, int N, int H, int W, int C);
// AvgPool2d(2) between packed activations [N H W pixels][C] of format fmt; packed rows -> fp32 [M, C]
int launch_avgpool2_packed(hipStream_t st, const void *in, void *out, int N, int H, int W, int C, int fmt);
int launch_unpack_rows(hipStream_t st, const void *in, float *out, int M, int C, int fmt);
int launch_attnpool_tokens(hipStream_t st, const float *feat, const float *pos, float *t, int N, int HW, int C);
int launch_attnpool_attend(hipStream_t st, const float *q, const float *k, const float *v, float *out, int N, int heads,
                           int T, int C);

// attention.hip
struct KvCache {
    void *k = nullptr;   // [layer][phys_row][head][ctx][hd], fp32 -- or bf16 when `bf16` is set (bf16 GEMM mode)
    void *v = nullptr;
    int rows = 0, heads = 0, ctx = 0, hd = 0;
    bool bf16 = false;
    int prefix_len = 0;           // decode: positions [0, prefix_len) of every row live in slot 0 (the CLIP prefix)
    bool fixed_variant = false;   // batch-invariant mode: the launch-size dependent kernel variants are pinned
    const Tuning *tune = nullptr; // the context's environment knobs (measurement builds read the attention overrides)
    size_t layer_stride() const { return (size_t)rows * heads * ctx * hd; }      // elements
    size_t elem_bytes() const { return bf16 ? 2 : 4; }
    template <typename T> T *kp(int layer) const { return reinterpret_cast<T *>(k) + (size_t)layer * layer_stride(); }
    template <typename T> T *vp(int layer) const { return reinterpret_cast<T *>(v) + (size_t)layer * layer_stride(); }
};
// write K/V of `rows` prefill rows (row = caption * P + i) into phys row caption*beam, position i
int launch_kv_scatter_prefill(hipStream_t st, const float *qkv, const KvCache &c, int layer, int ncap, int P,
                              int beam);
// prefill: query row (caption, i) attends cache positions 0..i of phys row caption*beam
int launch_attn_prefill(hipStream_t st, const float *qkv, const KvCache &c, int layer, int ncap, int P, int beam,
                        float *out, bool causal = true, void *packed_out = nullptr, int fmt = 0);
// decode: row r (caption = r / beam) at position L-1: its own k/v come from qkv (and are written to the cache
// at phys row r), positions p < L-1 from phys row caption*beam + anc[r][p] (anc == nullptr -> r itself)
int launch_attn_decode(hipStream_t st, const float *qkv, const KvCache &c, int layer, int rows, int beam, int L,
                       const uint8_t *anc, int anc_stride, float *out, void *packed_out = nullptr,
                       const int *cmap = nullptr, int fmt = 0, bool cur_cached = false);
// (cur_cached: the rows' own K / V are already in the cache at position L-1 -- written by the qkv GEMM's epilogue,
//  QkvScatter -- so the kernel neither reads them from `qkv` nor appends them)
// (cmap != nullptr: finished captions were compacted away -- activation row r belongs to caption cmap[r / beam];
//  KV cache, ancestor table and beam state stay indexed by the original caption)
// (packed_out != nullptr: the attention rows are written as the packed split-bf16 A operand of c_proj, K = d,
//  instead of fp32 `out`)
// TransformerMapper self-attention (no mask): q / k / v rows of n*seq tokens (row strides ldq, ldkv), head-major
int launch_attn_mapper(hipStream_t st, const float *q, int ldq, const float *k, const float *v, int ldkv, float *out,
                       int n, int seq, int heads, int hd);

// select.hip
int launch_topk_merge(hipStream_t st, const float *tile_max, const float *tile_sum, const float *cand_val,
                      const int *cand_idx, int rows, int ntiles, int k, float *lse, float *top_val, int *top_idx);
// Three candidates per (row, tile) in, the row's top 5 + logsumexp out; a row one of whose tiles may hide a fourth
// candidate that still matters (the tile's third kept candidate is strictly better than the row's fifth) is appended to
// flag_rows[atomicAdd(flag_count)] -- its top 5 are provisional until the exact second pass has rewritten them
int launch_topk_merge_k3(hipStream_t st, const float *tile_max, const float *tile_sum, const float *cand_val,
                         const int *cand_idx, int rows, int ntiles, float *lse, float *top_val, int *top_idx,
                         int *flag_rows, int *flag_count, int *flag_total);
// the second pass's merge: *count_dev compact rows of k = 5 lists -> top_val / top_idx of rows out_rows[i]
int launch_topk_merge_rows(hipStream_t st, const float *cand_val, const int *cand_idx, const int *count_dev,
                           const int *out_rows, int rows_cap, int ntiles, float *top_val, int *top_idx);
// rows src_rows[i], i < *count_dev, of a packed operand [*, K] (PK_F16X2 or a one-plane format) -> rows i of `out`
int launch_gather_packed_rows(hipStream_t st, const void *packed, int K, const int *src_rows, const int *count_dev,
                              int rows_cap, void *out, int fmt);
struct BeamState {
    int *tokens = nullptr;      // [ncap, beam, T]
    float *scores = nullptr;    // [ncap, beam]  running SUM of log-probs (reference `scores`)
    float *seq = nullptr;       // [ncap, beam]  reference `seq_lengths` (fp32)
    uint8_t *stopped = nullptr; // [ncap, beam]
    uint8_t *done = nullptr;    // [ncap]  caption's loop has broken (all beams stopped)
    uint8_t *anc = nullptr;     // [ncap, beam, ctx]
    int *next_tok = nullptr;    // [ncap * beam]
    int *alive_count = nullptr; // [1] captions still running (device counter, polled by the host)
    unsigned *kv_stat = nullptr;   // [ncap, 2] optional: per caption, sum over the decode steps of (distinct K/V slots the
                                   // next step's attention reads, positions it attends to) -- what the attention roofline depends on
    int diverge = 0;               // measurement only: every beam continues ITSELF (worst-case K/V traffic; results differ)
};
int launch_beam_init(hipStream_t st, const BeamState &s, const float *lse, const float *top_val, const int *top_idx,
                     int ncap, int beam, int k, int T, int ctx, int P, int stop_id);
int launch_beam_step(hipStream_t st, const BeamState &s, const float *lse, const float *top_val, const int *top_idx,
                     int ncap, int beam, int k, int T, int ctx, int step, int pos_new, int vocab, int stop_id,
                     const int *cmap = nullptr);
int launch_beam_finalize(hipStream_t st, const BeamState &s, int ncap, int beam, int T, int *ids, int *lens,
                         float *scores, int *order);
int launch_greedy_step(hipStream_t st, const int *top_idx, int rows, int step, int T, int stop_id, int alt_stop_id,
                       int *ids, int *lens, uint8_t *done, int *next_tok, int *alive_count,
                       const int *cmap = nullptr, int k = 1, const int *forced = nullptr, const float *top_val = nullptr,
                       const float *lse = nullptr, float *stats = nullptr);
// mean over the rows with label != ignore_index of logsumexp(logits[row]) - logits[row][label]; nll_ws: rows floats
int launch_cross_entropy_mean(hipStream_t st, const float *logits, int ld, const int *labels, int rows, int V,
                              int ignore_index, float *nll_ws, float *out);
// cmap[0..count) = indices of the captions with done[c] == 0, ascending; *count = how many (one block)
int launch_compact_alive(hipStream_t st, const uint8_t *done, int ncap, int *cmap, int *count);

// operands clamped to the fp16 range since the last reset, per translation unit (bf16x3.h: g_h2_saturated)
unsigned long long sat_count_gemm_f16x2(bool reset);
unsigned long long sat_count_gemm_h2w(bool reset);
unsigned long long sat_count_gemm_pp(bool reset);
unsigned long long sat_count_gemm_bf16x3(bool reset);
unsigned long long sat_count_elementwise(bool reset);
unsigned long long sat_count_attention(bool reset);
unsigned long long sat_count_resnet(bool reset);

// preprocess.hip: PIL-exact bicubic resize + centre crop + ToTensor + Normalize of a batch of uint8 RGB images
struct ImageDesc {
    long long off;     // byte offset of the image (HWC uint8) in the concatenated pixel buffer
    long long ioff;    // byte offset of its [H][n_px][3] intermediate (after the horizontal pass)
    int H, W;          // input size
    int rh, rw;        // size after Resize
    int top, left;     // crop origin inside the resized image
};
int launch_preprocess(hipStream_t st, const uint8_t *rgb, const ImageDesc *desc, int n, int max_h, int n_px,
                      uint8_t *inter, float *out, const float *mean, const float *stdv);

}  // namespace capdec
