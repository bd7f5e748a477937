// Internal declarations shared by the translation units behind the C ABI (include/capdec.h): the context, the weight
// containers, the per-family profiler and the host-side pieces each unit exports to the others.
//   capi_context.hip  context lifetime, streams, modes, memory, timers, profiler, decode statistics, the environment knobs
//   weights.hip       capdec_load_* (uploads; Conv1D transposes; BatchNorm folding)
//   gemm_dispatch.hip the GEMM planner: operand planes cache, which kernel / split for a projection, capdec_gemm_f32
//   decode.hip        the pre-LN block stack, fused lm_head + selection, the KV-cached greedy / beam decode loop, mapper
//   train_*.hip       the train step (train.h): step, mapping networks, shared backward pieces, optimizer + C entry points
//   clip.hip          CLIP ViT-B/32 towers, the ModifiedResNet tower, image preprocessing
//   comm.hip          caption-shard bounds and the RCCL all-gather (librccl dlopen'ed)
#pragma once
#include <rccl/rccl.h>      // types only: the library is dlopen'ed (no link-time dependency on RCCL)

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <utility>

#include "bf16x3.h"
#include "common.h"
#include "config.h"

namespace capdec {

// ---------------------------------------------------------------------------- device buffers
struct DBuf {   // grow-only device buffer
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) CAPDEC_HIP(hipFree(p));
        p = nullptr;
        cap = 0;
        CAPDEC_HIP(hipMalloc(&p, bytes));
        cap = bytes;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct Gpt2Layer {
    float *ln1w, *ln1b, *wqkv, *bqkv, *wproj, *bproj, *ln2w, *ln2b, *wfc, *bfc, *wproj2, *bproj2;
};
struct Gpt2 {
    bool loaded = false;
    int n_layer = 0, n_head = 0, d = 0, vocab = 0, n_pos = 0;
    float eps = 1e-5f;
    float *wte = nullptr, *wpe = nullptr, *lnfw = nullptr, *lnfb = nullptr;
    std::vector<Gpt2Layer> layers;
    std::vector<void *> owned;
};
struct TMapLayer {
    float *n1w, *n1b, *wqkv, *wproj, *bproj, *n2w, *n2b, *wfc1, *bfc1, *wfc2, *bfc2;
};
struct Mapper {
    int kind = 0;   // 0 none, 1 mlp, 2 transformer
    int D = 0, P = 0, d = 768;
    // mlp
    int hidden = 0;
    float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr;
    // transformer
    int clip_len = 0, n_layers = 0, heads = 8, mlp_hidden = 0;
    float *lin_w = nullptr, *lin_b = nullptr, *prefix_const = nullptr;
    std::vector<TMapLayer> layers;
    std::vector<void *> owned;
};

// CLIP tower = the same pre-LN block stack as GPT-2 (fused qkv, 4w MLP) with QuickGELU
struct Tower {
    bool loaded = false;
    int n_layer = 0, n_head = 0, d = 0, embed = 0;
    std::vector<Gpt2Layer> layers;
    std::vector<void *> owned;
    // text
    int ctx = 0, vocab = 0;
    float *tok_emb = nullptr, *pos_emb = nullptr, *lnf_w = nullptr, *lnf_b = nullptr, *proj_t = nullptr;  // proj_t [embed, d]
    // vision
    int image = 0, patch = 0, ntok = 0;
    float *conv_w = nullptr, *cls = nullptr, *ln_pre_w = nullptr, *ln_pre_b = nullptr;
};

// CLIP ModifiedResNet: a convolution with its BatchNorm folded in, as a GEMM operand
struct ConvW {
    float *w = nullptr;      // [cout_p, K]: K = k*k*cin_p in (ky, kx, c) order (first stem conv: 27 real columns, padded to 64)
    float *b = nullptr;      // [cout_p]
    int cin = 0, cout = 0, k = 0, cin_p = 0, cout_p = 0, K = 0;
};
struct ResNet {
    bool loaded = false;
    int image = 0, width = 0, embed = 0, feat = 0, heads = 0, sp = 0;
    int layers[4] = {0, 0, 0, 0};
    ConvW stem[3];
    std::vector<ConvW> blocks;          // 4 per bottleneck (conv1, conv2, conv3, downsample; downsample.w may be null)
    float *pos = nullptr, *wq = nullptr, *bq = nullptr, *wk = nullptr, *bk = nullptr, *wv = nullptr, *bv = nullptr,
          *wc = nullptr, *bc = nullptr;
    std::vector<void *> owned;
};

enum Family { F_GEMM = 0, F_LMHEAD, F_ATTN_DEC, F_ATTN_PRE, F_LN, F_EMBED, F_SELECT, F_MAP_ATTN, F_OTHER, F_GEMM_X3,
              F_LMHEAD_X3, F_GEMM_X3P, F_GEMM_BF16P, F_LMHEAD_BF16, F_GEMM_H2P, F_LMHEAD_H2, F_PACK, F_LMHEAD_2ND, F_COUNT };
extern const char *const kFamilyNames[F_COUNT];
constexpr int PROF_SLOTS = 24;   // capdec_profile_get fills at most this many families (engine.py sizes its arrays by it)
static_assert(F_COUNT <= PROF_SLOTS, "profile arrays too small");
enum GemmMode { GEMM_F32 = 0, GEMM_BF16X3 = 1, GEMM_BF16 = 2, GEMM_F16X2 = 3, GEMM_F16 = 4 };

struct Prof {
    bool on = false;
    int every = 1;                       // time every `every`-th launch of each family (1 = all)
    int64_t calls[PROF_SLOTS] = {0};     // launches seen per family (timed or not)
    struct Rec { int fam; hipEvent_t a, b; double flops; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    double ms[F_COUNT] = {0}, flops[F_COUNT] = {0};
    int64_t launches[F_COUNT] = {0};
};

}  // namespace capdec

using namespace capdec;

namespace capdec { struct TrainState; }      // train.h: saved activations, gradients, AdamW moments of the train step
struct capdec_ctx {
    capdec::Tuning tune;            // the environment knobs, parsed once by capdec_create (config.h)
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    size_t kv_budget = (size_t)192 << 30;
    Gpt2 gpt;
    Mapper map;
    Tower clip_text, clip_vision;
    ResNet clip_resnet;
    DBuf r_a, r_b, r_c, r_d, r_e, r_f, r_col;      // ResNet activation buffers (NHWC) + im2col
    DBuf r_pk1, r_pk2, r_xpk, r_ypk, r_xi, r_idp, r_zero;   // ... packed activations (GEMM / implicit-conv operands), zero rows
    Prof prof;
    int gemm_mode = GEMM_F16X2;
    struct Planes { void *p; size_t n; int fmt; bool wide_ok; };   // wide_ok: max |w| < 16 (see GemmEpilogue::wide_ok)
    std::map<std::pair<const void *, int>, Planes> planes;   // (fp32 weight, PackFmt) -> packed planes: a weight used in two
                                                             // formats (train forward f16x2, decode in a one-plane mode) keeps both
    DBuf x3_tmp, xpk, apk, fpk, a_tmp;   // scratch planes for un-cached matrices; packed LayerNorm output; packed fp32-A
    int stat_steps = 0, stat_compactions = 0;      // last decode call: steps run, compactions done,
    long long stat_row_steps = 0;                  // activation rows pushed through the GPT-2 body (prefill excluded)
    std::vector<int> stat_step_rows;               // ... per decode step (capdec_decode_step_rows)
    bool batch_invariant = false;   // capdec_set_batch_invariant: no launch-size dependent summation order (no split-K, pinned kernel variants)
    int diverge = 0;                // measurement: beams never share history (capdec_set_debug_diverge)
    double stat_kv_slots = 0.0, stat_kv_pos = 0.0;   // last beam decode: sums behind capdec_decode_counters (filled lazily)
    bool k3_off = false;                             // this decode call went back to k candidates per tile (poll_alive)
    long long k3_rows = 0;                           // rows of this call's lm_head launches that kept 3 per tile
    bool lmflag_live = false;                        // `lmflag` holds the second-pass row count of the last decode call
    int kvstat_n = 0;                                // captions of the last beam decode whose counters sit in `kvstat`
    bool compact = true;       // decode: drop finished captions from the batch at the poll points (CAPDEC_COMPACT=0: off)
    bool pack_chain = true;    // ... and attention / the fc GEMM epilogue emit the packed A operand of the GEMM that follows
    bool pack_a = true;        // bf16x3 mode: LayerNorm emits the packed A operand, GEMM moves both operands by LDS-DMA
    hipEvent_t t0 = nullptr, t1 = nullptr;
    // workspaces
    DBuf h, x, qkv, att, ff, xl, tmax, tsum, cval, cidx, lse, topv, topi, kc, vc;
    DBuf tokens, scores, seq, stopped, done, anc, next_tok, alive, gids, glens, cmap, kvstat;
    capdec::TrainState *train = nullptr;     // created by the first capdec_train_step, freed by train_release
    int train_scope = 0;                     // capdec_train_set_scope: survives capdec_train_reset and weight reloads
    float train_drop_p = 0.f;                // capdec_train_set_dropout: GPT-2's dropout probability in scope 1 (0 = off)
    unsigned long long train_drop_seed = 0;  // ... key of the Philox keep-mask stream (counter = element, train step)
    DBuf lmflag, xpk2;     // fused lm_head with 3 candidates per tile: [count, total, rows...] of the rows whose top 5 need
                           // the exact second pass; their compacted packed A operand (decode.hip: lm_head_select)
    DBuf m_hid, m_lin, m_seq, m_x, m_qkv, m_att, m_ff;
    DBuf t_idx, t_patch, t_pout, p_desc, p_inter, splitk, absmax;
    int *alive_host = nullptr;   // pinned: [captions still generating, second-pass rows so far]
    // caption-shard communicator (RCCL), see capdec_comm_init
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    DBuf g_pad, g_all;
};


namespace capdec {

// ---- capi_context.hip
// ---------------------------------------------------------------------------- profiling
inline hipEvent_t prof_event(capdec_ctx *c) {
    if (!c->prof.pool.empty()) {
        hipEvent_t e = c->prof.pool.back();
        c->prof.pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
struct ProfScope {
    capdec_ctx *c;
    int idx = -1;
    ProfScope(capdec_ctx *ctx, int fam, double flops = 0.0) : c(ctx) {
        if (!c->prof.on) return;
        if (c->prof.calls[fam]++ % c->prof.every != 0) return;
        Prof::Rec r{fam, prof_event(c), prof_event(c), flops};
        (void)hipEventRecord(r.a, c->stream);
        c->prof.recs.push_back(r);
        idx = (int)c->prof.recs.size() - 1;
    }
    ~ProfScope() {
        if (idx >= 0) (void)hipEventRecord(c->prof.recs[idx].b, c->stream);
    }
};
int prof_collect(capdec_ctx *c);

// ---- train_optim.hip (the train step: train.h)
void train_release(capdec_ctx *c);      // frees the train-step state (a mapper / GPT-2 reload invalidates it)

// ---- comm.hip
void comm_release(capdec_ctx *c);       // destroys the context's communicator, if any

inline int pad64(int c) { return (c + 63) / 64 * 64; }

// ---- weights.hip
int upload(std::vector<void *> &owned, const float *h_src, size_t n, float **out);
int upload_transposed(capdec_ctx *c, std::vector<void *> &owned, const float *h_src, int rows, int cols, float **out);
void free_all(std::vector<void *> &owned);

// ---- gemm_dispatch.hip: the planner
void drop_planes(capdec_ctx *c);
void drop_planes_of(capdec_ctx *c, const void *weight);      // one cached weight (its values changed: train_optim.hip)
int pack_fmt(const capdec_ctx *c);
inline bool mode_single(const capdec_ctx *c) { return c->gemm_mode == GEMM_BF16 || c->gemm_mode == GEMM_F16; }
int pack_any(capdec_ctx *c, const float *W, int N, int K, int fmt, void *out);
int planes_of(capdec_ctx *c, const float *W, int N, int K, bool cache, const void **out, int fmt_override = -1,
              bool *wide_ok = nullptr);
// C = act(A . Bt^T + bias) + resid with fp32 A in HBM (mapper, patch embedding, projections)
int gemm(capdec_ctx *c, const float *A, int lda, const float *Bt, int ldb, float *C, int ldc, int M, int N, int K,
         const float *bias, int act, const float *resid = nullptr, int ldr = 0, bool weight = true);
// C [N1, N2] = Xa^T Xb (fp32 Xa [rows, N1] / Xb [rows, N2], row strides lda / ldb): weight-gradient products, operands packed transposed
int gemm_tn(capdec_ctx *c, const float *Xa, int lda, const float *Xb, int ldb, int rows, int N1, int N2, float *C, int ldc);
bool use_packed_a(capdec_ctx *c, int K);
struct NextLn { const float *w, *b; float eps; int *done; };
int gemm_packed(capdec_ctx *c, const void *Apk, const float *W, float *C, int ldc, int M, int N, int K, const float *bias,
                int act, const float *resid = nullptr, int ldr = 0, void *packed_out = nullptr,
                const NextLn *next_ln = nullptr, const void *resid_packed = nullptr,
                const QkvScatter *qkv_scatter = nullptr);
int ln_gemm_packed(capdec_ctx *c, const float *h, int ldh, const float *lnw, const float *lnb, float eps, const float *W,
                   float *C, int ldc, int M, int N, int K, const float *bias, int act, void *packed_out = nullptr,
                   bool ln_ready = false, const QkvScatter *qkv_scatter = nullptr);

// ---- decode.hip: the block stack (GPT-2 and the CLIP towers run on it)
struct StepShape {
    bool prefill;
    int ncap, P, beam;      // prefill: rows = ncap * P
    int rows, L;            // decode: rows at context length L
    const uint8_t *anc;
    int anc_stride;
    const int *cmap;        // decode after compaction: activation row r -> caption cmap[r / beam] (nullptr: identity)
};
struct StackCfg {
    const std::vector<Gpt2Layer> *layers;
    int n_layer, d;
    float eps;
    int act;
    bool causal, keep_kv;
};
int ensure_body_ws(capdec_ctx *c, int M, int d);
int stack_body(capdec_ctx *c, const StackCfg &g, const StepShape &s, const KvCache &kv);
void kv_geometry(KvCache &kv, int rows, int ctx, int heads, int hd);

}  // namespace capdec
