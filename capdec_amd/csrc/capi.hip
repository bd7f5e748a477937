// C ABI of libcapdec_hip.so (include/capdec.h) and the host-side orchestration of the
// KV-cached batched decode.  One context per GPU; everything is enqueued on one HIP stream.
#include <dlfcn.h>
#include <rccl/rccl.h>      // types only: the library is dlopen'ed (no link-time dependency on RCCL)

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <unordered_map>

#include "bf16x3.h"
#include "common.h"

namespace capdec {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

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
              F_LMHEAD_X3, F_GEMM_X3P, F_GEMM_BF16P, F_LMHEAD_BF16, F_GEMM_H2P, F_LMHEAD_H2, F_PACK, F_COUNT };
static const char *kFamilyNames[F_COUNT] = {"gemm_f32", "gemm_f32_lmhead_topk", "attn_decode", "attn_prefill",
                                            "layernorm", "embed", "select", "attn_mapper", "other", "gemm_bf16x3",
                                            "gemm_bf16x3_lmhead_topk", "gemm_bf16x3p", "gemm_x1",
                                            "gemm_x1_lmhead_topk", "gemm_f16x2p", "gemm_f16x2p_lmhead_topk",
                                            "pack_activations"};
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

struct capdec_ctx {
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
    std::unordered_map<const void *, Planes> planes;   // fp32 weight -> (packed planes, elements, PackFmt)
    DBuf x3_tmp, xpk, apk, fpk, a_tmp;   // scratch planes for un-cached matrices; packed LayerNorm output; packed fp32-A
    int stat_steps = 0, stat_compactions = 0;      // last decode call: steps run, compactions done,
    long long stat_row_steps = 0;                  // activation rows pushed through the GPT-2 body (prefill excluded)
    bool batch_invariant = false;   // capdec_set_batch_invariant: no launch-size dependent summation order (no split-K, pinned kernel variants)
    int diverge = 0;                // measurement: beams never share history (capdec_set_debug_diverge)
    double stat_kv_slots = 0.0, stat_kv_pos = 0.0;   // last beam decode: sums behind capdec_decode_counters
    bool compact = true;       // decode: drop finished captions from the batch at the poll points (CAPDEC_COMPACT=0: off)
    bool pack_chain = true;    // ... and attention / the fc GEMM epilogue emit the packed A operand of the GEMM that follows
    bool pack_a = true;        // bf16x3 mode: LayerNorm emits the packed A operand, GEMM moves both operands by LDS-DMA
    hipEvent_t t0 = nullptr, t1 = nullptr;
    // workspaces
    DBuf h, x, qkv, att, ff, xl, tmax, tsum, cval, cidx, lse, topv, topi, kc, vc;
    DBuf tokens, scores, seq, stopped, done, anc, next_tok, alive, gids, glens, cmap, kvstat;
    DBuf m_hid, m_lin, m_seq, m_x, m_qkv, m_att, m_ff;
    DBuf t_idx, t_patch, t_pout, p_desc, p_inter, splitk, absmax;
    int *alive_host = nullptr;   // pinned
    // caption-shard communicator (RCCL), see capdec_comm_init
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    DBuf g_pad, g_all;
};

namespace capdec {

// ---------------------------------------------------------------------------- profiling
static hipEvent_t prof_event(capdec_ctx *c) {
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
static int prof_collect(capdec_ctx *c) {
    if (c->prof.recs.empty()) return 0;
    CAPDEC_HIP(hipStreamSynchronize(c->stream));
    for (auto &r : c->prof.recs) {
        float ms = 0.f;
        CAPDEC_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        c->prof.ms[r.fam] += ms;
        c->prof.flops[r.fam] += r.flops;
        c->prof.launches[r.fam] += 1;
        c->prof.pool.push_back(r.a);
        c->prof.pool.push_back(r.b);
    }
    c->prof.recs.clear();
    return 0;
}

// ---------------------------------------------------------------------------- uploads
static int upload(std::vector<void *> &owned, const float *h_src, size_t n, float **out) {
    CAPDEC_CHECK(h_src != nullptr, "weights: null host pointer");
    void *p = nullptr;
    CAPDEC_HIP(hipMalloc(&p, n * sizeof(float)));
    owned.push_back(p);
    CAPDEC_HIP(hipMemcpy(p, h_src, n * sizeof(float), hipMemcpyHostToDevice));
    *out = reinterpret_cast<float *>(p);
    return 0;
}
// host [rows, cols] -> device [cols, rows] (Conv1D [in,out] -> k-contiguous [out,in])
static int upload_transposed(capdec_ctx *c, std::vector<void *> &owned, const float *h_src, int rows, int cols,
                             float **out) {
    CAPDEC_CHECK(h_src != nullptr, "weights: null host pointer");
    void *tmp = nullptr, *p = nullptr;
    const size_t n = (size_t)rows * cols;
    CAPDEC_HIP(hipMalloc(&tmp, n * sizeof(float)));
    if (hipMalloc(&p, n * sizeof(float)) != hipSuccess) {
        (void)hipFree(tmp);
        set_error("weights: hipMalloc failed");
        return 1;
    }
    owned.push_back(p);
    int rc = 0;
    if (hipMemcpy(tmp, h_src, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = 1;
    if (!rc) rc = launch_transpose(c->stream, (const float *)tmp, (float *)p, rows, cols);
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = 1;
    (void)hipFree(tmp);
    if (rc) {
        set_error("weights: transpose upload failed");
        return 1;
    }
    *out = reinterpret_cast<float *>(p);
    return 0;
}
static void free_all(std::vector<void *> &owned) {
    for (void *p : owned) (void)hipFree(p);
    owned.clear();
}

// ---------------------------------------------------------------------------- GEMM wrappers
// bf16 planes of an [N, K] fp32 weight matrix: made on first use, dropped whenever weights are reloaded
static void drop_planes(capdec_ctx *c) {
    for (auto &kv : c->planes) (void)hipFree(kv.second.p);
    c->planes.clear();
}
// packed operand format of the block-stack / lm_head GEMMs in the current mode (bf16x3.h): two fp16 planes (f16x2),
// three bf16 planes (bf16x3), or ONE bf16 / fp16 plane (the reduced-precision modes)
static int pack_fmt(const capdec_ctx *c) {
    switch (c->gemm_mode) {
        case GEMM_F16X2: return PK_F16X2;
        case GEMM_BF16: return PK_BF16X1;
        case GEMM_F16: return PK_F16X1;
        default: return PK_BF16X3;
    }
}
static bool mode_single(const capdec_ctx *c) { return c->gemm_mode == GEMM_BF16 || c->gemm_mode == GEMM_F16; }
static int gemm_single(capdec_ctx *c, const void *A, const void *B, float *C, int ldc, int M, int N, int K,
                       const GemmEpilogue &e) {
    return launch_gemm_x1(c->stream, A, B, C, ldc, M, N, K, e, pack_fmt(c));
}
static int pack_any(capdec_ctx *c, const float *W, int N, int K, int fmt, void *out) {
    if (fmt == PK_F16X2) return launch_pack_planes_h2(c->stream, W, K, N, K, out);
    if (fmt == PK_BF16X3) return launch_pack_planes(c->stream, W, N, K, out);
    return launch_pack_planes_fmt(c->stream, W, K, N, K, out, fmt);
}
// max |w| < 16 ?  (one tiny reduction + a 4-byte read-back, once per cached weight)
static int weight_wide_ok(capdec_ctx *c, const float *W, size_t n, bool *ok) {
    CAPDEC_TRY(c->absmax.ensure(sizeof(unsigned)));
    CAPDEC_TRY(launch_absmax_bits(c->stream, W, n, c->absmax.as<unsigned>()));
    unsigned bits = 0;
    CAPDEC_HIP(hipMemcpyAsync(&bits, c->absmax.p, sizeof(bits), hipMemcpyDeviceToHost, c->stream));
    CAPDEC_HIP(hipStreamSynchronize(c->stream));
    float m;
    memcpy(&m, &bits, sizeof(m));
    *ok = m < 16.0f;
    return 0;
}
static int planes_of(capdec_ctx *c, const float *W, int N, int K, bool cache, const void **out, int fmt_override = -1,
                     bool *wide_ok = nullptr) {
    const int fmt = fmt_override >= 0 ? fmt_override : pack_fmt(c);
    const size_t n = (size_t)N * K, bytes = x3_packed_bytes(N, K, fmt);
    if (wide_ok) *wide_ok = false;
    if (cache) {
        auto it = c->planes.find(W);
        if (it != c->planes.end()) {
            if (it->second.n == n && it->second.fmt == fmt) {
                *out = it->second.p;
                if (wide_ok) *wide_ok = it->second.wide_ok;
                return 0;
            }
            (void)hipFree(it->second.p);      // same address, different matrix (or the GEMM mode changed)
            c->planes.erase(it);
        }
        void *p = nullptr;
        CAPDEC_HIP(hipMalloc(&p, bytes));
        bool ok = false;
        if (fmt == PK_F16X2) CAPDEC_TRY(weight_wide_ok(c, W, n, &ok));
        c->planes[W] = capdec_ctx::Planes{p, n, fmt, ok};
        CAPDEC_TRY(pack_any(c, W, N, K, fmt, p));
        *out = p;
        if (wide_ok) *wide_ok = ok;
        return 0;
    }
    CAPDEC_TRY(c->x3_tmp.ensure(bytes));
    CAPDEC_TRY(pack_any(c, W, N, K, fmt, c->x3_tmp.p));
    *out = c->x3_tmp.p;
    return 0;
}

static int gemm(capdec_ctx *c, const float *A, int lda, const float *Bt, int ldb, float *C, int ldc, int M, int N,
                int K, const float *bias, int act, const float *resid = nullptr, int ldr = 0, bool weight = true) {
    GemmEpilogue e;
    e.bias = bias;
    e.act = act;
    e.resid = resid;
    e.ldr = ldr;
    if ((c->gemm_mode == GEMM_F16X2 || mode_single(c)) && ldb == K && K % 64 == 0 && lda % 4 == 0 && M > 0) {
        // fp32 activations in HBM (mapper, patch embedding, CLIP projections): one packing pass (read 4 B, write 4 B
        // per element), then the packed LDS-DMA kernel -- fp32-accurate (f16x2) also in the reduced-precision modes,
        // whose 16-bit operands are confined to the GPT-2 / CLIP block stacks and the lm_head
        const void *pl = nullptr;
        CAPDEC_TRY(planes_of(c, Bt, N, K, weight, &pl, PK_F16X2, &e.wide_ok));
        if (c->batch_invariant) { e.wide_ok = false; e.invariant = true; }      // (the geometry planners look at M)
        CAPDEC_TRY(c->a_tmp.ensure(x3_packed_bytes(M, K, PK_F16X2)));
        { ProfScope ps(c, F_PACK); CAPDEC_TRY(launch_pack_planes_h2(c->stream, A, lda, M, K, c->a_tmp.p)); }
        const size_t wsb = c->batch_invariant ? 0 : gemm_splitk_ws_bytes(M, N, K);
        if (wsb) {
            CAPDEC_TRY(c->splitk.ensure(wsb));
            e.splitk_ws = c->splitk.p;
            e.splitk_ws_bytes = c->splitk.cap;
        }
        ProfScope ps(c, F_GEMM_H2P, 2.0 * M * (double)N * K);
        return launch_gemm_f16x2p(c->stream, c->a_tmp.p, pl, C, ldc, M, N, K, e);
    }
    // (bf16 mode: GEMMs whose A operand is fp32 in HBM -- mapper, patch embedding -- keep the split kernel)
    if (c->gemm_mode != GEMM_F32 && ldb == K && K % 64 == 0) {   // other K: native fp32 MFMA
        const void *pl = nullptr;
        CAPDEC_TRY(planes_of(c, Bt, N, K, weight, &pl));
        ProfScope ps(c, F_GEMM_X3, 2.0 * M * (double)N * K);
        return launch_gemm_bf16x3(c->stream, A, lda, pl, C, ldc, M, N, K, e);
    }
    ProfScope ps(c, F_GEMM, 2.0 * M * (double)N * K);
    return launch_gemm_f32(c->stream, A, lda, Bt, ldb, C, ldc, M, N, K, e);
}

// LayerNorm -> GEMM with the normalised rows handed over in packed split-bf16 form (never fp32 in HBM).
// Returns 1 in *done when the packed path ran; otherwise the caller runs the fp32-activation path.
static bool use_packed_a(capdec_ctx *c, int K) {
    return (mode_single(c) || c->gemm_mode == GEMM_F16X2 || (c->gemm_mode == GEMM_BF16X3 && c->pack_a)) && K % 64 == 0;
}

// C = act(Apk . W^T + bias) + resid with A already packed; packed_out != nullptr: the result is written as the
// packed A operand of the next GEMM instead of fp32 C
// (next_ln: the LayerNorm that follows this GEMM in the block stack; when the launch splits K it is fused into the
//  reduce pass, its packed output lands in c->xpk and *ln_done is set -- see GemmEpilogue)
struct NextLn { const float *w, *b; float eps; int *done; };
static int gemm_packed(capdec_ctx *c, const void *Apk, const float *W, float *C, int ldc, int M, int N, int K,
                       const float *bias, int act, const float *resid = nullptr, int ldr = 0,
                       void *packed_out = nullptr, const NextLn *next_ln = nullptr, const void *resid_packed = nullptr,
                       const QkvScatter *qkv_scatter = nullptr) {
    const void *pl = nullptr;
    GemmEpilogue e;
    e.qkv_scatter = qkv_scatter;
    CAPDEC_TRY(planes_of(c, W, N, K, true, &pl, -1, &e.wide_ok));
    if (c->batch_invariant) { e.wide_ok = false; e.invariant = true; }          // (the geometry planners look at M)
    e.bias = bias;
    e.act = act;
    e.resid = resid;
    e.ldr = ldr;
    e.packed_out = packed_out;
    e.resid_packed = resid_packed;
    if (next_ln && next_ln->w && ldc == N && (const void *)Apk != c->xpk.p) {
        CAPDEC_TRY(c->xpk.ensure(x3_packed_bytes_host(M, N)));
        e.ln_w = next_ln->w; e.ln_b = next_ln->b; e.ln_eps = next_ln->eps; e.ln_out = c->xpk.p; e.ln_done = next_ln->done;
    }
    if (c->gemm_mode != GEMM_F32) {
        static const bool x1_split = [] { const char *e = getenv("CAPDEC_X1_SPLITK"); return !(e && atoi(e) == 0); }();
        const size_t wsb = ((mode_single(c) && !x1_split) || c->batch_invariant || qkv_scatter) ? 0 : gemm_splitk_ws_bytes(M, N, K);
        if (wsb) {
            CAPDEC_TRY(c->splitk.ensure(wsb));
            e.splitk_ws = c->splitk.p;
            e.splitk_ws_bytes = c->splitk.cap;
        }
    }
    if (mode_single(c)) {   // one 16-bit plane per operand (bf16 / fp16), one MFMA per product
        ProfScope ps(c, F_GEMM_BF16P, 2.0 * M * (double)N * K);
        return gemm_single(c, Apk, pl, C, ldc, M, N, K, e);
    }
    if (c->gemm_mode == GEMM_F16X2) {   // two fp16 planes, three MFMAs per product (fp32-accurate)
        ProfScope ps(c, F_GEMM_H2P, 2.0 * M * (double)N * K);
        return launch_gemm_f16x2p(c->stream, Apk, pl, C, ldc, M, N, K, e);
    }
    ProfScope ps(c, F_GEMM_X3P, 2.0 * M * (double)N * K);
    return launch_gemm_bf16x3p(c->stream, Apk, pl, C, ldc, M, N, K, e);
}

// (ln_ready: c->xpk already holds LayerNorm(h) -- written by the fused split-K reduce of the previous GEMM)
static int ln_gemm_packed(capdec_ctx *c, const float *h, int ldh, const float *lnw, const float *lnb, float eps,
                          const float *W, float *C, int ldc, int M, int N, int K, const float *bias, int act,
                          void *packed_out = nullptr, bool ln_ready = false, const QkvScatter *qkv_scatter = nullptr) {
    CAPDEC_TRY(c->xpk.ensure(x3_packed_bytes_host(M, K)));
    if (!ln_ready) {
        ProfScope ps(c, F_LN);
        CAPDEC_TRY(launch_layernorm_packed(c->stream, h, ldh, lnw, lnb, eps, c->xpk.p, M, K, pack_fmt(c)));
    }
    return gemm_packed(c, c->xpk.p, W, C, ldc, M, N, K, bias, act, nullptr, 0, packed_out, nullptr, nullptr, qkv_scatter);
}

// ---------------------------------------------------------------------------- GPT-2 body
struct StepShape {
    bool prefill;
    int ncap, P, beam;      // prefill: rows = ncap * P
    int rows, L;            // decode: rows at context length L
    const uint8_t *anc;
    int anc_stride;
    const int *cmap;        // decode after compaction: activation row r -> caption cmap[r / beam] (nullptr: identity)
};

static int ensure_body_ws(capdec_ctx *c, int M, int d) {
    CAPDEC_TRY(c->h.ensure((size_t)M * d * 4));
    CAPDEC_TRY(c->x.ensure((size_t)M * d * 4));
    CAPDEC_TRY(c->qkv.ensure((size_t)M * 3 * d * 4));
    CAPDEC_TRY(c->att.ensure((size_t)M * d * 4));
    CAPDEC_TRY(c->ff.ensure((size_t)M * 4 * d * 4));
    return 0;
}

// h [M, d] (in c->h) -> h after all blocks (final LN NOT applied).  The same pre-LN block serves
// GPT-2 (gelu_new, causal, KV cache kept for the decode steps) and the CLIP towers (QuickGELU, attention straight from
// the qkv activations, nothing cached; the vision tower is not causal).
struct StackCfg {
    const std::vector<Gpt2Layer> *layers;
    int n_layer, d;
    float eps;
    int act;
    bool causal, keep_kv;
};
static int stack_body(capdec_ctx *c, const StackCfg &g, const StepShape &s, const KvCache &kv) {
    const int d = g.d, M = s.prefill ? s.ncap * s.P : s.rows;
    float *h = c->h.as<float>(), *x = c->x.as<float>(), *qkv = c->qkv.as<float>(), *att = c->att.as<float>(),
          *ff = c->ff.as<float>();
    // packed chain (bf16x3 mode): LN1 -> [packed] -> qkv GEMM -> attention -> [packed] -> c_proj (+h) -> LN2 ->
    // [packed] -> fc GEMM + act -> [packed] -> mlp c_proj (+h): every GEMM operand moves by LDS-DMA, and the
    // attention / MLP intermediates never exist in fp32 in HBM.
    const bool chain = use_packed_a(c, d) && c->pack_chain;
    static const bool kv_direct = [] { const char *e = getenv("CAPDEC_KV_DIRECT"); return !(e && atoi(e) == 0); }();
    void *apk = nullptr, *fpk = nullptr;
    if (chain) {
        CAPDEC_TRY(c->apk.ensure(x3_packed_bytes_host(M, d)));
        CAPDEC_TRY(c->fpk.ensure(x3_packed_bytes_host(M, 4 * d)));
        apk = c->apk.p;
        fpk = c->fpk.p;
    }
    int ln1_ready = 0;      // xpk already holds this layer's LN1(h): fused into the previous layer's mlp c_proj reduce
    for (int l = 0; l < g.n_layer; ++l) {
        const Gpt2Layer &w = (*g.layers)[l];
        const int kl = g.keep_kv ? l : 0;
        // decode steps in the default mode: K / V of the new token go from the qkv GEMM's epilogue straight into the cache
        // (QkvScatter) when that GEMM runs unsplit -- the attention then reads them like any other position
        QkvScatter sc;
        const bool scatter = kv_direct && !s.prefill && g.keep_kv && c->gemm_mode == GEMM_F16X2 && use_packed_a(c, d) && !kv.bf16 && w.bqkv &&
                             d % GEMM_BN == 0 && (s.beam == 1 || s.beam == 5) &&
                             (c->batch_invariant || gemm_splitk_slices(M, 3 * d, d) == 1);
        if (scatter) {
            sc.kc = kv.kp<float>(kl); sc.vc = kv.vp<float>(kl); sc.cmap = s.cmap;
            sc.beam = s.beam; sc.heads = kv.heads; sc.ctx = kv.ctx; sc.pos = s.L - 1; sc.d = d;
        }
        if (use_packed_a(c, d)) {
            CAPDEC_TRY(ln_gemm_packed(c, h, d, w.ln1w, w.ln1b, g.eps, w.wqkv, qkv, 3 * d, M, 3 * d, d, w.bqkv, CAPDEC_ACT_NONE,
                                      nullptr, ln1_ready != 0, scatter ? &sc : nullptr));
            ln1_ready = 0;
        } else {
            { ProfScope ps(c, F_LN); CAPDEC_TRY(launch_layernorm(c->stream, h, d, w.ln1w, w.ln1b, g.eps, x, d, M, d)); }
            CAPDEC_TRY(gemm(c, x, d, w.wqkv, d, qkv, 3 * d, M, 3 * d, d, w.bqkv, CAPDEC_ACT_NONE));
        }
        if (s.prefill) {
            ProfScope ps(c, F_ATTN_PRE);
            if (g.keep_kv)      // the towers never decode: only GPT-2 needs its prefix K/V in the cache
                CAPDEC_TRY(launch_kv_scatter_prefill(c->stream, qkv, kv, kl, s.ncap, s.P, s.beam));
            CAPDEC_TRY(launch_attn_prefill(c->stream, qkv, kv, kl, s.ncap, s.P, s.beam, att, g.causal, apk, pack_fmt(c)));
        } else {
            ProfScope ps(c, F_ATTN_DEC);
            CAPDEC_TRY(launch_attn_decode(c->stream, qkv, kv, kl, s.rows, s.beam, s.L, s.anc, s.anc_stride, att, apk, s.cmap, pack_fmt(c),
                                          scatter));
        }
        int ln2_ready = 0;
        if (chain) {
            const NextLn n2{w.ln2w, w.ln2b, g.eps, &ln2_ready};
            CAPDEC_TRY(gemm_packed(c, apk, w.wproj, h, d, M, d, d, w.bproj, CAPDEC_ACT_NONE, h, d, nullptr, &n2));
        } else {
            CAPDEC_TRY(gemm(c, att, d, w.wproj, d, h, d, M, d, d, w.bproj, CAPDEC_ACT_NONE, h, d));
        }
        if (chain) {
            CAPDEC_TRY(ln_gemm_packed(c, h, d, w.ln2w, w.ln2b, g.eps, w.wfc, ff, 4 * d, M, 4 * d, d, w.bfc, g.act, fpk,
                                      ln2_ready != 0));
            // the LayerNorm after mlp c_proj is the NEXT layer's ln_1 (the final ln_f runs on its own: it may see strided rows)
            const bool has_next = l + 1 < g.n_layer;
            const NextLn n1{has_next ? (*g.layers)[l + 1].ln1w : nullptr, has_next ? (*g.layers)[l + 1].ln1b : nullptr, g.eps,
                            &ln1_ready};
            CAPDEC_TRY(gemm_packed(c, fpk, w.wproj2, h, d, M, d, 4 * d, w.bproj2, CAPDEC_ACT_NONE, h, d, nullptr,
                                   has_next ? &n1 : nullptr));
            continue;
        }
        if (use_packed_a(c, d)) {
            CAPDEC_TRY(ln_gemm_packed(c, h, d, w.ln2w, w.ln2b, g.eps, w.wfc, ff, 4 * d, M, 4 * d, d, w.bfc, g.act));
        } else {
            { ProfScope ps(c, F_LN); CAPDEC_TRY(launch_layernorm(c->stream, h, d, w.ln2w, w.ln2b, g.eps, x, d, M, d)); }
            CAPDEC_TRY(gemm(c, x, d, w.wfc, d, ff, 4 * d, M, 4 * d, d, w.bfc, g.act));
        }
        CAPDEC_TRY(gemm(c, ff, 4 * d, w.wproj2, 4 * d, h, d, M, d, 4 * d, w.bproj2, CAPDEC_ACT_NONE, h, d));
    }
    return 0;
}

static int gpt2_body(capdec_ctx *c, const StepShape &s, const KvCache &kv) {
    const Gpt2 &g = c->gpt;
    StackCfg cfg{&g.layers, g.n_layer, g.d, g.eps, CAPDEC_ACT_GELU_NEW, true, true};
    return stack_body(c, cfg, s, kv);
}

// ln_f over `R` rows of h (row stride ldh floats, starting at h0) then the fused lm_head:
// -> lse [R], topv/topi [R, k]
static int lm_head_select(capdec_ctx *c, const float *h0, int ldh, int R, int k, float inv_temp) {
    const Gpt2 &g = c->gpt;
    const int d = g.d, nt = gemm_tiles_n(g.vocab);
    CAPDEC_TRY(c->xl.ensure((size_t)R * d * 4));
    CAPDEC_TRY(c->tmax.ensure((size_t)R * nt * 4));
    CAPDEC_TRY(c->tsum.ensure((size_t)R * nt * 4));
    CAPDEC_TRY(c->cval.ensure((size_t)R * nt * k * 4));
    CAPDEC_TRY(c->cidx.ensure((size_t)R * nt * k * 4));
    CAPDEC_TRY(c->lse.ensure((size_t)R * 4));
    CAPDEC_TRY(c->topv.ensure((size_t)R * k * 4));
    CAPDEC_TRY(c->topi.ensure((size_t)R * k * 4));
    if (use_packed_a(c, d)) {
        CAPDEC_TRY(c->xpk.ensure(x3_packed_bytes_host(R, d)));
        { ProfScope ps(c, F_LN); CAPDEC_TRY(launch_layernorm_packed(c->stream, h0, ldh, g.lnfw, g.lnfb, g.eps, c->xpk.p, R, d, pack_fmt(c))); }
        const void *pl = nullptr;
        bool wide_ok = false;
        CAPDEC_TRY(planes_of(c, g.wte, g.vocab, d, true, &pl, -1, &wide_ok));
        if (c->gemm_mode == GEMM_F16X2) {
            ProfScope ps(c, F_LMHEAD_H2, 2.0 * R * (double)g.vocab * d);
            // 256 x 128 tiles with one accumulator set once the grid is many rounds deep (each wte panel is then fetched
            // by half as many row tiles); small row counts keep the 128-row tile (more blocks, the same partial lists)
            static const int lm_wide = [] { const char *e = getenv("CAPDEC_LMHEAD_WIDE"); return e ? atoi(e) : 1; }();
            if (wide_ok && !c->batch_invariant && ((lm_wide && h2w_choice() >= 1 && R >= 2048) || h2w_choice() >= 2))   // (CAPDEC_H2W >= 2: forced, tests)
                CAPDEC_TRY(launch_gemm_h2w_topk(c->stream, c->xpk.p, pl, R, g.vocab, d, k, inv_temp, c->tmax.as<float>(),
                                                c->tsum.as<float>(), c->cval.as<float>(), c->cidx.as<int>()));
            else
            CAPDEC_TRY(launch_gemm_f16x2p_topk(c->stream, c->xpk.p, pl, R, g.vocab, d, k, inv_temp, c->tmax.as<float>(),
                                               c->tsum.as<float>(), c->cval.as<float>(), c->cidx.as<int>()));
        } else if (mode_single(c)) {
            ProfScope ps(c, F_LMHEAD_BF16, 2.0 * R * (double)g.vocab * d);
            CAPDEC_TRY(launch_gemm_x1_topk(c->stream, c->xpk.p, pl, R, g.vocab, d, k, inv_temp, c->tmax.as<float>(),
                                           c->tsum.as<float>(), c->cval.as<float>(), c->cidx.as<int>(), pack_fmt(c)));
        } else {
            ProfScope ps(c, F_LMHEAD_X3, 2.0 * R * (double)g.vocab * d);
            CAPDEC_TRY(launch_gemm_bf16x3p_topk(c->stream, c->xpk.p, pl, R, g.vocab, d, k, inv_temp,
                                                c->tmax.as<float>(), c->tsum.as<float>(), c->cval.as<float>(),
                                                c->cidx.as<int>()));
        }
    } else if (c->gemm_mode != GEMM_F32) {
        { ProfScope ps(c, F_LN); CAPDEC_TRY(launch_layernorm(c->stream, h0, ldh, g.lnfw, g.lnfb, g.eps, c->xl.as<float>(), d, R, d)); }
        const void *pl = nullptr;
        CAPDEC_TRY(planes_of(c, g.wte, g.vocab, d, true, &pl));
        ProfScope ps(c, F_LMHEAD_X3, 2.0 * R * (double)g.vocab * d);
        CAPDEC_TRY(launch_gemm_bf16x3_topk(c->stream, c->xl.as<float>(), d, pl, R, g.vocab, d, k, inv_temp,
                                           c->tmax.as<float>(), c->tsum.as<float>(), c->cval.as<float>(),
                                           c->cidx.as<int>()));
    } else {
        { ProfScope ps(c, F_LN); CAPDEC_TRY(launch_layernorm(c->stream, h0, ldh, g.lnfw, g.lnfb, g.eps, c->xl.as<float>(), d, R, d)); }
        ProfScope ps(c, F_LMHEAD, 2.0 * R * (double)g.vocab * d);
        CAPDEC_TRY(launch_gemm_f32_topk(c->stream, c->xl.as<float>(), d, g.wte, d, R, g.vocab, d, k, inv_temp,
                                        c->tmax.as<float>(), c->tsum.as<float>(), c->cval.as<float>(),
                                        c->cidx.as<int>()));
    }
    {
        ProfScope ps(c, F_SELECT);
        CAPDEC_TRY(launch_topk_merge(c->stream, c->tmax.as<float>(), c->tsum.as<float>(), c->cval.as<float>(),
                                     c->cidx.as<int>(), R, nt, k, c->lse.as<float>(), c->topv.as<float>(),
                                     c->topi.as<int>()));
    }
    return 0;
}

// geometry only (the CLIP towers attend straight from the qkv activations and never touch a cache)
static void kv_geometry(KvCache &kv, int rows, int ctx, int heads, int hd) {
    kv.rows = rows;
    kv.heads = heads;
    kv.ctx = ctx;
    kv.hd = hd;
    kv.k = kv.v = nullptr;
}
static int ensure_kv(capdec_ctx *c, KvCache &kv, int rows, int ctx, int heads = 0, int hd = 0, int layers = 0) {
    const Gpt2 &g = c->gpt;
    kv.rows = rows;
    kv.heads = heads ? heads : g.n_head;
    kv.ctx = ctx;
    kv.hd = hd ? hd : g.d / g.n_head;
    kv.bf16 = c->gemm_mode == GEMM_BF16;      // BASELINE configs[1]: bf16 weights / GEMM operands / KV cache
    const size_t bytes = kv.layer_stride() * (layers ? layers : g.n_layer) * kv.elem_bytes();
    CAPDEC_TRY(c->kc.ensure(bytes));
    CAPDEC_TRY(c->vc.ensure(bytes));
    kv.k = c->kc.p;
    kv.v = c->vc.p;
    return 0;
}

static int poll_alive(capdec_ctx *c, int *alive) {
    CAPDEC_HIP(hipMemcpyAsync(c->alive_host, c->alive.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    CAPDEC_HIP(hipStreamSynchronize(c->stream));
    *alive = *c->alive_host;
    return 0;
}

// captions per chunk so that the fp32 KV cache fits the budget: the configured budget (capdec_set_kv_budget, default
// 192 GiB), clamped to 85 % of what the device can still give (free memory + what the KV buffers already hold), so
// a GPU that is partly occupied -- torch's caching allocator, the CLIP towers, a smaller part -- gets smaller chunks
// instead of a failed hipMalloc
static int chunk_captions(capdec_ctx *c, int n, int beam, int ctx) {
    const Gpt2 &g = c->gpt;
    const size_t per_cap = (size_t)beam * ctx * g.d * 2 * (c->gemm_mode == GEMM_BF16 ? 2 : 4) * g.n_layer;
    size_t budget = c->kv_budget, free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const size_t avail = (size_t)((double)(free_b + c->kc.cap + c->vc.cap) * 0.85);
        budget = std::min(budget, avail);
    }
    size_t m = budget / std::max<size_t>(per_cap, 1);
    m = std::max<size_t>(m, 1);
    return (int)std::min<size_t>(m, (size_t)n);
}

// ---------------------------------------------------------------------------- decode drivers
static int decode_chunk(capdec_ctx *c, const float *prefix, int nc, int P, int beam, bool greedy, int stop_id,
                        int alt_stop_id, int T, float temperature, int *ids, int *lens, float *scores, int *order,
                        const int *forced = nullptr, float *stats = nullptr) {
    const Gpt2 &g = c->gpt;
    const int d = g.d;
    const int ctx = P + T - 1;
    const int rows = nc * beam;
    const int k = (greedy && stats) ? 2 : beam;   // candidates kept per row (teacher-forced statistics: top-2)
    const float inv_temp = 1.0f / (temperature > 0.f ? temperature : 1.0f);
    KvCache kv;
    CAPDEC_TRY(ensure_kv(c, kv, rows, ctx));
    kv.fixed_variant = c->batch_invariant;
    kv.prefix_len = P;
    CAPDEC_TRY(ensure_body_ws(c, std::max(nc * P, rows), d));
    CAPDEC_TRY(c->next_tok.ensure((size_t)rows * 4));
    CAPDEC_TRY(c->alive.ensure(sizeof(int)));
    CAPDEC_TRY(c->done.ensure((size_t)rows));
    BeamState bs;
    if (!greedy) {
        CAPDEC_TRY(c->tokens.ensure((size_t)rows * T * 4));
        CAPDEC_TRY(c->scores.ensure((size_t)rows * 4));
        CAPDEC_TRY(c->seq.ensure((size_t)rows * 4));
        CAPDEC_TRY(c->stopped.ensure((size_t)rows));
        CAPDEC_TRY(c->anc.ensure((size_t)rows * ctx));
        bs.tokens = c->tokens.as<int>();
        bs.scores = c->scores.as<float>();
        bs.seq = c->seq.as<float>();
        bs.stopped = c->stopped.as<uint8_t>();
        bs.done = c->done.as<uint8_t>();
        bs.anc = c->anc.as<uint8_t>();
        bs.next_tok = c->next_tok.as<int>();
        bs.alive_count = c->alive.as<int>();
        bs.diverge = c->diverge;
        CAPDEC_TRY(c->kvstat.ensure((size_t)nc * 2 * sizeof(unsigned)));
        bs.kv_stat = c->kvstat.as<unsigned>();
        CAPDEC_HIP(hipMemsetAsync(bs.kv_stat, 0, (size_t)nc * 2 * sizeof(unsigned), c->stream));
        CAPDEC_HIP(hipMemsetAsync(bs.tokens, 0, (size_t)rows * T * 4, c->stream));
        CAPDEC_HIP(hipMemsetAsync(bs.anc, 0, (size_t)rows * ctx, c->stream));
    } else {
        CAPDEC_HIP(hipMemsetAsync(ids, 0, (size_t)nc * T * 4, c->stream));
        CAPDEC_HIP(hipMemsetAsync(lens, 0, (size_t)nc * 4, c->stream));
    }
    CAPDEC_HIP(hipMemsetAsync(c->done.p, 0, (size_t)rows, c->stream));
    CAPDEC_HIP(hipMemsetAsync(c->alive.p, 0, sizeof(int), c->stream));

    // ---- step 0: prefill the prefix (positions 0..P-1), logits of the last prefix row
    { ProfScope ps(c, F_EMBED); CAPDEC_TRY(launch_embed_prefix(c->stream, prefix, g.wpe, c->h.as<float>(), nc, P, 0, d)); }
    StepShape sp{};
    sp.prefill = true;
    sp.ncap = nc;
    sp.P = P;
    sp.beam = beam;
    CAPDEC_TRY(gpt2_body(c, sp, kv));
    CAPDEC_TRY(lm_head_select(c, c->h.as<float>() + (size_t)(P - 1) * d, P * d, nc, k, inv_temp));
    if (greedy) {
        ProfScope ps(c, F_SELECT);
        CAPDEC_TRY(launch_greedy_step(c->stream, c->topi.as<int>(), nc, 0, T, stop_id, alt_stop_id, ids, lens,
                                      c->done.as<uint8_t>(), c->next_tok.as<int>(), c->alive.as<int>(), nullptr, k, forced,
                                      c->topv.as<float>(), c->lse.as<float>(), stats));
    } else {
        ProfScope ps(c, F_SELECT);
        CAPDEC_TRY(launch_beam_init(c->stream, bs, c->lse.as<float>(), c->topv.as<float>(), c->topi.as<int>(), nc,
                                    beam, k, T, ctx, P, stop_id));
    }
    // ---- steps 1..T-1: one token per row per step.  Finished captions (stop token on every beam) are dropped from
    // the batch at the poll points: `cmap` lists the captions still generating, the activations of a step are the
    // na * beam rows of those captions only, while KV cache / ancestor table / beam state keep their original rows.
    const int poll_every = 8;
    int na = nc;
    const int *cmap = nullptr;
    CAPDEC_TRY(c->cmap.ensure(((size_t)nc + 1) * 4));
    for (int i = 1; i < T; ++i) {
        if ((i - 1) % poll_every == 0) {
            int alive = 0;
            CAPDEC_TRY(poll_alive(c, &alive));
            if (alive == 0) break;
            if (c->compact && alive <= na - std::max(1, na / 32)) {
                ProfScope ps(c, F_SELECT);
                CAPDEC_TRY(launch_compact_alive(c->stream, c->done.as<uint8_t>(), nc, c->cmap.as<int>(),
                                                c->cmap.as<int>() + nc));
                na = alive;
                cmap = c->cmap.as<int>();
                c->stat_compactions += 1;
            }
        }
        const int pos = P + i - 1;   // position of the token fed this step
        const int arows = na * beam;
        c->stat_steps = std::max(c->stat_steps, i + 1);
        c->stat_row_steps += arows;
        CAPDEC_HIP(hipMemsetAsync(c->alive.p, 0, sizeof(int), c->stream));
        {
            ProfScope ps(c, F_EMBED);
            CAPDEC_TRY(launch_embed_tokens(c->stream, c->next_tok.as<int>(), g.wte, g.wpe + (size_t)pos * d,
                                           c->h.as<float>(), arows, d, cmap, beam));
        }
        StepShape sd{};
        sd.prefill = false;
        sd.rows = arows;
        sd.beam = beam;
        sd.L = pos + 1;
        sd.anc = greedy ? nullptr : bs.anc;
        sd.anc_stride = ctx;
        sd.cmap = cmap;
        CAPDEC_TRY(gpt2_body(c, sd, kv));
        CAPDEC_TRY(lm_head_select(c, c->h.as<float>(), d, arows, k, inv_temp));
        ProfScope ps(c, F_SELECT);
        if (greedy) {
            CAPDEC_TRY(launch_greedy_step(c->stream, c->topi.as<int>(), arows, i, T, stop_id, alt_stop_id, ids, lens,
                                          c->done.as<uint8_t>(), c->next_tok.as<int>(), c->alive.as<int>(), cmap, k, forced,
                                          c->topv.as<float>(), c->lse.as<float>(), stats));
        } else {
            CAPDEC_TRY(launch_beam_step(c->stream, bs, c->lse.as<float>(), c->topv.as<float>(), c->topi.as<int>(), na,
                                        beam, k, T, ctx, i, pos, g.vocab, stop_id, cmap));
        }
    }
    if (!greedy) {
        ProfScope ps(c, F_SELECT);
        CAPDEC_TRY(launch_beam_finalize(c->stream, bs, nc, beam, T, ids, lens, scores, order));
    }
    if (!greedy && bs.kv_stat) {     // (one small read-back per chunk, after the last kernel of the chunk was enqueued)
        std::vector<unsigned> hs((size_t)nc * 2);
        CAPDEC_HIP(hipMemcpyAsync(hs.data(), bs.kv_stat, hs.size() * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        CAPDEC_HIP(hipStreamSynchronize(c->stream));
        for (int i = 0; i < nc; ++i) { c->stat_kv_slots += hs[2 * i]; c->stat_kv_pos += hs[2 * i + 1]; }
    }
    return 0;
}

static int decode_common(capdec_ctx *c, const float *prefix, int n, int P, int beam, bool greedy, int stop_id,
                         int alt_stop_id, int T, float temperature, int *ids, int *lens, float *scores,
                         int *order, const int *forced = nullptr, float *stats = nullptr) {
    CAPDEC_CHECK(c && c->gpt.loaded, "decode: GPT-2 weights not loaded");
    CAPDEC_CHECK(n >= 0 && P >= 1 && T >= 1, "decode: bad sizes");
    CAPDEC_CHECK(P + T - 1 <= c->gpt.n_pos, "decode: prefix + entry_length exceeds n_positions");
    CAPDEC_CHECK(P + T - 1 <= 256 && T <= 128, "decode: context > 256 or entry_length > 128 not supported");
    CAPDEC_CHECK(beam >= 1 && beam <= 8, "decode: beam size must be in 1..8");
    CAPDEC_CHECK(c->gpt.d / c->gpt.n_head == 64, "decode: head_dim must be 64");
    CAPDEC_HIP(hipSetDevice(c->device));
    c->stat_steps = n > 0 ? 1 : 0;
    c->stat_compactions = 0;
    c->stat_row_steps = 0;
    c->stat_kv_slots = c->stat_kv_pos = 0.0;
    if (n == 0) return 0;
    const int ctx = P + T - 1;
    const int chunk = chunk_captions(c, n, beam, ctx);
    for (int c0 = 0; c0 < n; c0 += chunk) {
        const int nc = std::min(chunk, n - c0);
        CAPDEC_TRY(decode_chunk(c, prefix + (size_t)c0 * P * c->gpt.d, nc, P, beam, greedy, stop_id, alt_stop_id, T,
                                temperature, ids + (size_t)c0 * beam * T, lens + (size_t)c0 * beam,
                                scores ? scores + (size_t)c0 * beam : nullptr,
                                order ? order + (size_t)c0 * beam : nullptr,
                                forced ? forced + (size_t)c0 * T : nullptr, stats ? stats + (size_t)c0 * T * 3 : nullptr));
    }
    CAPDEC_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

// ---------------------------------------------------------------------------- mapper forward
static int mapper_chunk(capdec_ctx *c, const float *x, int n, float *out) {
    Mapper &m = c->map;
    const int d = m.d;
    if (m.kind == 1) {
        CAPDEC_TRY(c->m_hid.ensure((size_t)n * m.hidden * 4));
        CAPDEC_TRY(gemm(c, x, m.D, m.w1, m.D, c->m_hid.as<float>(), m.hidden, n, m.hidden, m.D, m.b1, CAPDEC_ACT_TANH));
        CAPDEC_TRY(gemm(c, c->m_hid.as<float>(), m.hidden, m.w2, m.hidden, out, m.P * d, n, m.P * d, m.hidden, m.b2,
                        CAPDEC_ACT_NONE));
        return 0;
    }
    const int S = m.clip_len + m.P, M = n * S, hd = d / m.heads;
    CAPDEC_TRY(c->m_lin.ensure((size_t)n * m.clip_len * d * 4));
    CAPDEC_TRY(c->m_seq.ensure((size_t)M * d * 4));
    CAPDEC_TRY(c->m_x.ensure((size_t)M * d * 4));
    CAPDEC_TRY(c->m_qkv.ensure((size_t)M * 3 * d * 4));
    CAPDEC_TRY(c->m_att.ensure((size_t)M * d * 4));
    CAPDEC_TRY(c->m_ff.ensure((size_t)M * m.mlp_hidden * 4));
    float *seq = c->m_seq.as<float>(), *xn = c->m_x.as<float>(), *qkv = c->m_qkv.as<float>(),
          *att = c->m_att.as<float>(), *ff = c->m_ff.as<float>();
    CAPDEC_TRY(gemm(c, x, m.D, m.lin_w, m.D, c->m_lin.as<float>(), m.clip_len * d, n, m.clip_len * d, m.D, m.lin_b,
                    CAPDEC_ACT_NONE));
    { ProfScope ps(c, F_OTHER); CAPDEC_TRY(launch_tmapper_concat(c->stream, c->m_lin.as<float>(), m.prefix_const, seq, n, m.clip_len, m.P, d)); }
    for (int l = 0; l < m.n_layers; ++l) {
        const TMapLayer &w = m.layers[l];
        { ProfScope ps(c, F_LN); CAPDEC_TRY(launch_layernorm(c->stream, seq, d, w.n1w, w.n1b, 1e-5f, xn, d, M, d)); }
        CAPDEC_TRY(gemm(c, xn, d, w.wqkv, d, qkv, 3 * d, M, 3 * d, d, nullptr, CAPDEC_ACT_NONE));
        { ProfScope ps(c, F_MAP_ATTN); CAPDEC_TRY(launch_attn_mapper(c->stream, qkv, 3 * d, qkv + d, qkv + 2 * d, 3 * d, att, n, S, m.heads, hd)); }
        CAPDEC_TRY(gemm(c, att, d, w.wproj, d, seq, d, M, d, d, w.bproj, CAPDEC_ACT_NONE, seq, d));
        { ProfScope ps(c, F_LN); CAPDEC_TRY(launch_layernorm(c->stream, seq, d, w.n2w, w.n2b, 1e-5f, xn, d, M, d)); }
        CAPDEC_TRY(gemm(c, xn, d, w.wfc1, d, ff, m.mlp_hidden, M, m.mlp_hidden, d, w.bfc1, CAPDEC_ACT_RELU));
        CAPDEC_TRY(gemm(c, ff, m.mlp_hidden, w.wfc2, m.mlp_hidden, seq, d, M, d, m.mlp_hidden, w.bfc2, CAPDEC_ACT_NONE,
                        seq, d));
    }
    { ProfScope ps(c, F_OTHER); CAPDEC_TRY(launch_tmapper_take(c->stream, seq, out, n, m.clip_len, m.P, d)); }
    return 0;
}

// ---------------------------------------------------------------------------- CLIP towers
static int upload_blocks(capdec_ctx *c, Tower &t, const capdec_clip_block *blocks) {
    const int d = t.d;
    t.layers.resize(t.n_layer);
    for (int l = 0; l < t.n_layer; ++l) {
        const capdec_clip_block &s = blocks[l];
        Gpt2Layer &w = t.layers[l];
        CAPDEC_TRY(upload(t.owned, s.ln_1_w, d, &w.ln1w));
        CAPDEC_TRY(upload(t.owned, s.ln_1_b, d, &w.ln1b));
        CAPDEC_TRY(upload(t.owned, s.in_proj_w, (size_t)3 * d * d, &w.wqkv));     // already [out, in]
        CAPDEC_TRY(upload(t.owned, s.in_proj_b, 3 * d, &w.bqkv));
        CAPDEC_TRY(upload(t.owned, s.out_proj_w, (size_t)d * d, &w.wproj));
        CAPDEC_TRY(upload(t.owned, s.out_proj_b, d, &w.bproj));
        CAPDEC_TRY(upload(t.owned, s.ln_2_w, d, &w.ln2w));
        CAPDEC_TRY(upload(t.owned, s.ln_2_b, d, &w.ln2b));
        CAPDEC_TRY(upload(t.owned, s.c_fc_w, (size_t)4 * d * d, &w.wfc));
        CAPDEC_TRY(upload(t.owned, s.c_fc_b, 4 * d, &w.bfc));
        CAPDEC_TRY(upload(t.owned, s.c_proj_w, (size_t)4 * d * d, &w.wproj2));
        CAPDEC_TRY(upload(t.owned, s.c_proj_b, d, &w.bproj2));
    }
    return 0;
}

// one chunk of captions through the text tower: tokens [n, ctx] -> out [n, embed]
static int clip_text_chunk(capdec_ctx *c, const int *tokens, int n, float *out) {
    Tower &t = c->clip_text;
    const int d = t.d, L = t.ctx;
    KvCache kv;
    kv_geometry(kv, n, L, t.n_head, d / t.n_head);
    CAPDEC_TRY(ensure_body_ws(c, n * L, d));
    CAPDEC_TRY(c->t_idx.ensure((size_t)n * 4));
    CAPDEC_TRY(c->xl.ensure((size_t)2 * n * d * 4));
    { ProfScope ps(c, F_EMBED); CAPDEC_TRY(launch_clip_text_embed(c->stream, tokens, t.tok_emb, t.pos_emb, c->h.as<float>(), n, L, d)); }
    StepShape sp{};
    sp.prefill = true;
    sp.ncap = n;
    sp.P = L;
    sp.beam = 1;
    StackCfg cfg{&t.layers, t.n_layer, d, 1e-5f, CAPDEC_ACT_QUICK_GELU, true, false};
    CAPDEC_TRY(stack_body(c, cfg, sp, kv));
    float *rows = c->xl.as<float>(), *rows_ln = c->xl.as<float>() + (size_t)n * d;
    {
        ProfScope ps(c, F_EMBED);
        CAPDEC_TRY(launch_eot_index(c->stream, tokens, c->t_idx.as<int>(), n, L));
        CAPDEC_TRY(launch_gather_rows(c->stream, c->h.as<float>(), c->t_idx.as<int>(), rows, n, d));
    }
    { ProfScope ps(c, F_LN); CAPDEC_TRY(launch_layernorm(c->stream, rows, d, t.lnf_w, t.lnf_b, 1e-5f, rows_ln, d, n, d)); }
    return gemm(c, rows_ln, d, t.proj_t, d, out, t.embed, n, t.embed, d, nullptr, CAPDEC_ACT_NONE);
}

// one chunk of images through the vision tower: pixels [n, 3, S, S] -> out [n, embed]
static int clip_vision_chunk(capdec_ctx *c, const float *pixels, int n, float *out) {
    Tower &t = c->clip_vision;
    const int d = t.d, L = t.ntok, np = t.ntok - 1, kdim = 3 * t.patch * t.patch;
    KvCache kv;
    kv_geometry(kv, n, L, t.n_head, d / t.n_head);
    CAPDEC_TRY(ensure_body_ws(c, n * L, d));
    CAPDEC_TRY(c->t_patch.ensure((size_t)n * np * kdim * 4));
    CAPDEC_TRY(c->t_pout.ensure((size_t)n * np * d * 4));
    CAPDEC_TRY(c->xl.ensure((size_t)n * d * 4));
    { ProfScope ps(c, F_EMBED); CAPDEC_TRY(launch_im2col_patches(c->stream, pixels, c->t_patch.as<float>(), n, t.image, t.patch)); }
    CAPDEC_TRY(gemm(c, c->t_patch.as<float>(), kdim, t.conv_w, kdim, c->t_pout.as<float>(), d, n * np, d, kdim, nullptr,
                    CAPDEC_ACT_NONE));
    // ln_pre runs in place on the assembled sequence (x holds the pre-LN copy)
    { ProfScope ps(c, F_EMBED); CAPDEC_TRY(launch_vision_assemble(c->stream, c->t_pout.as<float>(), t.cls, t.pos_emb, c->x.as<float>(), n, L, d)); }
    { ProfScope ps(c, F_LN); CAPDEC_TRY(launch_layernorm(c->stream, c->x.as<float>(), d, t.ln_pre_w, t.ln_pre_b, 1e-5f, c->h.as<float>(), d, n * L, d)); }
    StepShape sp{};
    sp.prefill = true;
    sp.ncap = n;
    sp.P = L;
    sp.beam = 1;
    StackCfg cfg{&t.layers, t.n_layer, d, 1e-5f, CAPDEC_ACT_QUICK_GELU, false, false};
    CAPDEC_TRY(stack_body(c, cfg, sp, kv));
    // ln_post on the class token (row 0 of every sequence: row stride L*d)
    { ProfScope ps(c, F_LN); CAPDEC_TRY(launch_layernorm(c->stream, c->h.as<float>(), L * d, t.lnf_w, t.lnf_b, 1e-5f, c->xl.as<float>(), d, n, d)); }
    return gemm(c, c->xl.as<float>(), d, t.proj_t, d, out, t.embed, n, t.embed, d, nullptr, CAPDEC_ACT_NONE);
}

}  // namespace capdec

// ---------------------------------------------------------------------------- CLIP ModifiedResNet tower
static int pad64(int c) { return (c + 63) / 64 * 64; }

// fold BatchNorm (inference) into the convolution, reorder to [cout_p][(ky, kx, c_p)], pad, upload
static int upload_conv_bn(ResNet &r, const capdec_conv_bn &s, bool first, ConvW *out) {
    CAPDEC_CHECK(s.w && s.bn_w && s.bn_b && s.bn_mean && s.bn_var, "load_clip_resnet: null convolution tensor");
    CAPDEC_CHECK((s.k == 1 || s.k == 3) && s.cin >= 1 && s.cout >= 1, "load_clip_resnet: 1x1 or 3x3 convolutions only");
    ConvW c;
    c.cin = s.cin; c.cout = s.cout; c.k = s.k;
    c.cin_p = first ? s.cin : pad64(s.cin);
    c.cout_p = pad64(s.cout);
    c.K = first ? 64 : s.k * s.k * c.cin_p;
    CAPDEC_CHECK(!first || (s.cin == 3 && s.k == 3), "load_clip_resnet: the first convolution is 3x3 on 3 channels");
    std::vector<float> w((size_t)c.cout_p * c.K, 0.f), b((size_t)c.cout_p, 0.f);
    for (int o = 0; o < s.cout; ++o) {
        const float scale = s.bn_w[o] / std::sqrt(s.bn_var[o] + 1e-5f);
        b[o] = s.bn_b[o] - s.bn_mean[o] * scale;
        for (int ci = 0; ci < s.cin; ++ci)
            for (int t = 0; t < s.k * s.k; ++t)
                w[(size_t)o * c.K + (size_t)t * c.cin_p + ci] = s.w[((size_t)o * s.cin + ci) * s.k * s.k + t] * scale;
    }
    CAPDEC_TRY(upload(r.owned, w.data(), w.size(), &c.w));
    CAPDEC_TRY(upload(r.owned, b.data(), b.size(), &c.b));
    *out = c;
    return 0;
}

// out[N, Ho, Wo, cout_p] = act(conv(in) folded-BN (+ resid)); k = 3: im2col + GEMM; returns the output spatial size
static int conv_bn_forward(capdec_ctx *c, const ConvW &w, const float *in, int N, int H, int W, int stride, bool nchw3,
                           float *out, int act, const float *resid, int *Ho_, int *Wo_) {
    int Ho = H, Wo = W;
    if (w.k == 3) {
        Ho = (H + 2 - 3) / stride + 1;
        Wo = (W + 2 - 3) / stride + 1;
    }
    const int M = N * Ho * Wo;
    static const bool fused = [] { const char *e = getenv("CAPDEC_RN_PACKED"); return !(e && atoi(e) == 0); }();
    if (fused && c->gemm_mode != GEMM_F32) {
        // the A operand goes straight into the packed planes of the mode (two fp16 planes by default, one 16-bit plane
        // under clip.load(..., precision="fp16" | "bf16")): im2col for 3x3, a plain packing pass for 1x1
        const int fmt = pack_fmt(c);
        CAPDEC_TRY(c->r_col.ensure(x3_packed_bytes(M, w.K, fmt)));
        {
            ProfScope ps(c, F_PACK);
            if (w.k == 3) CAPDEC_TRY(launch_im2col3x3_packed(c->stream, in, c->r_col.p, N, H, W, w.cin_p, stride, nchw3, w.K, fmt));
            else CAPDEC_TRY(pack_any(c, in, M, w.K, fmt, c->r_col.p));
        }
        CAPDEC_TRY(gemm_packed(c, c->r_col.p, w.w, out, w.cout_p, M, w.cout_p, w.K, w.b, act, resid, w.cout_p));
    } else {
        const float *A = in;
        if (w.k == 3) {
            CAPDEC_TRY(c->r_col.ensure((size_t)M * w.K * 4));
            { ProfScope ps(c, F_OTHER); CAPDEC_TRY(launch_im2col3x3(c->stream, in, c->r_col.as<float>(), N, H, W, w.cin_p, stride, nchw3, w.K)); }
            A = c->r_col.as<float>();
        }
        CAPDEC_TRY(gemm(c, A, w.K, w.w, w.K, out, w.cout_p, M, w.cout_p, w.K, w.b, act, resid, w.cout_p));
    }
    if (Ho_) *Ho_ = Ho;
    if (Wo_) *Wo_ = Wo;
    return 0;
}

// stem + the four stages with fp32 NHWC activations between the convolutions (every convolution packs / im2cols its own
// operand): the path of the bf16x3 / f32 modes.  On return x holds the [n, H, W, feat] features.
static int resnet_body_fp32(capdec_ctx *c, const float *pixels, int n, float *&x, float *&y, float *t1, float *t2, float *xi,
                            float *idb, int *Hp, int *Wp) {
    ResNet &r = c->clip_resnet;
    int H = *Hp, W = *Wp;
    // stem: conv3x3 stride 2 (from NCHW pixels), two conv3x3, AvgPool2d(2)
    CAPDEC_TRY(conv_bn_forward(c, r.stem[0], pixels, n, H, W, 2, true, t1, CAPDEC_ACT_RELU, nullptr, &H, &W));
    CAPDEC_TRY(conv_bn_forward(c, r.stem[1], t1, n, H, W, 1, false, t2, CAPDEC_ACT_RELU, nullptr, &H, &W));
    CAPDEC_TRY(conv_bn_forward(c, r.stem[2], t2, n, H, W, 1, false, t1, CAPDEC_ACT_RELU, nullptr, &H, &W));
    { ProfScope ps(c, F_OTHER); CAPDEC_TRY(launch_avgpool2(c->stream, t1, x, n, H, W, r.stem[2].cout_p)); }
    H /= 2; W /= 2;
    size_t bi = 0;
    for (int li = 0; li < 4; ++li) {
        for (int b = 0; b < r.layers[li]; ++b, bi += 4) {
            const ConvW &c1 = r.blocks[bi], &c2 = r.blocks[bi + 1], &c3 = r.blocks[bi + 2], &ds = r.blocks[bi + 3];
            const int stride = (b == 0 && li > 0) ? 2 : 1;
            CAPDEC_TRY(conv_bn_forward(c, c1, x, n, H, W, 1, false, t1, CAPDEC_ACT_RELU, nullptr, nullptr, nullptr));
            CAPDEC_TRY(conv_bn_forward(c, c2, t1, n, H, W, 1, false, t2, CAPDEC_ACT_RELU, nullptr, nullptr, nullptr));
            int Ho = H, Wo = W;
            const float *branch = t2, *idt = x;
            if (stride > 1) {      // anti-aliased stride: an average pool on the branch and in front of the downsample conv
                ProfScope ps(c, F_OTHER);
                CAPDEC_TRY(launch_avgpool2(c->stream, t2, t1, n, H, W, c2.cout_p));
                CAPDEC_TRY(launch_avgpool2(c->stream, x, xi, n, H, W, c1.cin_p));
                branch = t1;
                Ho = H / 2; Wo = W / 2;
            }
            if (ds.w) {
                CAPDEC_TRY(conv_bn_forward(c, ds, stride > 1 ? xi : x, n, Ho, Wo, 1, false, idb, CAPDEC_ACT_NONE, nullptr,
                                           nullptr, nullptr));
                idt = idb;
            }
            CAPDEC_TRY(conv_bn_forward(c, c3, branch, n, Ho, Wo, 1, false, y, CAPDEC_ACT_RESID_RELU, idt, nullptr, nullptr));
            std::swap(x, y);
            H = Ho; W = Wo;
        }
    }
    *Hp = H; *Wp = W;
    return 0;
}

// The same with PACKED activations wherever the consumer is a GEMM operand (modes f16x2 / f16 / bf16):
//  * a 1x1 convolution whose result feeds a 3x3 one, and a 3x3 one that feeds a 1x1 one, write their result straight as
//    the packed operand of the consumer (GEMM epilogue packed_out: bias + ReLU + split, no fp32 copy in HBM);
//  * a 3x3 convolution (stride 1, padding 1 -- all of them but the very first) is an IMPLICIT GEMM over that packed
//    activation (launch_conv3x3_packed): no im2col matrix exists;
//  * fp32 NHWC remains where it is needed as such: the residual stream (identity / residual add), the inputs of the
//    average pools, the first convolution's pixels.
static int conv3x3_implicit(capdec_ctx *c, const ConvW &w, const void *in_pk, int n, int H, int W, float *out,
                            void *packed_out, int act) {
    const int fmt = pack_fmt(c);
    const void *pl = nullptr;
    CAPDEC_TRY(planes_of(c, w.w, w.cout_p, w.K, true, &pl));
    const size_t zb = (size_t)(w.cin_p / 16 + 2) * 8192;
    if (c->r_zero.cap < zb) {
        CAPDEC_TRY(c->r_zero.ensure(std::max<size_t>(zb, (size_t)512 << 10)));
        CAPDEC_HIP(hipMemsetAsync(c->r_zero.p, 0, c->r_zero.cap, c->stream));
    }
    GemmEpilogue e;
    e.bias = w.b;
    e.act = act;
    e.packed_out = packed_out;
    ProfScope ps(c, mode_single(c) ? F_GEMM_BF16P : F_GEMM_H2P, 2.0 * n * H * W * (double)w.cout_p * w.K);
    return launch_conv3x3_packed(c->stream, in_pk, pl, out, w.cout_p, n, H, W, w.cin_p, w.cout_p, e, fmt, c->r_zero.p,
                                 c->r_zero.cap);
}
// In this path NO fp32 activation exists between the pixels and the attention pool: every convolution writes its result as
// the packed operand of its consumer, the residual stream included -- the last convolution of a bottleneck adds the packed
// identity in its epilogue (GemmEpilogue::resid_packed) -- and the average pools run packed -> packed.
static int resnet_body_packed(capdec_ctx *c, const float *pixels, int n, float *feat, int *Hp, int *Wp) {
    ResNet &r = c->clip_resnet;
    const int fmt = pack_fmt(c);
    int H = *Hp, W = *Wp;
    DBuf *xp = &c->r_xpk, *yp = &c->r_ypk;
    auto gemm1x1 = [&](const ConvW &w, const void *a, int M, void *dst, int act, const void *resid_pk) {
        return gemm_packed(c, a, w.w, nullptr, w.cout_p, M, w.cout_p, w.K, w.b, act, nullptr, 0, dst, nullptr, resid_pk);
    };
    {   // stem: conv1 (stride 2, 3 input channels: im2col of K = 27 -> 64 straight into the operand), conv2, conv3, pool
        const ConvW &s0 = r.stem[0], &s1 = r.stem[1], &s2 = r.stem[2];
        const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, M = n * Ho * Wo;
        CAPDEC_TRY(c->r_col.ensure(x3_packed_bytes(M, s0.K, fmt)));
        { ProfScope ps(c, F_PACK); CAPDEC_TRY(launch_im2col3x3_packed(c->stream, pixels, c->r_col.p, n, H, W, 3, 2, true, s0.K, fmt)); }
        CAPDEC_TRY(c->r_pk1.ensure(x3_packed_bytes(M, std::max(s0.cout_p, s2.cout_p), fmt)));
        CAPDEC_TRY(c->r_pk2.ensure(x3_packed_bytes(M, s1.cout_p, fmt)));
        CAPDEC_TRY(gemm1x1(s0, c->r_col.p, M, c->r_pk1.p, CAPDEC_ACT_RELU, nullptr));
        H = Ho; W = Wo;
        CAPDEC_TRY(conv3x3_implicit(c, s1, c->r_pk1.p, n, H, W, nullptr, c->r_pk2.p, CAPDEC_ACT_RELU));
        CAPDEC_TRY(conv3x3_implicit(c, s2, c->r_pk2.p, n, H, W, nullptr, c->r_pk1.p, CAPDEC_ACT_RELU));
        CAPDEC_TRY(xp->ensure(x3_packed_bytes(M / 4, s2.cout_p, fmt)));
        { ProfScope ps(c, F_OTHER); CAPDEC_TRY(launch_avgpool2_packed(c->stream, c->r_pk1.p, xp->p, n, H, W, s2.cout_p, fmt)); }
        H /= 2; W /= 2;
    }
    size_t bi = 0;
    int C = r.stem[2].cout_p;
    for (int li = 0; li < 4; ++li) {
        for (int b = 0; b < r.layers[li]; ++b, bi += 4) {
            const ConvW &c1 = r.blocks[bi], &c2 = r.blocks[bi + 1], &c3 = r.blocks[bi + 2], &ds = r.blocks[bi + 3];
            const int stride = (b == 0 && li > 0) ? 2 : 1;
            const int M = n * H * W;
            CAPDEC_CHECK(c1.cin_p == C, "clip_resnet: channel mismatch between consecutive blocks");
            CAPDEC_TRY(c->r_pk1.ensure(x3_packed_bytes(M, c1.cout_p, fmt)));
            CAPDEC_TRY(c->r_pk2.ensure(x3_packed_bytes(M, c2.cout_p, fmt)));
            CAPDEC_TRY(gemm1x1(c1, xp->p, M, c->r_pk1.p, CAPDEC_ACT_RELU, nullptr));
            CAPDEC_TRY(conv3x3_implicit(c, c2, c->r_pk1.p, n, H, W, nullptr, c->r_pk2.p, CAPDEC_ACT_RELU));
            int Ho = H, Wo = W;
            const void *branch = c->r_pk2.p, *idt = xp->p;
            if (stride > 1) {    // anti-aliased stride: average pools on the branch and in front of the downsample conv
                Ho = H / 2; Wo = W / 2;
                CAPDEC_CHECK(ds.w != nullptr, "clip_resnet: a strided block without a downsample branch");
                CAPDEC_TRY(c->r_xi.ensure(x3_packed_bytes(n * Ho * Wo, C, fmt)));
                ProfScope ps(c, F_OTHER);
                CAPDEC_TRY(launch_avgpool2_packed(c->stream, c->r_pk2.p, c->r_pk1.p, n, H, W, c2.cout_p, fmt));
                CAPDEC_TRY(launch_avgpool2_packed(c->stream, xp->p, c->r_xi.p, n, H, W, C, fmt));
                branch = c->r_pk1.p;
            }
            const int Mo = n * Ho * Wo;
            if (ds.w) {
                CAPDEC_TRY(c->r_idp.ensure(x3_packed_bytes(Mo, ds.cout_p, fmt)));
                CAPDEC_TRY(gemm1x1(ds, stride > 1 ? c->r_xi.p : xp->p, Mo, c->r_idp.p, CAPDEC_ACT_NONE, nullptr));
                idt = c->r_idp.p;
            }
            CAPDEC_TRY(yp->ensure(x3_packed_bytes(Mo, c3.cout_p, fmt)));
            CAPDEC_TRY(gemm1x1(c3, branch, Mo, yp->p, CAPDEC_ACT_RESID_RELU, idt));
            std::swap(xp, yp);
            C = c3.cout_p;
            H = Ho; W = Wo;
        }
    }
    { ProfScope ps(c, F_OTHER); CAPDEC_TRY(launch_unpack_rows(c->stream, xp->p, feat, n * H * W, C, fmt)); }
    *Hp = H; *Wp = W;
    return 0;
}

// one chunk of images: pixels [n, 3, S, S] (NCHW) -> out [n, embed]
static int clip_resnet_chunk(capdec_ctx *c, const float *pixels, int n, float *out) {
    ResNet &r = c->clip_resnet;
    const int S = r.image;
    static const bool implicit_on = [] { const char *e = getenv("CAPDEC_RN_IMPLICIT"); return !(e && atoi(e) == 0); }();
    const bool packed_path = implicit_on && (c->gemm_mode == GEMM_F16X2 || mode_single(c));
    // fp32 buffers (floats per image): the attention pool's tokens / keys / values always; in the fp32-activation path
    // also the worst-case activation: stem conv outputs at S/2, stage outputs at S/4 ... S/32
    size_t act = ((size_t)r.sp * r.sp + 1) * r.feat;
    if (!packed_path) {
        const size_t half = (size_t)(S / 2) * (S / 2), quarter = (size_t)(S / 4) * (S / 4);
        act = std::max(act, half * pad64(r.width));                                // stem
        int planes = r.width, sp = S / 4;
        for (int li = 0; li < 4; ++li, planes *= 2) {
            const int spin = sp;                                                   // spatial size entering the stage
            if (li > 0) sp /= 2;
            act = std::max(act, (size_t)spin * spin * pad64(planes * 4));          // identity / stage output
            act = std::max(act, (size_t)spin * spin * pad64(planes));              // conv1 / conv2 outputs before the pool
            act = std::max(act, (size_t)spin * spin * pad64(li ? planes * 2 : planes));   // stage input
        }
        act = std::max(act, quarter * pad64(r.width));
    }
    for (DBuf *b : {&c->r_a, &c->r_b, &c->r_c, &c->r_d, &c->r_e, &c->r_f}) CAPDEC_TRY(b->ensure((size_t)n * act * 4));
    float *x = c->r_a.as<float>(), *y = c->r_b.as<float>(), *t1 = c->r_c.as<float>(), *t2 = c->r_d.as<float>(),
          *xi = c->r_e.as<float>(), *idb = c->r_f.as<float>();
    int H = S, W = S;
    if (packed_path) {
        CAPDEC_TRY(resnet_body_packed(c, pixels, n, x, &H, &W));
    } else {
        CAPDEC_TRY(resnet_body_fp32(c, pixels, n, x, y, t1, t2, xi, idb, &H, &W));
    }
    // attention pool: tokens = [mean; features] + pos; one query (the mean token) over all tokens
    const int C = r.feat, HW = H * W, T = HW + 1;
    CAPDEC_CHECK(H == r.sp && W == r.sp, "clip_resnet: unexpected spatial size in front of the attention pool");
    float *tok = y, *kk = t1, *vv = t2, *qq = xi, *oo = idb;
    { ProfScope ps(c, F_OTHER); CAPDEC_TRY(launch_attnpool_tokens(c->stream, x, r.pos, tok, n, HW, C)); }
    CAPDEC_TRY(gemm(c, tok, C, r.wk, C, kk, C, n * T, C, C, r.bk, CAPDEC_ACT_NONE));
    CAPDEC_TRY(gemm(c, tok, C, r.wv, C, vv, C, n * T, C, C, r.bv, CAPDEC_ACT_NONE));
    CAPDEC_TRY(gemm(c, tok, T * C, r.wq, C, qq, C, n, C, C, r.bq, CAPDEC_ACT_NONE));          // token 0 of every image
    { ProfScope ps(c, F_MAP_ATTN); CAPDEC_TRY(launch_attnpool_attend(c->stream, qq, kk, vv, oo, n, r.heads, T, C)); }
    return gemm(c, oo, C, r.wc, C, out, r.embed, n, r.embed, C, r.bc, CAPDEC_ACT_NONE);
}

// ---- RCCL (dlopen'ed): only the five entry points the path needs
namespace {
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl *rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char *names[] = {getenv("CAPDEC_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
        }
        if (r.h) {
            r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
            r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
            r.AllGather = (decltype(r.AllGather))dlsym(r.h, "ncclAllGather");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
            if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) r.h = nullptr;
        }
    }
    return r.h ? &r : nullptr;
}
#define CAPDEC_NCCL(expr)                                                                               \
    do {                                                                                                \
        ncclResult_t _r = (expr);                                                                       \
        if (_r != ncclSuccess) {                                                                        \
            capdec::set_error(std::string(#expr) + ": " + (rccl()->GetErrorString ? rccl()->GetErrorString(_r) : "RCCL error")); \
            return 1;                                                                                   \
        }                                                                                               \
    } while (0)
}  // namespace

// =============================================================================== C ABI
extern "C" {

int capdec_abi_version(void) { return CAPDEC_ABI_VERSION; }
#ifndef CAPDEC_BUILD_ID
#define CAPDEC_BUILD_ID "unknown"
#endif
const char *capdec_build_id(void) { return CAPDEC_BUILD_ID; }
const char *capdec_last_error(void) { return g_err.c_str(); }

int capdec_create(int device_id, capdec_ctx **out) {
    CAPDEC_CHECK(out != nullptr, "create: null out pointer");
    int ndev = 0;
    CAPDEC_HIP(hipGetDeviceCount(&ndev));
    CAPDEC_CHECK(device_id >= 0 && device_id < ndev, "create: no such HIP device");
    CAPDEC_HIP(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    CAPDEC_HIP(hipGetDeviceProperties(&prop, device_id));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
        set_error(std::string("create: libcapdec_hip is built for gfx950 (MI355X) only, found ") + prop.gcnArchName);
        return 1;
    }
    std::unique_ptr<capdec_ctx> c(new capdec_ctx());
    c->device = device_id;
    if (const char *e = getenv("CAPDEC_X3_PACKA")) c->pack_a = atoi(e) != 0;
    if (const char *e = getenv("CAPDEC_X3_CHAIN")) c->pack_chain = atoi(e) != 0;
    if (const char *e = getenv("CAPDEC_COMPACT")) c->compact = atoi(e) != 0;
    if (const char *e = getenv("CAPDEC_GEMM_MODE")) {
        const std::string m(e);
        if (m != "f32" && m != "bf16" && m != "bf16x3" && m != "f16" && m != "f16x2") {     // a typo must not silently select another precision
            set_error("create: CAPDEC_GEMM_MODE=" + m + " is not one of f16x2 | bf16x3 | f32 | bf16 | f16");
            return 1;
        }
        c->gemm_mode = m == "f32" ? GEMM_F32 : m == "bf16" ? GEMM_BF16 : m == "bf16x3" ? GEMM_BF16X3 : m == "f16" ? GEMM_F16 : GEMM_F16X2;
    }
    if (const char *e = getenv("CAPDEC_BATCH_INVARIANT")) c->batch_invariant = atoi(e) != 0;
    CAPDEC_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    CAPDEC_HIP(hipEventCreate(&c->t0));
    CAPDEC_HIP(hipEventCreate(&c->t1));
    CAPDEC_HIP(hipHostMalloc((void **)&c->alive_host, sizeof(int), hipHostMallocDefault));
    *out = c.release();
    return 0;
}

void capdec_destroy(capdec_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->comm && rccl()) { (void)rccl()->CommDestroy(c->comm); c->comm = nullptr; }
    c->g_pad.release();
    c->g_all.release();
    free_all(c->gpt.owned);
    free_all(c->map.owned);
    free_all(c->clip_text.owned);
    free_all(c->clip_vision.owned);
    free_all(c->clip_resnet.owned);
    drop_planes(c);
    c->x3_tmp.release();
    DBuf *bufs[] = {&c->h, &c->x, &c->qkv, &c->att, &c->ff, &c->xl, &c->tmax, &c->tsum, &c->cval, &c->cidx,
                    &c->lse, &c->topv, &c->topi, &c->kc, &c->vc, &c->tokens, &c->scores, &c->seq, &c->stopped,
                    &c->done, &c->anc, &c->next_tok, &c->alive, &c->gids, &c->glens, &c->m_hid, &c->m_lin, &c->m_seq,
                    &c->m_x, &c->m_qkv, &c->m_att, &c->m_ff, &c->t_idx, &c->t_patch, &c->t_pout, &c->xpk, &c->apk, &c->fpk, &c->cmap, &c->kvstat, &c->p_desc, &c->p_inter, &c->splitk, &c->absmax, &c->a_tmp,
                    &c->r_a, &c->r_b, &c->r_c, &c->r_d, &c->r_e, &c->r_f, &c->r_col, &c->r_pk1, &c->r_pk2, &c->r_xpk,
                    &c->r_ypk, &c->r_xi, &c->r_idp, &c->r_zero};
    for (DBuf *b : bufs) b->release();
    for (auto &r : c->prof.recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : c->prof.pool) (void)hipEventDestroy(e);
    if (c->t0) (void)hipEventDestroy(c->t0);
    if (c->t1) (void)hipEventDestroy(c->t1);
    if (c->alive_host) (void)hipHostFree(c->alive_host);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int capdec_set_stream(capdec_ctx *c, void *hip_stream) {
    CAPDEC_CHECK(c, "null context");
    c->stream = reinterpret_cast<hipStream_t>(hip_stream);   // NULL = HIP's default stream
    return 0;
}
int capdec_use_own_stream(capdec_ctx *c) {
    CAPDEC_CHECK(c, "null context");
    c->stream = c->own_stream;
    return 0;
}
int capdec_synchronize(capdec_ctx *c) {
    CAPDEC_CHECK(c, "null context");
    CAPDEC_HIP(hipStreamSynchronize(c->stream));
    return 0;
}
int capdec_set_gemm_mode(capdec_ctx *c, int mode) {
    CAPDEC_CHECK(c && mode >= GEMM_F32 && mode <= GEMM_F16,
                 "set_gemm_mode: mode must be 0 (f32 MFMA), 1 (bf16x3, fp32-accurate), 2 (bf16 operands), 3 (f16x2, "
                 "fp32-accurate, default) or 4 (fp16 operands)");
    c->gemm_mode = mode;
    return 0;
}
int capdec_get_gemm_mode(capdec_ctx *c) { return c ? c->gemm_mode : -1; }
int capdec_set_batch_invariant(capdec_ctx *c, int on) {
    CAPDEC_CHECK(c, "null context");
    c->batch_invariant = on != 0;
    return 0;
}
int capdec_set_debug_diverge(capdec_ctx *c, int on) {
    CAPDEC_CHECK(c, "null context");
    c->diverge = on != 0;
    return 0;
}
int capdec_set_kv_budget(capdec_ctx *c, size_t bytes) {
    CAPDEC_CHECK(c, "null context");
    c->kv_budget = bytes ? bytes : ((size_t)192 << 30);
    return 0;
}
int capdec_malloc(capdec_ctx *c, size_t bytes, void **d_ptr) {
    CAPDEC_CHECK(c && d_ptr, "null argument");
    CAPDEC_HIP(hipSetDevice(c->device));
    CAPDEC_HIP(hipMalloc(d_ptr, bytes ? bytes : 1));
    return 0;
}
int capdec_free(capdec_ctx *c, void *d_ptr) {
    CAPDEC_CHECK(c, "null context");
    if (d_ptr) CAPDEC_HIP(hipFree(d_ptr));
    return 0;
}
int capdec_memcpy_h2d(capdec_ctx *c, void *d_dst, const void *h_src, size_t bytes) {
    CAPDEC_CHECK(c, "null context");
    if (!bytes) return 0;
    CAPDEC_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
    CAPDEC_HIP(hipStreamSynchronize(c->stream));
    return 0;
}
int capdec_memcpy_d2h(capdec_ctx *c, void *h_dst, const void *d_src, size_t bytes) {
    CAPDEC_CHECK(c, "null context");
    if (!bytes) return 0;
    CAPDEC_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    CAPDEC_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int capdec_load_gpt2(capdec_ctx *c, const capdec_gpt2_weights *w) {
    CAPDEC_CHECK(c && w, "null argument");
    CAPDEC_CHECK(w->n_layer >= 1 && w->n_head >= 1 && w->vocab >= 8 && w->n_pos >= 1, "load_gpt2: bad geometry");
    CAPDEC_CHECK(w->n_embd % w->n_head == 0 && w->n_embd / w->n_head == 64, "load_gpt2: head_dim must be 64");
    CAPDEC_CHECK(w->n_embd % 32 == 0 && w->n_embd <= 1024, "load_gpt2: n_embd must be a multiple of 32, <= 1024");
    CAPDEC_HIP(hipSetDevice(c->device));
    Gpt2 &g = c->gpt;
    free_all(g.owned);
    drop_planes(c);
    g = Gpt2();
    g.n_layer = w->n_layer; g.n_head = w->n_head; g.d = w->n_embd; g.vocab = w->vocab; g.n_pos = w->n_pos;
    g.eps = w->ln_eps > 0 ? w->ln_eps : 1e-5f;
    const int d = g.d;
    CAPDEC_TRY(upload(g.owned, w->wte, (size_t)g.vocab * d, &g.wte));
    CAPDEC_TRY(upload(g.owned, w->wpe, (size_t)g.n_pos * d, &g.wpe));
    CAPDEC_TRY(upload(g.owned, w->ln_f_w, d, &g.lnfw));
    CAPDEC_TRY(upload(g.owned, w->ln_f_b, d, &g.lnfb));
    g.layers.resize(g.n_layer);
    for (int l = 0; l < g.n_layer; ++l) {
        const capdec_gpt2_layer &s = w->layers[l];
        Gpt2Layer &t = g.layers[l];
        CAPDEC_TRY(upload(g.owned, s.ln_1_w, d, &t.ln1w));
        CAPDEC_TRY(upload(g.owned, s.ln_1_b, d, &t.ln1b));
        CAPDEC_TRY(upload_transposed(c, g.owned, s.c_attn_w, d, 3 * d, &t.wqkv));
        CAPDEC_TRY(upload(g.owned, s.c_attn_b, 3 * d, &t.bqkv));
        CAPDEC_TRY(upload_transposed(c, g.owned, s.c_proj_w, d, d, &t.wproj));
        CAPDEC_TRY(upload(g.owned, s.c_proj_b, d, &t.bproj));
        CAPDEC_TRY(upload(g.owned, s.ln_2_w, d, &t.ln2w));
        CAPDEC_TRY(upload(g.owned, s.ln_2_b, d, &t.ln2b));
        CAPDEC_TRY(upload_transposed(c, g.owned, s.c_fc_w, d, 4 * d, &t.wfc));
        CAPDEC_TRY(upload(g.owned, s.c_fc_b, 4 * d, &t.bfc));
        CAPDEC_TRY(upload_transposed(c, g.owned, s.mlp_c_proj_w, 4 * d, d, &t.wproj2));
        CAPDEC_TRY(upload(g.owned, s.mlp_c_proj_b, d, &t.bproj2));
    }
    g.loaded = true;
    return 0;
}

int capdec_load_mapper_mlp(capdec_ctx *c, int D, int P, int hidden, const float *w1, const float *b1, const float *w2,
                           const float *b2) {
    CAPDEC_CHECK(c, "null context");
    CAPDEC_CHECK(D % 32 == 0 && hidden % 32 == 0 && P >= 1, "load_mapper_mlp: dims must be multiples of 32");
    CAPDEC_HIP(hipSetDevice(c->device));
    Mapper &m = c->map;
    free_all(m.owned);
    drop_planes(c);
    m = Mapper();
    m.D = D; m.P = P; m.hidden = hidden;
    m.d = c->gpt.loaded ? c->gpt.d : 768;
    CAPDEC_TRY(upload(m.owned, w1, (size_t)hidden * D, &m.w1));
    CAPDEC_TRY(upload(m.owned, b1, hidden, &m.b1));
    CAPDEC_TRY(upload(m.owned, w2, (size_t)m.P * m.d * hidden, &m.w2));
    CAPDEC_TRY(upload(m.owned, b2, (size_t)m.P * m.d, &m.b2));
    m.kind = 1;
    return 0;
}

int capdec_load_mapper_transformer(capdec_ctx *c, const capdec_tmapper_weights *w) {
    CAPDEC_CHECK(c && w, "null argument");
    CAPDEC_CHECK(w->prefix_dim % 32 == 0 && w->d % 32 == 0 && w->mlp_hidden % 32 == 0, "load_mapper_transformer: dims must be multiples of 32");
    CAPDEC_CHECK(w->num_heads >= 1 && w->d % w->num_heads == 0, "load_mapper_transformer: bad head count");
    CAPDEC_CHECK(w->clip_length >= 1 && w->prefix_length >= 1 && w->num_layers >= 1, "load_mapper_transformer: bad geometry");
    CAPDEC_HIP(hipSetDevice(c->device));
    Mapper &m = c->map;
    free_all(m.owned);
    drop_planes(c);
    m = Mapper();
    m.D = w->prefix_dim; m.P = w->prefix_length; m.clip_len = w->clip_length; m.n_layers = w->num_layers;
    m.heads = w->num_heads; m.d = w->d; m.mlp_hidden = w->mlp_hidden;
    const int d = m.d;
    CAPDEC_TRY(upload(m.owned, w->linear_w, (size_t)m.clip_len * d * m.D, &m.lin_w));
    CAPDEC_TRY(upload(m.owned, w->linear_b, (size_t)m.clip_len * d, &m.lin_b));
    CAPDEC_TRY(upload(m.owned, w->prefix_const, (size_t)m.P * d, &m.prefix_const));
    m.layers.resize(m.n_layers);
    for (int l = 0; l < m.n_layers; ++l) {
        const capdec_tmapper_layer &s = w->layers[l];
        TMapLayer &t = m.layers[l];
        CAPDEC_TRY(upload(m.owned, s.norm1_w, d, &t.n1w));
        CAPDEC_TRY(upload(m.owned, s.norm1_b, d, &t.n1b));
        // fused projection [3d, d] = [to_queries ; to_keys_values] -> rows [q | k | v]
        CAPDEC_CHECK(s.to_queries_w && s.to_keys_values_w, "load_mapper_transformer: null weight");
        void *p = nullptr;
        CAPDEC_HIP(hipMalloc(&p, (size_t)3 * d * d * 4));
        m.owned.push_back(p);
        t.wqkv = (float *)p;
        CAPDEC_HIP(hipMemcpy(t.wqkv, s.to_queries_w, (size_t)d * d * 4, hipMemcpyHostToDevice));
        CAPDEC_HIP(hipMemcpy(t.wqkv + (size_t)d * d, s.to_keys_values_w, (size_t)2 * d * d * 4, hipMemcpyHostToDevice));
        CAPDEC_TRY(upload(m.owned, s.project_w, (size_t)d * d, &t.wproj));
        CAPDEC_TRY(upload(m.owned, s.project_b, d, &t.bproj));
        CAPDEC_TRY(upload(m.owned, s.norm2_w, d, &t.n2w));
        CAPDEC_TRY(upload(m.owned, s.norm2_b, d, &t.n2b));
        CAPDEC_TRY(upload(m.owned, s.fc1_w, (size_t)m.mlp_hidden * d, &t.wfc1));
        CAPDEC_TRY(upload(m.owned, s.fc1_b, m.mlp_hidden, &t.bfc1));
        CAPDEC_TRY(upload(m.owned, s.fc2_w, (size_t)d * m.mlp_hidden, &t.wfc2));
        CAPDEC_TRY(upload(m.owned, s.fc2_b, d, &t.bfc2));
    }
    m.kind = 2;
    return 0;
}

int capdec_load_clip_text(capdec_ctx *c, const capdec_clip_text_weights *w) {
    CAPDEC_CHECK(c && w, "null argument");
    CAPDEC_CHECK(w->width % 32 == 0 && w->heads >= 1 && w->width / w->heads == 64 && w->width % w->heads == 0,
                 "load_clip_text: head_dim must be 64");
    CAPDEC_CHECK(w->context_length >= 1 && w->context_length <= 256 && w->layers >= 1 && w->embed_dim >= 1 && w->vocab >= 2,
                 "load_clip_text: bad geometry");
    CAPDEC_HIP(hipSetDevice(c->device));
    Tower &t = c->clip_text;
    free_all(t.owned);
    drop_planes(c);
    t = Tower();
    t.n_layer = w->layers; t.n_head = w->heads; t.d = w->width; t.embed = w->embed_dim; t.ctx = w->context_length;
    t.vocab = w->vocab;
    CAPDEC_TRY(upload(t.owned, w->token_embedding, (size_t)t.vocab * t.d, &t.tok_emb));
    CAPDEC_TRY(upload(t.owned, w->positional_embedding, (size_t)t.ctx * t.d, &t.pos_emb));
    CAPDEC_TRY(upload(t.owned, w->ln_final_w, t.d, &t.lnf_w));
    CAPDEC_TRY(upload(t.owned, w->ln_final_b, t.d, &t.lnf_b));
    CAPDEC_TRY(upload_transposed(c, t.owned, w->text_projection, t.d, t.embed, &t.proj_t));
    CAPDEC_TRY(upload_blocks(c, t, w->blocks));
    t.loaded = true;
    return 0;
}

int capdec_load_clip_vision(capdec_ctx *c, const capdec_clip_vision_weights *w) {
    CAPDEC_CHECK(c && w, "null argument");
    CAPDEC_CHECK(w->width % 32 == 0 && w->heads >= 1 && w->width % w->heads == 0 && w->width / w->heads == 64,
                 "load_clip_vision: head_dim must be 64");
    CAPDEC_CHECK(w->patch >= 4 && w->patch % 4 == 0 && w->image_size % w->patch == 0 && (3 * w->patch * w->patch) % 32 == 0,
                 "load_clip_vision: bad patch geometry");
    CAPDEC_HIP(hipSetDevice(c->device));
    Tower &t = c->clip_vision;
    free_all(t.owned);
    free_all(c->clip_resnet.owned);                 // one image tower at a time
    c->clip_resnet = ResNet();
    drop_planes(c);
    t = Tower();
    t.n_layer = w->layers; t.n_head = w->heads; t.d = w->width; t.embed = w->embed_dim; t.image = w->image_size;
    t.patch = w->patch;
    const int g = t.image / t.patch;
    t.ntok = g * g + 1;
    CAPDEC_CHECK(t.ntok <= 256, "load_clip_vision: more than 256 tokens per image");
    CAPDEC_TRY(upload(t.owned, w->conv1_w, (size_t)t.d * 3 * t.patch * t.patch, &t.conv_w));
    CAPDEC_TRY(upload(t.owned, w->class_embedding, t.d, &t.cls));
    CAPDEC_TRY(upload(t.owned, w->positional_embedding, (size_t)t.ntok * t.d, &t.pos_emb));
    CAPDEC_TRY(upload(t.owned, w->ln_pre_w, t.d, &t.ln_pre_w));
    CAPDEC_TRY(upload(t.owned, w->ln_pre_b, t.d, &t.ln_pre_b));
    CAPDEC_TRY(upload(t.owned, w->ln_post_w, t.d, &t.lnf_w));
    CAPDEC_TRY(upload(t.owned, w->ln_post_b, t.d, &t.lnf_b));
    CAPDEC_TRY(upload_transposed(c, t.owned, w->proj, t.d, t.embed, &t.proj_t));
    CAPDEC_TRY(upload_blocks(c, t, w->blocks));
    t.loaded = true;
    return 0;
}

int capdec_load_clip_resnet(capdec_ctx *c, const capdec_clip_resnet_weights *w) {
    CAPDEC_CHECK(c && w && w->stem && w->blocks, "load_clip_resnet: null argument");
    CAPDEC_CHECK(w->width >= 2 && w->width % 2 == 0 && (w->width * 32) % 64 == 0 && w->embed_dim >= 1 &&
                     w->image_size >= 64 && w->image_size % 32 == 0,
                 "load_clip_resnet: bad geometry (width even, 32 * width a multiple of 64, image_size a multiple of 32)");
    CAPDEC_CHECK((w->image_size / 32) * (w->image_size / 32) + 1 <= 256, "load_clip_resnet: more than 256 attention-pool tokens");
    CAPDEC_HIP(hipSetDevice(c->device));
    ResNet &r = c->clip_resnet;
    free_all(r.owned);
    drop_planes(c);
    r = ResNet();
    free_all(c->clip_vision.owned);                 // one image tower at a time
    c->clip_vision = Tower();
    r.image = w->image_size; r.width = w->width; r.embed = w->embed_dim; r.feat = w->width * 32; r.heads = r.feat / 64;
    r.sp = w->image_size / 32;
    int nblocks = 0;
    for (int i = 0; i < 4; ++i) {
        CAPDEC_CHECK(w->layers[i] >= 1, "load_clip_resnet: every stage needs at least one block");
        r.layers[i] = w->layers[i];
        nblocks += w->layers[i];
    }
    for (int i = 0; i < 3; ++i) CAPDEC_TRY(upload_conv_bn(r, w->stem[i], i == 0, &r.stem[i]));
    r.blocks.resize((size_t)4 * nblocks);
    for (int i = 0; i < 4 * nblocks; ++i) {
        if (i % 4 == 3 && w->blocks[i].w == nullptr) continue;          // no downsample in this block
        CAPDEC_TRY(upload_conv_bn(r, w->blocks[i], false, &r.blocks[(size_t)i]));
    }
    const size_t C = (size_t)r.feat, T = (size_t)r.sp * r.sp + 1;
    CAPDEC_CHECK(r.feat % 64 == 0 && r.embed % 32 == 0, "load_clip_resnet: feature / embedding widths must be multiples of 64 / 32");
    CAPDEC_TRY(upload(r.owned, w->positional_embedding, T * C, &r.pos));
    CAPDEC_TRY(upload(r.owned, w->q_w, C * C, &r.wq)); CAPDEC_TRY(upload(r.owned, w->q_b, C, &r.bq));
    CAPDEC_TRY(upload(r.owned, w->k_w, C * C, &r.wk)); CAPDEC_TRY(upload(r.owned, w->k_b, C, &r.bk));
    CAPDEC_TRY(upload(r.owned, w->v_w, C * C, &r.wv)); CAPDEC_TRY(upload(r.owned, w->v_b, C, &r.bv));
    CAPDEC_TRY(upload(r.owned, w->c_w, (size_t)r.embed * C, &r.wc)); CAPDEC_TRY(upload(r.owned, w->c_b, (size_t)r.embed, &r.bc));
    r.loaded = true;
    return 0;
}

int capdec_clip_encode_text(capdec_ctx *c, const int32_t *tokens, int n, float *out) {
    CAPDEC_CHECK(c && c->clip_text.loaded, "clip_encode_text: text tower not loaded");
    CAPDEC_CHECK(n >= 0 && (n == 0 || (tokens && out)), "clip_encode_text: bad argument");
    CAPDEC_HIP(hipSetDevice(c->device));
    const Tower &t = c->clip_text;
    const int chunk = 4096;
    for (int c0 = 0; c0 < n; c0 += chunk) {
        const int nc = std::min(chunk, n - c0);
        CAPDEC_TRY(clip_text_chunk(c, tokens + (size_t)c0 * t.ctx, nc, out + (size_t)c0 * t.embed));
    }
    return 0;
}

int capdec_clip_encode_image(capdec_ctx *c, const float *pixels, int n, float *out) {
    CAPDEC_CHECK(c && (c->clip_vision.loaded || c->clip_resnet.loaded), "clip_encode_image: vision tower not loaded");
    CAPDEC_CHECK(n >= 0 && (n == 0 || (pixels && out)), "clip_encode_image: bad argument");
    CAPDEC_HIP(hipSetDevice(c->device));
    if (c->clip_resnet.loaded) {
        const ResNet &r = c->clip_resnet;
        // images per chunk: the late stages have few pixels per image (9 x 9 at the end), so their GEMMs only fill the
        // chip with ~100 images in flight; the largest temporary is the im2col operand of the stem (S/2 x S/2 pixels x
        // 9 x 64 values x 4 B): up to 8 GB of it (a 288 GB part), less when the device is short of free memory
        const size_t per_img = (size_t)(r.image / 2) * (r.image / 2) * 9 * pad64(r.width / 2) * 4;
        size_t free_b = 0, total_b = 0;
        CAPDEC_HIP(hipMemGetInfo(&free_b, &total_b));
        const size_t have = c->r_col.cap + c->a_tmp.cap + c->r_a.cap * 6;         // what this path already holds
        const size_t budget = std::min<size_t>((size_t)8 << 30, (free_b + have) / 4);
        const int chunk = (int)std::max<size_t>(1, budget / std::max<size_t>(per_img, 1));
        for (int c0 = 0; c0 < n; c0 += chunk) {
            const int nc = std::min(chunk, n - c0);
            CAPDEC_TRY(clip_resnet_chunk(c, pixels + (size_t)c0 * 3 * r.image * r.image, nc, out + (size_t)c0 * r.embed));
        }
        return 0;
    }
    const Tower &t = c->clip_vision;
    const int chunk = 2048;
    for (int c0 = 0; c0 < n; c0 += chunk) {
        const int nc = std::min(chunk, n - c0);
        CAPDEC_TRY(clip_vision_chunk(c, pixels + (size_t)c0 * 3 * t.image * t.image, nc, out + (size_t)c0 * t.embed));
    }
    return 0;
}

int capdec_normalize_prefix(capdec_ctx *c, const float *x, int n, int dim, int normalize, const float *offset,
                            float *out) {
    CAPDEC_CHECK(c && n >= 0 && dim >= 1 && (n == 0 || (x && out)), "normalize_prefix: bad argument");
    if (n == 0) return 0;
    CAPDEC_HIP(hipSetDevice(c->device));
    ProfScope ps(c, F_OTHER);
    return launch_normalize_prefix(c->stream, x, n, dim, normalize, offset, out);
}

int capdec_noise_inject(capdec_ctx *c, const float *x, int n, int dim, float variance, const float *offset,
                        int uniform, int dont_norm, uint64_t seed, const float *noise, const float *u, float *out) {
    CAPDEC_CHECK(c && n >= 0 && dim >= 1 && (n == 0 || (x && out)), "noise_inject: bad argument");
    CAPDEC_CHECK(variance >= 0.f, "noise_inject: negative variance");
    if (n == 0) return 0;
    CAPDEC_HIP(hipSetDevice(c->device));
    ProfScope ps(c, F_OTHER);
    return launch_noise_inject(c->stream, x, n, dim, variance, offset, uniform, dont_norm, seed, noise, u, out);
}

int capdec_mapper_forward(capdec_ctx *c, const float *x, int n, float *out) {
    CAPDEC_CHECK(c && c->map.kind != 0, "mapper_forward: no mapper loaded");
    CAPDEC_CHECK(n >= 0 && (n == 0 || (x && out)), "mapper_forward: bad argument");
    CAPDEC_HIP(hipSetDevice(c->device));
    const Mapper &m = c->map;
    const int chunk = 8192;
    for (int c0 = 0; c0 < n; c0 += chunk) {
        const int nc = std::min(chunk, n - c0);
        CAPDEC_TRY(mapper_chunk(c, x + (size_t)c0 * m.D, nc, out + (size_t)c0 * m.P * m.d));
    }
    return 0;
}

int capdec_gpt2_logits(capdec_ctx *c, const float *embeds, int n, int L, int all_positions, float *logits) {
    CAPDEC_CHECK(c && c->gpt.loaded, "gpt2_logits: GPT-2 weights not loaded");
    CAPDEC_CHECK(embeds && logits && n >= 1 && L >= 1 && L <= 256 && L <= c->gpt.n_pos, "gpt2_logits: bad argument");
    CAPDEC_HIP(hipSetDevice(c->device));
    const Gpt2 &g = c->gpt;
    const int d = g.d;
    KvCache kv;
    CAPDEC_TRY(ensure_kv(c, kv, n, L));
    CAPDEC_TRY(ensure_body_ws(c, n * L, d));
    { ProfScope ps(c, F_EMBED); CAPDEC_TRY(launch_embed_prefix(c->stream, embeds, g.wpe, c->h.as<float>(), n, L, 0, d)); }
    StepShape sp{};
    sp.prefill = true;
    sp.ncap = n;
    sp.P = L;
    sp.beam = 1;
    CAPDEC_TRY(gpt2_body(c, sp, kv));
    const int R = all_positions ? n * L : n;
    CAPDEC_TRY(c->xl.ensure((size_t)R * d * 4));
    const float *h0 = all_positions ? c->h.as<float>() : c->h.as<float>() + (size_t)(L - 1) * d;
    const int ldh = all_positions ? d : L * d;
    if (use_packed_a(c, d))   // same operand path as the decode loop's fused lm_head (bf16 mode: bf16 operands)
        return ln_gemm_packed(c, h0, ldh, g.lnfw, g.lnfb, g.eps, g.wte, logits, g.vocab, R, g.vocab, d, nullptr,
                              CAPDEC_ACT_NONE);
    { ProfScope ps(c, F_LN); CAPDEC_TRY(launch_layernorm(c->stream, h0, ldh, g.lnfw, g.lnfb, g.eps, c->xl.as<float>(), d, R, d)); }
    CAPDEC_TRY(gemm(c, c->xl.as<float>(), d, g.wte, d, logits, g.vocab, R, g.vocab, d, nullptr, CAPDEC_ACT_NONE));
    return 0;
}

int capdec_cross_entropy(capdec_ctx *c, const float *logits, int ld, const int32_t *labels, int rows, int vocab,
                         int ignore_index, float *loss) {
    CAPDEC_CHECK(c && loss && (rows == 0 || (logits && labels)), "cross_entropy: null argument");
    CAPDEC_CHECK(rows >= 0 && vocab > 0 && ld >= vocab, "cross_entropy: bad sizes");
    CAPDEC_HIP(hipSetDevice(c->device));
    if (rows == 0) return 0;
    CAPDEC_TRY(c->xl.ensure((size_t)rows * sizeof(float)));
    ProfScope ps(c, F_SELECT);
    return launch_cross_entropy_mean(c->stream, logits, ld, labels, rows, vocab, ignore_index, c->xl.as<float>(), loss);
}

int capdec_wte_lookup(capdec_ctx *c, const int32_t *ids, int n, float *out) {
    CAPDEC_CHECK(c && c->gpt.loaded, "wte_lookup: GPT-2 weights not loaded");
    CAPDEC_HIP(hipSetDevice(c->device));
    ProfScope ps(c, F_EMBED);
    return launch_gather_rows(c->stream, c->gpt.wte, ids, out, n, c->gpt.d);
}

int capdec_decode_greedy(capdec_ctx *c, const float *prefix, int n, int P, int stop_id, int alt_stop_id,
                         int entry_length, int32_t *ids, int32_t *lens) {
    CAPDEC_CHECK(c && (n == 0 || (prefix && ids && lens)), "decode_greedy: null argument");
    return decode_common(c, prefix, n, P, 1, true, stop_id, alt_stop_id, entry_length, 1.0f, ids, lens, nullptr,
                         nullptr);
}

int capdec_decode_greedy_forced(capdec_ctx *c, const float *prefix, int n, int P, int entry_length,
                                const int32_t *forced, int32_t *ids, float *stats) {
    CAPDEC_CHECK(c && (n == 0 || (prefix && forced && ids)), "decode_greedy_forced: null argument");
    DBuf lens;
    CAPDEC_TRY(lens.ensure((size_t)std::max(n, 1) * 4));
    const bool compact = c->compact;
    c->compact = false;                       // every caption runs every step
    const int rc = decode_common(c, prefix, n, P, 1, true, -1, -1, entry_length, 1.0f, ids, lens.as<int>(), nullptr, nullptr,
                                 forced, stats);
    c->compact = compact;
    lens.release();
    return rc;
}

int capdec_decode_beam(capdec_ctx *c, const float *prefix, int n, int P, int beam, int stop_id, int entry_length,
                       float temperature, int32_t *ids, int32_t *lens, float *scores, int32_t *order) {
    CAPDEC_CHECK(c && (n == 0 || (prefix && ids && lens && scores)), "decode_beam: null argument");
    CAPDEC_CHECK(c->gpt.loaded && c->gpt.vocab >= beam, "decode_beam: vocabulary smaller than the beam");
    return decode_common(c, prefix, n, P, beam, false, stop_id, -1, entry_length, temperature, ids, lens, scores,
                         order);
}

int capdec_gemm_f32(capdec_ctx *c, const float *a, int lda, const float *bt, int ldb, float *cc, int ldc, int M, int N,
                    int K, const float *bias, const float *resid, int ldr, int act) {
    CAPDEC_CHECK(c && a && bt && cc, "gemm: null argument");
    CAPDEC_HIP(hipSetDevice(c->device));
    const bool cache = getenv("CAPDEC_HOOK_CACHE") != nullptr;   // benchmarking: treat Bt as a resident weight
    const bool packa = getenv("CAPDEC_HOOK_PACKA") != nullptr;   // tests / benchmarking: pre-packed A (the LayerNorm -> GEMM path)
    if ((packa || mode_single(c)) && c->gemm_mode != GEMM_F32 && lda == K && ldb == K && K % 64 == 0) {
        const void *pa = nullptr, *pb = nullptr;
        if (cache) {
            CAPDEC_TRY(planes_of(c, a, M, K, true, &pa));
        } else {   // tests: always re-pack A (the plane cache is keyed by address, torch recycles addresses)
            CAPDEC_TRY(c->xpk.ensure(x3_packed_bytes_host(M, K)));
            CAPDEC_TRY(pack_any(c, a, M, K, pack_fmt(c), c->xpk.p));
            pa = c->xpk.p;
        }
        GemmEpilogue e;
        CAPDEC_TRY(planes_of(c, bt, N, K, cache, &pb, -1, &e.wide_ok));
        if (!cache && pack_fmt(c) == PK_F16X2) CAPDEC_TRY(weight_wide_ok(c, bt, (size_t)N * K, &e.wide_ok));
        e.bias = bias; e.act = act; e.resid = resid; e.ldr = ldr;
        if ((c->gemm_mode == GEMM_F16X2 || c->gemm_mode == GEMM_BF16X3) && !c->batch_invariant) {
            const size_t wsb = gemm_splitk_ws_bytes(M, N, K);
            if (wsb) {
                CAPDEC_TRY(c->splitk.ensure(wsb));
                e.splitk_ws = c->splitk.p;
                e.splitk_ws_bytes = c->splitk.cap;
                }
        }
        if (c->gemm_mode == GEMM_F16X2) {
            ProfScope ps(c, F_GEMM_H2P, 2.0 * M * (double)N * K);
            return launch_gemm_f16x2p(c->stream, pa, pb, cc, ldc, M, N, K, e);
        }
        if (mode_single(c)) {
            ProfScope ps(c, F_GEMM_BF16P, 2.0 * M * (double)N * K);
            return gemm_single(c, pa, pb, cc, ldc, M, N, K, e);
        }
        ProfScope ps(c, F_GEMM_X3P, 2.0 * M * (double)N * K);
        return launch_gemm_bf16x3p(c->stream, pa, pb, cc, ldc, M, N, K, e);
    }
    return gemm(c, a, lda, bt, ldb, cc, ldc, M, N, K, bias, act, resid, ldr, /*weight=*/cache);
}

int capdec_preprocess_images(capdec_ctx *c, const uint8_t *d_rgb, const int64_t *offsets, const int32_t *heights,
                             const int32_t *widths, int n, int n_px, int stretch, const float *mean, const float *stdv,
                             float *d_out) {
    CAPDEC_CHECK(c && (n == 0 || (d_rgb && offsets && heights && widths && mean && stdv && d_out)),
                 "preprocess_images: null argument");
    CAPDEC_CHECK(n >= 0 && n_px >= 1 && n_px <= 1024, "preprocess_images: bad sizes");
    CAPDEC_HIP(hipSetDevice(c->device));
    if (n == 0) return 0;
    std::vector<ImageDesc> desc((size_t)n);
    long long ioff = 0;
    int max_h = 0;
    for (int i = 0; i < n; ++i) {
        const int H = heights[i], W = widths[i];
        CAPDEC_CHECK(H >= 1 && W >= 1 && H <= 16384 && W <= 16384, "preprocess_images: image size out of range");
        ImageDesc &d = desc[(size_t)i];
        d.off = offsets[i];
        d.ioff = ioff;
        d.H = H;
        d.W = W;
        if (stretch) {               // clip_transform_full (predictions_runner.py:116-122): Resize((n_px, n_px)), no crop
            d.rh = d.rw = n_px;
            d.top = d.left = 0;
        } else {                     // torchvision Resize(n_px): shorter side -> n_px, longer = int(n_px * long / short);
            if (W <= H) {            // CenterCrop: origin int(round((size - n_px) / 2.0)), round-half-even like Python
                d.rw = n_px;
                d.rh = (int)((double)((long long)n_px * H) / (double)W);
            } else {
                d.rh = n_px;
                d.rw = (int)((double)((long long)n_px * W) / (double)H);
            }
            d.top = (int)nearbyint((d.rh - n_px) / 2.0);
            d.left = (int)nearbyint((d.rw - n_px) / 2.0);
        }
        ioff += (long long)H * n_px * 3;
        max_h = std::max(max_h, H);
    }
    CAPDEC_TRY(c->p_desc.ensure(desc.size() * sizeof(ImageDesc)));
    CAPDEC_TRY(c->p_inter.ensure((size_t)ioff));
    CAPDEC_HIP(hipMemcpyAsync(c->p_desc.p, desc.data(), desc.size() * sizeof(ImageDesc), hipMemcpyHostToDevice, c->stream));
    CAPDEC_HIP(hipStreamSynchronize(c->stream));      // `desc` is pageable host memory about to go out of scope
    ProfScope ps(c, F_OTHER);
    return launch_preprocess(c->stream, d_rgb, c->p_desc.as<ImageDesc>(), n, max_h, n_px, c->p_inter.as<uint8_t>(),
                             d_out, mean, stdv);
}


int capdec_comm_unique_id(char *id) {
    static_assert(CAPDEC_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    CAPDEC_CHECK(id != nullptr, "comm_unique_id: null id");
    CAPDEC_CHECK(rccl() != nullptr, "comm: librccl.so.1 could not be loaded (set CAPDEC_RCCL_LIB)");
    ncclUniqueId u;
    CAPDEC_NCCL(rccl()->GetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

int capdec_comm_init(capdec_ctx *c, int rank, int nranks, const char *id) {
    CAPDEC_CHECK(c && id && nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: bad argument");
    CAPDEC_CHECK(c->comm == nullptr, "comm_init: this context already has a communicator");
    CAPDEC_CHECK(rccl() != nullptr, "comm: librccl.so.1 could not be loaded (set CAPDEC_RCCL_LIB)");
    CAPDEC_HIP(hipSetDevice(c->device));
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    CAPDEC_NCCL(rccl()->CommInitRank(&c->comm, nranks, u, rank));
    c->comm_rank = rank;
    c->comm_world = nranks;
    return 0;
}

int capdec_comm_destroy(capdec_ctx *c) {
    CAPDEC_CHECK(c, "null context");
    if (c->comm) {
        (void)hipStreamSynchronize(c->stream);
        CAPDEC_NCCL(rccl()->CommDestroy(c->comm));
        c->comm = nullptr;
    }
    c->comm_rank = 0;
    c->comm_world = 1;
    return 0;
}

int capdec_shard_bounds(int n_total, int rank, int nranks, int *lo, int *hi) {
    CAPDEC_CHECK(lo && hi && n_total >= 0 && nranks >= 1 && rank >= 0 && rank < nranks, "shard_bounds: bad argument");
    const int per = (n_total + nranks - 1) / nranks;
    *lo = std::min(rank * per, n_total);
    *hi = std::min(*lo + per, n_total);
    return 0;
}

int capdec_gather_rows(capdec_ctx *c, const void *d_local, int n_local, int row_elems, int n_total, void *d_global) {
    CAPDEC_CHECK(c && n_local >= 0 && row_elems >= 1 && n_total >= 0 && (n_total == 0 || d_global) &&
                     (n_local == 0 || d_local), "gather_rows: bad argument");
    CAPDEC_HIP(hipSetDevice(c->device));
    const size_t row_b = (size_t)row_elems * 4;
    if (c->comm == nullptr) {
        CAPDEC_CHECK(n_local == n_total, "gather_rows: this rank holds only part of the rows but the context has no "
                                         "communicator (capdec_comm_init)");
        if (n_total && d_local != d_global)
            CAPDEC_HIP(hipMemcpyAsync(d_global, d_local, (size_t)n_total * row_b, hipMemcpyDeviceToDevice, c->stream));
        return 0;
    }
    int lo = 0, hi = 0;
    CAPDEC_TRY(capdec_shard_bounds(n_total, c->comm_rank, c->comm_world, &lo, &hi));
    CAPDEC_CHECK(n_local == hi - lo, "gather_rows: n_local is not this rank's shard of n_total (capdec_shard_bounds)");
    if (n_total == 0) return 0;
    const int per = (n_total + c->comm_world - 1) / c->comm_world;
    CAPDEC_TRY(c->g_pad.ensure((size_t)per * row_b));
    CAPDEC_TRY(c->g_all.ensure((size_t)per * c->comm_world * row_b));
    CAPDEC_HIP(hipMemsetAsync(c->g_pad.p, 0, (size_t)per * row_b, c->stream));
    if (n_local)
        CAPDEC_HIP(hipMemcpyAsync(c->g_pad.p, d_local, (size_t)n_local * row_b, hipMemcpyDeviceToDevice, c->stream));
    CAPDEC_NCCL(rccl()->AllGather(c->g_pad.p, c->g_all.p, (size_t)per * row_elems, ncclInt32, c->comm, c->stream));
    CAPDEC_HIP(hipMemcpyAsync(d_global, c->g_all.p, (size_t)n_total * row_b, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int capdec_gather_ids(capdec_ctx *c, const int32_t *d_ids, const int32_t *d_lens, const float *d_scores, int n_local,
                      int T, int n_total, int32_t *d_ids_global, int32_t *d_lens_global, float *d_scores_global) {
    CAPDEC_CHECK(c && T >= 1, "gather_ids: bad argument");
    CAPDEC_TRY(capdec_gather_rows(c, d_ids, n_local, T, n_total, d_ids_global));
    CAPDEC_TRY(capdec_gather_rows(c, d_lens, n_local, 1, n_total, d_lens_global));
    if (d_scores && d_scores_global) CAPDEC_TRY(capdec_gather_rows(c, d_scores, n_local, 1, n_total, d_scores_global));
    CAPDEC_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int capdec_decode_stats(capdec_ctx *c, int *steps, int *compactions, long long *row_steps) {
    CAPDEC_CHECK(c, "null context");
    if (steps) *steps = c->stat_steps;
    if (compactions) *compactions = c->stat_compactions;
    if (row_steps) *row_steps = c->stat_row_steps;
    return 0;
}

int capdec_decode_counters(capdec_ctx *c, double *kv_slots_per_position, long long *saturated_quads) {
    CAPDEC_CHECK(c, "null context");
    CAPDEC_HIP(hipSetDevice(c->device));
    if (kv_slots_per_position) *kv_slots_per_position = c->stat_kv_pos > 0 ? c->stat_kv_slots / c->stat_kv_pos : 0.0;
    if (saturated_quads) {
        CAPDEC_HIP(hipStreamSynchronize(c->stream));
        *saturated_quads = (long long)(sat_count_gemm_f16x2(true) + sat_count_gemm_h2w(true) + sat_count_gemm_pp(true) + sat_count_gemm_bf16x3(true) +
                                       sat_count_elementwise(true) + sat_count_attention(true) + sat_count_resnet(true));
    }
    return 0;
}

int capdec_timer_start(capdec_ctx *c) {
    CAPDEC_CHECK(c, "null context");
    CAPDEC_HIP(hipEventRecord(c->t0, c->stream));
    return 0;
}
int capdec_timer_stop_ms(capdec_ctx *c, float *ms) {
    CAPDEC_CHECK(c && ms, "null argument");
    CAPDEC_HIP(hipEventRecord(c->t1, c->stream));
    CAPDEC_HIP(hipEventSynchronize(c->t1));
    CAPDEC_HIP(hipEventElapsedTime(ms, c->t0, c->t1));
    return 0;
}
int capdec_profile_enable(capdec_ctx *c, int on) {
    CAPDEC_CHECK(c, "null context");
    c->prof.on = on != 0;       // on = N > 1: time every N-th launch of each family (sampling keeps the event
    c->prof.every = on > 1 ? on : 1;   // overhead out of a timed region; pick N coprime to the per-layer launch cycle)
    return 0;
}
int capdec_profile_reset(capdec_ctx *c) {
    CAPDEC_CHECK(c, "null context");
    CAPDEC_TRY(prof_collect(c));
    for (int f = 0; f < F_COUNT; ++f) { c->prof.ms[f] = 0; c->prof.flops[f] = 0; c->prof.launches[f] = 0; c->prof.calls[f] = 0; }
    return 0;
}
int capdec_profile_get(capdec_ctx *c, int *count, const char **names, float *ms, int64_t *launches, double *flops,
                       int64_t *calls) {
    CAPDEC_CHECK(c && count, "null argument");
    CAPDEC_CHECK(*count >= F_COUNT, "profile_get: *count must hold the capacity of the caller's arrays (>= 24 is always enough)");
    CAPDEC_TRY(prof_collect(c));
    *count = F_COUNT;
    for (int f = 0; f < F_COUNT; ++f) {
        if (names) names[f] = kFamilyNames[f];
        if (ms) ms[f] = (float)c->prof.ms[f];
        if (launches) launches[f] = c->prof.launches[f];
        if (flops) flops[f] = c->prof.flops[f];
        if (calls) calls[f] = c->prof.calls[f];
    }
    return 0;
}

}  // extern "C"
