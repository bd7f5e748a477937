// The GEMM planner behind every projection of the path: the cache of packed weight planes, which kernel family and
// operand format a launch takes in the context's precision mode, the LayerNorm -> GEMM and packed-chain forms of the block
// stack, and capdec_gemm_f32 (the test / micro-benchmark hook onto the same launchers).  Tile geometry and split-K are
// chosen one level down (launch_gemm_f16x2p / pp_plan / h2w_plan) from the shape and the context's Tuning.
#include "context.h"

namespace capdec {

// ---------------------------------------------------------------------------- GEMM wrappers
// bf16 planes of an [N, K] fp32 weight matrix: made on first use, dropped whenever weights are reloaded
void drop_planes(capdec_ctx *c) {
    for (auto &kv : c->planes) (void)hipFree(kv.second.p);
    c->planes.clear();
}
void drop_planes_of(capdec_ctx *c, const void *weight) {       // every format cached for this weight
    for (auto it = c->planes.lower_bound({weight, -1}); it != c->planes.end() && it->first.first == weight;) {
        (void)hipFree(it->second.p);
        it = c->planes.erase(it);
    }
}
// packed operand format of the block-stack / lm_head GEMMs in the current mode (bf16x3.h): two fp16 planes (f16x2),
// three bf16 planes (bf16x3), or ONE bf16 / fp16 plane (the reduced-precision modes)
int pack_fmt(const capdec_ctx *c) {
    switch (c->gemm_mode) {
        case GEMM_F16X2: return PK_F16X2;
        case GEMM_BF16: return PK_BF16X1;
        case GEMM_F16: return PK_F16X1;
        default: return PK_BF16X3;
    }
}
static int gemm_single(capdec_ctx *c, const void *A, const void *B, float *C, int ldc, int M, int N, int K,
                       const GemmEpilogue &e) {
    return launch_gemm_x1(c->stream, A, B, C, ldc, M, N, K, e, pack_fmt(c));
}
int pack_any(capdec_ctx *c, const float *W, int N, int K, int fmt, void *out) {
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
int planes_of(capdec_ctx *c, const float *W, int N, int K, bool cache, const void **out, int fmt_override,
                     bool *wide_ok) {
    const int fmt = fmt_override >= 0 ? fmt_override : pack_fmt(c);
    const size_t n = (size_t)N * K, bytes = x3_packed_bytes(N, K, fmt);
    if (wide_ok) *wide_ok = false;
    if (cache) {
        auto it = c->planes.find({W, fmt});
        if (it != c->planes.end()) {
            if (it->second.n == n) {
                *out = it->second.p;
                if (wide_ok) *wide_ok = it->second.wide_ok;
                return 0;
            }
            (void)hipFree(it->second.p);      // same address, different matrix
            c->planes.erase(it);
        }
        bool ok = false;
        if (fmt == PK_F16X2) CAPDEC_TRY(weight_wide_ok(c, W, n, &ok));      // (before the allocation: a failure leaks nothing)
        void *p = nullptr;
        CAPDEC_HIP(hipMalloc(&p, bytes));
        if (pack_any(c, W, N, K, fmt, p)) {
            (void)hipFree(p);
            return 1;
        }
        c->planes[{W, fmt}] = capdec_ctx::Planes{p, n, fmt, ok};
        *out = p;
        if (wide_ok) *wide_ok = ok;
        return 0;
    }
    CAPDEC_TRY(c->x3_tmp.ensure(bytes));
    CAPDEC_TRY(pack_any(c, W, N, K, fmt, c->x3_tmp.p));
    *out = c->x3_tmp.p;
    return 0;
}

int gemm(capdec_ctx *c, const float *A, int lda, const float *Bt, int ldb, float *C, int ldc, int M, int N,
                int K, const float *bias, int act, const float *resid, int ldr, bool weight) {
    GemmEpilogue e;
    e.tune = &c->tune;
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
        const size_t wsb = c->batch_invariant ? 0 : gemm_splitk_ws_bytes(M, N, K, c->tune);
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

// C [N1, N2] = Xa^T Xb for fp32 Xa [rows, N1], Xb [rows, N2] (row-major, row strides lda / ldb): the weight-gradient product of the train
// step (dW = dY^T X, K = the token rows).  Both operands are packed TRANSPOSED straight from their row-major form
// (launch_pack_planes_h2_t: no fp32 transposed copies), then the two-fp16-plane kernels run with K = rows padded to 64.
int gemm_tn(capdec_ctx *c, const float *Xa, int lda, const float *Xb, int ldb, int rows, int N1, int N2, float *C, int ldc) {
    const int Kp = (rows + 63) / 64 * 64;
    CAPDEC_TRY(c->a_tmp.ensure(x3_packed_bytes(N1, Kp, PK_F16X2)));
    CAPDEC_TRY(c->x3_tmp.ensure(x3_packed_bytes(N2, Kp, PK_F16X2)));
    {
        ProfScope ps(c, F_PACK);
        CAPDEC_TRY(launch_pack_planes_h2_t(c->stream, Xa, lda, rows, N1, Kp, c->a_tmp.p));
        CAPDEC_TRY(launch_pack_planes_h2_t(c->stream, Xb, ldb, rows, N2, Kp, c->x3_tmp.p));
    }
    GemmEpilogue e;
    e.tune = &c->tune;
    if (c->batch_invariant) e.invariant = true;
    const size_t wsb = c->batch_invariant ? 0 : gemm_splitk_ws_bytes(N1, N2, Kp, c->tune);
    if (wsb) {
        CAPDEC_TRY(c->splitk.ensure(wsb));
        e.splitk_ws = c->splitk.p;
        e.splitk_ws_bytes = c->splitk.cap;
    }
    ProfScope ps(c, F_GEMM_H2P, 2.0 * N1 * (double)N2 * rows);
    return launch_gemm_f16x2p(c->stream, c->a_tmp.p, c->x3_tmp.p, C, ldc, N1, N2, Kp, e);
}

// LayerNorm -> GEMM with the normalised rows handed over in packed split-bf16 form (never fp32 in HBM).
// Returns 1 in *done when the packed path ran; otherwise the caller runs the fp32-activation path.
bool use_packed_a(capdec_ctx *c, int K) {
    return (mode_single(c) || c->gemm_mode == GEMM_F16X2 || (c->gemm_mode == GEMM_BF16X3 && c->pack_a)) && K % 64 == 0;
}

// C = act(Apk . W^T + bias) + resid with A already packed; packed_out != nullptr: the result is written as the
// packed A operand of the next GEMM instead of fp32 C
// (next_ln: the LayerNorm that follows this GEMM in the block stack; when the launch splits K it is fused into the
//  reduce pass, its packed output lands in c->xpk and *ln_done is set -- see GemmEpilogue)
int gemm_packed(capdec_ctx *c, const void *Apk, const float *W, float *C, int ldc, int M, int N, int K,
                       const float *bias, int act, const float *resid, int ldr,
                       void *packed_out, const NextLn *next_ln, const void *resid_packed,
                       const QkvScatter *qkv_scatter) {
    const void *pl = nullptr;
    GemmEpilogue e;
    e.tune = &c->tune;
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
        const bool x1_split = c->tune.x1_splitk;
        const size_t wsb = ((mode_single(c) && !x1_split) || c->batch_invariant || qkv_scatter) ? 0 : gemm_splitk_ws_bytes(M, N, K, c->tune);
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
int ln_gemm_packed(capdec_ctx *c, const float *h, int ldh, const float *lnw, const float *lnb, float eps,
                          const float *W, float *C, int ldc, int M, int N, int K, const float *bias, int act,
                          void *packed_out, bool ln_ready, const QkvScatter *qkv_scatter) {
    CAPDEC_TRY(c->xpk.ensure(x3_packed_bytes_host(M, K)));
    if (!ln_ready) {
        ProfScope ps(c, F_LN);
        CAPDEC_TRY(launch_layernorm_packed(c->stream, h, ldh, lnw, lnb, eps, c->xpk.p, M, K, pack_fmt(c)));
    }
    return gemm_packed(c, c->xpk.p, W, C, ldc, M, N, K, bias, act, nullptr, 0, packed_out, nullptr, nullptr, qkv_scatter);
}


}  // namespace capdec

using namespace capdec;

extern "C" {

int capdec_gemm_f32(capdec_ctx *c, const float *a, int lda, const float *bt, int ldb, float *cc, int ldc, int M, int N,
                    int K, const float *bias, const float *resid, int ldr, int act) {
    CAPDEC_CHECK(c && a && bt && cc, "gemm: null argument");
    CAPDEC_HIP(hipSetDevice(c->device));
    const bool cache = c->tune.hook_cache;   // benchmarking: treat Bt as a resident weight
    const bool packa = c->tune.hook_packa;   // tests / benchmarking: pre-packed A (the LayerNorm -> GEMM path)
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
        e.tune = &c->tune;
    e.tune = &c->tune;
        CAPDEC_TRY(planes_of(c, bt, N, K, cache, &pb, -1, &e.wide_ok));
        // (an uncached B keeps the two-accumulator kernels: measuring max |b| costs a reduction and a stream synchronisation
        //  per call -- paid only when a geometry is FORCED, CAPDEC_H2W >= 2: how the parity tests reach the wide tiles)
        if (!cache && pack_fmt(c) == PK_F16X2 && c->tune.h2w >= 2) CAPDEC_TRY(weight_wide_ok(c, bt, (size_t)N * K, &e.wide_ok));
        e.bias = bias; e.act = act; e.resid = resid; e.ldr = ldr;
        if ((c->gemm_mode == GEMM_F16X2 || c->gemm_mode == GEMM_BF16X3) && !c->batch_invariant) {
            const size_t wsb = gemm_splitk_ws_bytes(M, N, K, c->tune);
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


}  // extern "C"
