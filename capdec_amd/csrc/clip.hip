// CLIP towers on the block stack of decode.hip: ViT-B/32 text / image encoders, the ModifiedResNet (RN50x4) image tower
// with its attention pool, and the PIL-exact image preprocessing entry point.
#include "context.h"

namespace capdec {

// One chunk of captions through the text tower.  perm == nullptr: captions [0, n) of `tokens` with all L positions, features
// to out[0 .. n).  perm != nullptr (device, n entries; pos = every caption's EOT position, device): caption i of the chunk is
// row perm[i] of `tokens`, only positions [0, P) are computed -- P > the largest EOT position of the chunk: the tower is
// causal and the feature is read at the EOT row, so nothing behind a caption's EOT can reach it -- and the features are
// scattered to out[perm[i]].
static int clip_text_chunk(capdec_ctx *c, const int *tokens, int n, float *out, const int *perm = nullptr, const int *pos = nullptr,
                           int P = 0) {
    Tower &t = c->clip_text;
    const int d = t.d, L = t.ctx;
    if (!perm) P = L;
    KvCache kv;
    kv_geometry(kv, n, P, t.n_head, d / t.n_head);
    CAPDEC_TRY(ensure_body_ws(c, n * P, d));
    CAPDEC_TRY(c->t_idx.ensure((size_t)n * 4));
    CAPDEC_TRY(c->xl.ensure((size_t)2 * n * d * 4));
    { ProfScope ps(c, F_EMBED); CAPDEC_TRY(launch_clip_text_embed(c->stream, tokens, t.tok_emb, t.pos_emb, c->h.as<float>(), n, L, d, P, perm)); }
    StepShape sp{};
    sp.prefill = true;
    sp.ncap = n;
    sp.P = P;
    sp.beam = 1;
    StackCfg cfg{&t.layers, t.n_layer, d, 1e-5f, CAPDEC_ACT_QUICK_GELU, true, false};
    CAPDEC_TRY(stack_body(c, cfg, sp, kv));
    float *rows = c->xl.as<float>(), *rows_ln = c->xl.as<float>() + (size_t)n * d;
    {
        ProfScope ps(c, F_EMBED);
        if (perm) CAPDEC_TRY(launch_eot_rows(c->stream, pos, perm, c->t_idx.as<int>(), n, P));
        else CAPDEC_TRY(launch_eot_index(c->stream, tokens, c->t_idx.as<int>(), n, L));
        CAPDEC_TRY(launch_gather_rows(c->stream, c->h.as<float>(), c->t_idx.as<int>(), rows, n, d));
    }
    { ProfScope ps(c, F_LN); CAPDEC_TRY(launch_layernorm(c->stream, rows, d, t.lnf_w, t.lnf_b, 1e-5f, rows_ln, d, n, d)); }
    if (!perm) return gemm(c, rows_ln, d, t.proj_t, d, out, t.embed, n, t.embed, d, nullptr, CAPDEC_ACT_NONE);
    CAPDEC_TRY(c->t_pout.ensure((size_t)n * t.embed * 4));
    CAPDEC_TRY(gemm(c, rows_ln, d, t.proj_t, d, c->t_pout.as<float>(), t.embed, n, t.embed, d, nullptr, CAPDEC_ACT_NONE));
    ProfScope ps(c, F_EMBED);
    return launch_scatter_rows(c->stream, c->t_pout.as<float>(), perm, out, n, t.embed);
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

using namespace capdec;

static int conv_bn_forward(capdec_ctx *c, const ConvW &w, const float *in, int N, int H, int W, int stride, bool nchw3,
                           float *out, int act, const float *resid, int *Ho_, int *Wo_) {
    int Ho = H, Wo = W;
    if (w.k == 3) {
        Ho = (H + 2 - 3) / stride + 1;
        Wo = (W + 2 - 3) / stride + 1;
    }
    const int M = N * Ho * Wo;
    const bool fused = c->tune.rn_packed;
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
    e.tune = &c->tune;
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
    const bool implicit_on = c->tune.rn_implicit;
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


extern "C" {

int capdec_clip_encode_text(capdec_ctx *c, const int32_t *tokens, int n, float *out) {
    CAPDEC_CHECK(c && c->clip_text.loaded, "clip_encode_text: text tower not loaded");
    CAPDEC_CHECK(n >= 0 && (n == 0 || (tokens && out)), "clip_encode_text: bad argument");
    CAPDEC_HIP(hipSetDevice(c->device));
    const Tower &t = c->clip_text;
    const int L = t.ctx;
    const int row_budget = 4096 * L;            // token rows per chunk (4096 full-length captions: the workspaces' size class)
    if (!c->tune.clip_trunc || n == 0 || t.embed % 4 != 0) {
        const int chunk = 4096;
        for (int c0 = 0; c0 < n; c0 += chunk) {
            const int nc = std::min(chunk, n - c0);
            CAPDEC_TRY(clip_text_chunk(c, tokens + (size_t)c0 * L, nc, out + (size_t)c0 * t.embed));
        }
        return 0;
    }
    // Captions are short (COCO: ~14 tokens with SOT / EOT of the 77 the reference pads to, embeddings_generator.py:80-86) and
    // the tower is causal: a caption's feature is its EOT row, which sees positions <= EOT only.  So: every caption's EOT
    // position (one small kernel + an n-int copy), captions sorted by it on the host, chunks of neighbours in that order, each
    // computed with P = its own largest EOT + 1 positions per caption instead of 77 -- the GEMM / LayerNorm / attention rows of
    // a chunk shrink by 77 / P; features identical to the full-length computation up to fp32 summation order (the launch sizes
    // differ).  CAPDEC_CLIP_TRUNC=0 computes all 77 positions.
    DBuf &ib = c->p_inter;                      // [pos n][perm n] ints
    CAPDEC_TRY(ib.ensure((size_t)2 * n * sizeof(int)));
    int *pos_d = ib.as<int>(), *perm_d = pos_d + n;
    { ProfScope ps(c, F_EMBED); CAPDEC_TRY(launch_eot_index(c->stream, tokens, pos_d, n, L, /*flat=*/0)); }
    std::vector<int> pos((size_t)n), order((size_t)n);
    CAPDEC_HIP(hipMemcpyAsync(pos.data(), pos_d, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    CAPDEC_HIP(hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; ++i) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return pos[(size_t)a] < pos[(size_t)b]; });
    CAPDEC_HIP(hipMemcpyAsync(perm_d, order.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    CAPDEC_HIP(hipStreamSynchronize(c->stream));                    // (`order` is pageable host memory: it must outlive the copy)
    for (int i0 = 0; i0 < n;) {
        int m = 1;
        while (i0 + m < n && m < 65536 && (long long)(m + 1) * (pos[(size_t)order[(size_t)(i0 + m)]] + 1) <= row_budget) ++m;
        const int P = pos[(size_t)order[(size_t)(i0 + m - 1)]] + 1;
        CAPDEC_TRY(clip_text_chunk(c, tokens, m, out, perm_d + i0, pos_d, P));
        i0 += m;
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



}  // extern "C"
