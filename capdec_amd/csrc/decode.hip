// The KV-cached batched decode: the pre-LN block stack (GPT-2, and the CLIP towers that run on the same stack), the fused
// lm_head + candidate selection, the greedy / beam drivers with finished-caption compaction, the mapping networks and the
// prefix stage -- host-side orchestration only: every operation is a launcher of common.h enqueued on the context's stream.
#include "context.h"

namespace capdec {

int ensure_body_ws(capdec_ctx *c, int M, int d) {
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
int stack_body(capdec_ctx *c, const StackCfg &g, const StepShape &s, const KvCache &kv) {
    const int d = g.d, M = s.prefill ? s.ncap * s.P : s.rows;
    float *h = c->h.as<float>(), *x = c->x.as<float>(), *qkv = c->qkv.as<float>(), *att = c->att.as<float>(),
          *ff = c->ff.as<float>();
    // packed chain (bf16x3 mode): LN1 -> [packed] -> qkv GEMM -> attention -> [packed] -> c_proj (+h) -> LN2 ->
    // [packed] -> fc GEMM + act -> [packed] -> mlp c_proj (+h): every GEMM operand moves by LDS-DMA, and the
    // attention / MLP intermediates never exist in fp32 in HBM.
    const bool chain = use_packed_a(c, d) && c->pack_chain;
    const bool kv_direct = c->tune.kv_direct;
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
            // (the scatter epilogue is float4-only: a host whose c_attn bias is not 16-byte aligned keeps the other path)
        // (round 5: also in bf16 mode -- the one-plane qkv GEMM rounds K / V to bf16 as it writes them into the bf16 cache)
        const bool mode_ok = (c->gemm_mode == GEMM_F16X2 && !kv.bf16) || (c->gemm_mode == GEMM_BF16 && kv.bf16);
        const bool scatter = kv_direct && !s.prefill && g.keep_kv && mode_ok && use_packed_a(c, d) && w.bqkv &&
                             (((uintptr_t)w.bqkv | (uintptr_t)qkv) & 15) == 0 &&
                             d % GEMM_BN == 0 && (s.beam == 1 || s.beam == 5) &&
                             (c->batch_invariant || gemm_splitk_slices(M, 3 * d, d, c->tune) == 1);
        if (scatter) {
            sc.kc = reinterpret_cast<float *>(kv.bf16 ? (void *)kv.kp<__bf16>(kl) : (void *)kv.kp<float>(kl));
            sc.vc = reinterpret_cast<float *>(kv.bf16 ? (void *)kv.vp<__bf16>(kl) : (void *)kv.vp<float>(kl));
            sc.cmap = s.cmap;
            sc.beam = s.beam; sc.heads = kv.heads; sc.ctx = kv.ctx; sc.pos = s.L - 1; sc.d = d;
            sc.bf16 = kv.bf16;
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
    bool k3 = false;                 // the fused kernel kept 3 candidates per tile of a k = 5 selection (below)
    const void *wte_planes = nullptr;
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
            const bool lm_wide = c->tune.lmhead_wide;
            const int h2w = c->tune.h2w;
            if (wide_ok && !c->batch_invariant && ((lm_wide && h2w >= 1 && R >= 2048) || h2w >= 2)) {   // (CAPDEC_H2W >= 2: forced, tests)
                // Beam search (k = 5): keep THREE candidates per (row, 128-column tile) -- two selection rounds fewer in
                // every tile's epilogue.  The merge then knows exactly which rows that can have been too few for (some
                // tile's third candidate is still strictly better than the row's fifth: with 393 tiles a rare event) and
                // those rows alone go through the k = 5 kernel again, compacted; the result is the k = 5 result.
                k3 = c->tune.lmhead_k3 && k == 5 && !c->k3_off;
                CAPDEC_TRY(launch_gemm_h2w_topk(c->stream, c->xpk.p, pl, R, g.vocab, d, k3 ? 3 : k, inv_temp, c->tmax.as<float>(),
                                                c->tsum.as<float>(), c->cval.as<float>(), c->cidx.as<int>(), &c->tune));
                wte_planes = pl;
            } else
            CAPDEC_TRY(launch_gemm_f16x2p_topk(c->stream, c->xpk.p, pl, R, g.vocab, d, k, inv_temp, c->tmax.as<float>(),
                                               c->tsum.as<float>(), c->cval.as<float>(), c->cidx.as<int>()));
        } else if (mode_single(c)) {
            ProfScope ps(c, F_LMHEAD_BF16, 2.0 * R * (double)g.vocab * d);
            k3 = c->tune.lmhead_k3 && k == 5 && R >= 2048 && !c->batch_invariant && !c->k3_off;   // (as in the two-plane mode above)
            wte_planes = pl;
            CAPDEC_TRY(launch_gemm_x1_topk(c->stream, c->xpk.p, pl, R, g.vocab, d, k3 ? 3 : k, inv_temp, c->tmax.as<float>(),
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
                                        c->cidx.as<int>(), &c->tune));
    }
    if (k3) {
        CAPDEC_TRY(c->lmflag.ensure(((size_t)R + 2) * 4));
        CAPDEC_TRY(c->xpk2.ensure(x3_packed_bytes_host(R, d)));
        int *cnt = c->lmflag.as<int>(), *total = cnt + 1, *rows = cnt + 2;
        {
            ProfScope ps(c, F_SELECT);
            CAPDEC_HIP(hipMemsetAsync(cnt, 0, sizeof(int), c->stream));
            CAPDEC_TRY(launch_topk_merge_k3(c->stream, c->tmax.as<float>(), c->tsum.as<float>(), c->cval.as<float>(),
                                            c->cidx.as<int>(), R, nt, c->lse.as<float>(), c->topv.as<float>(),
                                            c->topi.as<int>(), rows, cnt, total));
        }
        // the second pass: gather, k = 5 kernel over the device-side row count, merge (the partial lists of the first pass
        // are dead once its merge has run: their buffers are reused)
        ProfScope ps(c, F_LMHEAD_2ND);
        CAPDEC_TRY(launch_gather_packed_rows(c->stream, c->xpk.p, d, rows, cnt, R, c->xpk2.p, pack_fmt(c)));
        if (mode_single(c))
            CAPDEC_TRY(launch_gemm_x1_topk_dev(c->stream, c->xpk2.p, wte_planes, cnt, g.vocab, d, inv_temp, c->tmax.as<float>(),
                                               c->tsum.as<float>(), c->cval.as<float>(), c->cidx.as<int>(), pack_fmt(c)));
        else
            CAPDEC_TRY(launch_gemm_h2w_topk_dev(c->stream, c->xpk2.p, wte_planes, cnt, g.vocab, d, inv_temp, c->tmax.as<float>(),
                                                c->tsum.as<float>(), c->cval.as<float>(), c->cidx.as<int>()));
        CAPDEC_TRY(launch_topk_merge_rows(c->stream, c->cval.as<float>(), c->cidx.as<int>(), cnt, rows, R, nt,
                                          c->topv.as<float>(), c->topi.as<int>()));
        c->lmflag_live = true;
        c->k3_rows += R;
        return 0;
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
void kv_geometry(KvCache &kv, int rows, int ctx, int heads, int hd) {
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
    const bool watch = c->lmflag_live && !c->k3_off;
    if (watch) CAPDEC_HIP(hipMemcpyAsync(c->alive_host + 1, c->lmflag.as<int>() + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    CAPDEC_HIP(hipStreamSynchronize(c->stream));
    *alive = *c->alive_host;
    // a vocabulary that clusters a row's best candidates inside one 128-column tile sends many rows through the lm_head's
    // second pass: past the break-even the rest of the call keeps k candidates per tile (same results either way)
    if (watch && c->k3_rows > 0 && (double)c->alive_host[1] * 1000.0 > (double)c->tune.lmhead_k3_max * (double)c->k3_rows) c->k3_off = true;
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
                        const int *forced = nullptr, float *stats = nullptr, int kv_stat_off = 0) {
    const Gpt2 &g = c->gpt;
    const int d = g.d;
    const int ctx = P + T - 1;
    const int rows = nc * beam;
    const int k = (greedy && stats) ? 2 : beam;   // candidates kept per row (teacher-forced statistics: top-2)
    const float inv_temp = 1.0f / (temperature > 0.f ? temperature : 1.0f);
    KvCache kv;
    CAPDEC_TRY(ensure_kv(c, kv, rows, ctx));
    kv.fixed_variant = c->batch_invariant;
    kv.tune = &c->tune;
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
        // (distinct-K/V-slot statistic: this chunk's slice of the per-call array decode_common zeroed; nothing is read back
        //  here -- capdec_decode_counters sums it when somebody asks)
        bs.kv_stat = c->kvstat.p ? c->kvstat.as<unsigned>() + (size_t)kv_stat_off * 2 : nullptr;
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
    // Poll cadence: a poll drains the stream (a 4-byte copy + a synchronisation: ~150 us of idle GPU by the time the host has
    // woken up and refilled the queue).  While captions ARE finishing a poll pays for itself -- every step that still carries
    // finished captions costs more: a step of >= 8192 rows takes >= 8 ms, one of ~3000 rows ~4 ms (round 6, captions that
    // stop after ~11 tokens: with a poll every 8 steps 25 000 rows were launched for 8 steps while 40 % of them had
    // finished; 519 k row-steps against 427 k alive) -- so the base interval is 1 step at >= 8192 rows, 2 at >= 2048, 4 at
    // >= 512, 8 below.  While NOTHING finishes (the synthetic headline weights never emit the stop id) every poll is pure
    // loss: each poll that finds no finished caption doubles the interval (up to 8), the first one that does resets it.
    int na = nc, next_poll = 1, backoff = 1, last_alive = nc + 1;      // (+ 1: the first poll never backs off)
    const int *cmap = nullptr;
    CAPDEC_TRY(c->cmap.ensure(((size_t)nc + 1) * 4));
    for (int i = 1; i < T; ++i) {
        if (i >= next_poll) {
            int alive = 0;
            CAPDEC_TRY(poll_alive(c, &alive));
            if (alive == 0) break;
            const int rows_now = alive * beam;
            const int base = rows_now >= 8192 ? 1 : rows_now >= 2048 ? 2 : rows_now >= 512 ? 4 : 8;
            backoff = alive < last_alive ? 1 : std::min(8, backoff * 2);
            last_alive = alive;
            next_poll = i + std::min(8, std::max(base, backoff));
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
        if ((int)c->stat_step_rows.size() < i) c->stat_step_rows.resize(i, 0);     // (chunks of one call add up step by step)
        c->stat_step_rows[i - 1] += arows;
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
    return 0;
}

static int decode_common(capdec_ctx *c, const float *prefix, int n, int P, int beam, bool greedy, int stop_id,
                         int alt_stop_id, int T, float temperature, int *ids, int *lens, float *scores,
                         int *order, const int *forced = nullptr, float *stats = nullptr) {
    CAPDEC_CHECK(c && c->gpt.loaded, "decode: GPT-2 weights not loaded");
    CAPDEC_CHECK(n >= 0 && P >= 1 && T >= 1, "decode: bad sizes");
    CAPDEC_CHECK(P + T - 1 <= c->gpt.n_pos, "decode: prefix + entry_length exceeds n_positions");
    CAPDEC_CHECK(P + T - 1 <= 1024 && T <= 1024, "decode: context or entry_length > 1024 not supported");
    CAPDEC_CHECK(beam >= 1 && beam <= 8, "decode: beam size must be in 1..8");
    CAPDEC_CHECK(c->gpt.d / c->gpt.n_head == 64, "decode: head_dim must be 64");
    CAPDEC_HIP(hipSetDevice(c->device));
    c->stat_steps = n > 0 ? 1 : 0;
    c->stat_compactions = 0;
    c->stat_row_steps = 0;
    c->stat_step_rows.clear();
    c->stat_kv_slots = c->stat_kv_pos = 0.0;
    c->kvstat_n = 0;
    if (n == 0) return 0;
    if (!greedy) {
        CAPDEC_TRY(c->kvstat.ensure((size_t)n * 2 * sizeof(unsigned)));
        CAPDEC_HIP(hipMemsetAsync(c->kvstat.p, 0, (size_t)n * 2 * sizeof(unsigned), c->stream));
        c->kvstat_n = n;
    }
    const int ctx = P + T - 1;
    const int chunk = chunk_captions(c, n, beam, ctx);
    // lm_head second-pass bookkeeping ([count, total, rows...], lm_head_select): sized once for the largest step, total zeroed
    CAPDEC_TRY(c->lmflag.ensure(((size_t)std::min(chunk, n) * beam + 2) * 4));
    CAPDEC_HIP(hipMemsetAsync(c->lmflag.p, 0, 2 * sizeof(int), c->stream));
    c->lmflag_live = false;
    c->k3_off = false;
    c->k3_rows = 0;
    for (int c0 = 0; c0 < n; c0 += chunk) {
        const int nc = std::min(chunk, n - c0);
        CAPDEC_TRY(decode_chunk(c, prefix + (size_t)c0 * P * c->gpt.d, nc, P, beam, greedy, stop_id, alt_stop_id, T,
                                temperature, ids + (size_t)c0 * beam * T, lens + (size_t)c0 * beam,
                                scores ? scores + (size_t)c0 * beam : nullptr,
                                order ? order + (size_t)c0 * beam : nullptr,
                                forced ? forced + (size_t)c0 * T : nullptr, stats ? stats + (size_t)c0 * T * 3 : nullptr, c0));
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


}  // namespace capdec

using namespace capdec;

extern "C" {

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
    CAPDEC_CHECK(embeds && logits && n >= 1 && L >= 1 && L <= 1024 && L <= c->gpt.n_pos, "gpt2_logits: bad argument");
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


}  // extern "C"
