// C ABI of libcapdec_hip.so (include/capdec.h), part 1: context lifetime, streams, precision modes, raw device memory,
// timers, the per-family profiler, decode statistics -- and the ONE place the environment is read (config.h).
#include "context.h"

namespace capdec {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

const char *const kFamilyNames[F_COUNT] = {"gemm_f32", "gemm_f32_lmhead_topk", "attn_decode", "attn_prefill",
                                           "layernorm", "embed", "select", "attn_mapper", "other", "gemm_bf16x3",
                                           "gemm_bf16x3_lmhead_topk", "gemm_bf16x3p", "gemm_x1",
                                           "gemm_x1_lmhead_topk", "gemm_f16x2p", "gemm_f16x2p_lmhead_topk",
                                           "pack_activations", "lmhead_second_pass"};

int prof_collect(capdec_ctx *c) {
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


// ---------------------------------------------------------------------------- environment knobs (config.h)
const Tuning &default_tuning() {
    static const Tuning t;
    return t;
}
static bool env_int(const char *name, int *out) {
    const char *e = getenv(name);
    if (!e || !*e) return false;
    *out = atoi(e);
    return true;
}
static void env_flag(const char *name, bool *out) {      // "0" switches off, anything else on
    int v;
    if (env_int(name, &v)) *out = v != 0;
}
bool tuning_from_env(Tuning *t, std::string *err) {
    *t = Tuning();
    if (const char *e = getenv("CAPDEC_GEMM_MODE")) {
        const std::string m(e);
        if (m != "f32" && m != "bf16" && m != "bf16x3" && m != "f16" && m != "f16x2") {     // a typo must not silently select another precision
            *err = "create: CAPDEC_GEMM_MODE=" + m + " is not one of f16x2 | bf16x3 | f32 | bf16 | f16";
            return false;
        }
        t->gemm_mode = m == "f32" ? GEMM_F32 : m == "bf16" ? GEMM_BF16 : m == "bf16x3" ? GEMM_BF16X3 : m == "f16" ? GEMM_F16 : GEMM_F16X2;
    }
    env_flag("CAPDEC_BATCH_INVARIANT", &t->batch_invariant);
    env_flag("CAPDEC_COMPACT", &t->compact);
    env_flag("CAPDEC_X3_PACKA", &t->x3_pack_a);
    env_flag("CAPDEC_X3_CHAIN", &t->pack_chain);
    env_flag("CAPDEC_SPLITK", &t->splitk);
    env_flag("CAPDEC_SPLITK_MID", &t->splitk_mid);
    env_flag("CAPDEC_X1_SPLITK", &t->x1_splitk);
    env_flag("CAPDEC_FUSE_LN", &t->fuse_ln);
    env_int("CAPDEC_H2_PERSIST", &t->h2_persist);
    if (env_int("CAPDEC_H2W", &t->h2w)) {
        const int v = t->h2w;
        bool ok = v == 0 || v == 1 || v == 2 || v == 8 || v == 10 || v == 14;
#ifdef CAPDEC_MEASURE
        ok = ok || v == 3 || v == 6 || v == 12;
#endif
        if (!ok) {
            *err = "create: CAPDEC_H2W=" + std::to_string(v) + " is not a geometry of this build (0, 1, 2, 8, 10, 14)";
            return false;
        }
    }
    env_int("CAPDEC_PP", &t->pp);
    env_flag("CAPDEC_LMHEAD_WIDE", &t->lmhead_wide);
    env_flag("CAPDEC_TRAIN_F16X2", &t->train_f16x2);
    env_flag("CAPDEC_TRAIN_ATTN_BLK", &t->train_attn_blk);
    env_flag("CAPDEC_LMHEAD_K3", &t->lmhead_k3);
    env_int("CAPDEC_LMHEAD_K3_MAX", &t->lmhead_k3_max);
    env_flag("CAPDEC_KV_DIRECT", &t->kv_direct);
    env_flag("CAPDEC_CLIP_TRUNC", &t->clip_trunc);
    env_flag("CAPDEC_RN_PACKED", &t->rn_packed);
    env_flag("CAPDEC_RN_IMPLICIT", &t->rn_implicit);
    t->hook_packa = getenv("CAPDEC_HOOK_PACKA") != nullptr;
    t->hook_cache = getenv("CAPDEC_HOOK_CACHE") != nullptr;
    if (const char *e = getenv("CAPDEC_RCCL_LIB")) t->rccl_lib = e;
#ifdef CAPDEC_MEASURE
    int v;
    if (env_int("CAPDEC_H2_NS", &v) && (v == 3 || v == 5)) t->h2_ns = v;
    env_int("CAPDEC_H2_ABL", &t->h2_abl);
    if (env_int("CAPDEC_X1_NS", &v) && v == 4) t->x1_ns = 4;
    env_int("CAPDEC_ABL_DMA", &t->x3_abl_dma);
    env_int("CAPDEC_X3_TILE_M", &t->x3_tile_m);
    env_int("CAPDEC_GEMM_BK", &t->f32_bk);
    env_int("CAPDEC_LMHEAD_BK", &t->f32_lmhead_bk);
    env_int("CAPDEC_ATT_PRELOAD", &t->att_preload);
    env_int("CAPDEC_ATT_WSYNC", &t->att_wsync);
    env_int("CAPDEC_ATT_DMA", &t->att_dma);
    if (env_int("CAPDEC_ATT_OCC", &v) && v == 3) t->att_occ = 3;
    env_int("CAPDEC_ATT_NA", &t->att_na);
    env_int("CAPDEC_PP_ABL", &t->pp_abl);
    if (const char *e = getenv("CAPDEC_PP_STAMPS")) t->pp_stamps = e;
    env_int("CAPDEC_LMHEAD_K1", &t->lmhead_k1);
#endif
    return true;
}

}  // namespace capdec

using namespace capdec;

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
    Tuning tune;
    std::string terr;
    if (!tuning_from_env(&tune, &terr)) {      // before anything is allocated: a bad environment leaks nothing
        set_error(terr);
        return 1;
    }
    std::unique_ptr<capdec_ctx> c(new capdec_ctx());
    c->device = device_id;
    c->tune = tune;
    c->gemm_mode = tune.gemm_mode;
    c->batch_invariant = tune.batch_invariant;
    c->compact = tune.compact;
    c->pack_a = tune.x3_pack_a;
    c->pack_chain = tune.pack_chain;
    CAPDEC_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    CAPDEC_HIP(hipEventCreate(&c->t0));
    CAPDEC_HIP(hipEventCreate(&c->t1));
    CAPDEC_HIP(hipHostMalloc((void **)&c->alive_host, 2 * sizeof(int), hipHostMallocDefault));
    *out = c.release();
    return 0;
}

void capdec_destroy(capdec_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    comm_release(c);
    c->g_pad.release();
    c->g_all.release();
    free_all(c->gpt.owned);
    free_all(c->map.owned);
    free_all(c->clip_text.owned);
    free_all(c->clip_vision.owned);
    free_all(c->clip_resnet.owned);
    drop_planes(c);
    train_release(c);
    c->x3_tmp.release();
    DBuf *bufs[] = {&c->h, &c->x, &c->qkv, &c->att, &c->ff, &c->xl, &c->tmax, &c->tsum, &c->cval, &c->cidx,
                    &c->lse, &c->topv, &c->topi, &c->kc, &c->vc, &c->tokens, &c->scores, &c->seq, &c->stopped,
                    &c->done, &c->anc, &c->next_tok, &c->alive, &c->gids, &c->glens, &c->m_hid, &c->m_lin, &c->m_seq,
                    &c->m_x, &c->m_qkv, &c->m_att, &c->m_ff, &c->t_idx, &c->t_patch, &c->t_pout, &c->xpk, &c->apk, &c->fpk, &c->cmap, &c->kvstat, &c->lmflag, &c->xpk2, &c->p_desc, &c->p_inter, &c->splitk, &c->absmax, &c->a_tmp,
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
#ifdef CAPDEC_MEASURE
int capdec_set_debug_diverge(capdec_ctx *c, int on) {      // measurement builds only (include/capdec.h)
    CAPDEC_CHECK(c, "null context");
    c->diverge = on != 0;
    return 0;
}
#endif
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

int capdec_decode_stats(capdec_ctx *c, int *steps, int *compactions, long long *row_steps) {
    CAPDEC_CHECK(c, "null context");
    if (steps) *steps = c->stat_steps;
    if (compactions) *compactions = c->stat_compactions;
    if (row_steps) *row_steps = c->stat_row_steps;
    return 0;
}

int capdec_set_compact(capdec_ctx *c, int on) {
    CAPDEC_CHECK(c, "null context");
    c->compact = on != 0;
    return 0;
}

int capdec_decode_step_rows(capdec_ctx *c, int *rows, int cap, int *n) {
    CAPDEC_CHECK(c && n && (rows || cap <= 0), "null argument");
    *n = (int)c->stat_step_rows.size();
    for (int i = 0; i < *n && i < cap; ++i) rows[i] = c->stat_step_rows[i];
    return 0;
}

int capdec_decode_second_pass_rows(capdec_ctx *c, long long *rows) {
    CAPDEC_CHECK(c && rows, "null argument");
    *rows = 0;
    if (!c->lmflag_live) return 0;
    CAPDEC_HIP(hipSetDevice(c->device));
    int total = 0;
    CAPDEC_HIP(hipMemcpyAsync(&total, c->lmflag.as<int>() + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    CAPDEC_HIP(hipStreamSynchronize(c->stream));
    *rows = total;
    return 0;
}

int capdec_decode_counters(capdec_ctx *c, double *kv_slots_per_position, long long *saturated_quads) {
    CAPDEC_CHECK(c, "null context");
    CAPDEC_HIP(hipSetDevice(c->device));
    if (kv_slots_per_position) {
        if (c->kvstat_n > 0) {      // the decode loop only accumulates on the device: read the per-caption sums back now
            std::vector<unsigned> hs((size_t)c->kvstat_n * 2);
            CAPDEC_HIP(hipMemcpyAsync(hs.data(), c->kvstat.p, hs.size() * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
            CAPDEC_HIP(hipStreamSynchronize(c->stream));
            c->stat_kv_slots = c->stat_kv_pos = 0.0;
            for (int i = 0; i < c->kvstat_n; ++i) { c->stat_kv_slots += hs[2 * i]; c->stat_kv_pos += hs[2 * i + 1]; }
            c->kvstat_n = 0;
        }
        *kv_slots_per_position = c->stat_kv_pos > 0 ? c->stat_kv_slots / c->stat_kv_pos : 0.0;
    }
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
