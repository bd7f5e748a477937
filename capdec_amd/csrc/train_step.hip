// The train step itself (train.h; reference train.py:344-354): mapper forward, GPT-2 forward keeping what the backward
// needs, the loss of :348-349, the backward pass down to the mapper, the update -- and the kernels only this file uses: GELU,
// GPT-2's dropouts, the embedding / loss-row maps, the cross-entropy.  capdec_train_step.
#include "train.h"

namespace capdec {

// ---------------------------------------------------------------------------------------------- elementwise
__device__ __forceinline__ float gelu_new_f(float x) {
    const float c = 0.7978845608028654f;
    return 0.5f * x * (1.0f + tanhf(c * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ float gelu_new_grad_f(float x) {
    const float c = 0.7978845608028654f;
    const float t = tanhf(c * (x + 0.044715f * x * x * x));
    return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * c * (1.0f + 3.0f * 0.044715f * x * x);
}
__global__ void gelu_new_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = gelu_new_f(x[i]);
}
// dx = dy * gelu_new'(x)   (in place on dy allowed)
__global__ void gelu_new_bwd_kernel(const float *__restrict__ x, const float *dy, float *dx, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dx[i] = dy[i] * gelu_new_grad_f(x[i]);
}
// ---- GPT-2's dropouts (scope 1).  Keep-masks: one byte per element (1 = keep), all sites of one step in one buffer in
// the call order of transformers' GPT2Model: embd [B, S, d], then per block attn [B, H, S, S],
// resid [B, S, d], mlp [B, S, d].  Survivors are scaled by 1 / (1 - p), like torch's dropout.
// Philox4x32-10, key = seed, counter = (4-element group, train step): 4 mask bytes per thread
__global__ void dropout_mask_kernel(uint32_t *__restrict__ mask4, size_t n4, float p, unsigned long long seed,
                                    unsigned long long step) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    uint32_t o[4];
    philox4x32((uint32_t)i, (uint32_t)(i >> 32), (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
    uint32_t m = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e)      // torch: keep = rand >= p, rand = 24 random bits / 2^24
        m |= ((float)(o[e] >> 8) * (1.0f / 16777216.0f) >= p ? 1u : 0u) << (8 * e);
    mask4[i] = m;
}
__device__ __forceinline__ float4 mask4f(uint32_t m, float inv_keep) {
    return make_float4((m & 0xffu) ? inv_keep : 0.f, (m & 0xff00u) ? inv_keep : 0.f, (m & 0xff0000u) ? inv_keep : 0.f,
                       (m & 0xff000000u) ? inv_keep : 0.f);
}
// out = (resid ? resid : 0) + y * mask / keep   (y == out allowed: the embedding dropout; resid: the block's residual
// branch h + dropout(conv1d(...)); backward through a dropout: resid = nullptr)
__global__ void dropout_apply_kernel(const float4 *y, const uint32_t *__restrict__ mask4, const float4 *__restrict__ resid,
                                     float4 *out, size_t n4, float inv_keep) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 v = y[i], k = mask4f(mask4[i], inv_keep);
    float4 r = make_float4(v.x * k.x, v.y * k.y, v.z * k.z, v.w * k.w);
    if (resid) { const float4 a = resid[i]; r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w; }
    out[i] = r;
}
// Row maps of the step's sequences (S = P + L positions per sample, d4 = d / 4 float4 per row):
// embeds[(b, s)] = s < P ? pe[b, s] : wte[tokens[b, s - P]]
__global__ void build_embeds_kernel(const float *__restrict__ pe, const float *__restrict__ wte, const int *__restrict__ tokens,
                                    float *__restrict__ emb, int B, int P, int L, int d4, int V) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int S = P + L;
    if (i >= (size_t)B * S * d4) return;
    const int c = (int)(i % d4), s_ = (int)((i / d4) % S), b = (int)(i / ((size_t)d4 * S));
    int tok = s_ < P ? 0 : tokens[(size_t)b * L + (s_ - P)];
    if (tok < 0 || tok >= V) tok = 0;            // (an id outside the table: the step is flagged bad by ce_finish_kernel)
    const float4 *src = s_ < P ? reinterpret_cast<const float4 *>(pe) + ((size_t)b * P + s_) * d4
                               : reinterpret_cast<const float4 *>(wte) + (size_t)tok * d4;
    reinterpret_cast<float4 *>(emb)[i] = src[c];
}
// the rows the loss reads, logits[:, P-1:-1]: out[(b, t)] = in[(b, P - 1 + t)]
__global__ void take_loss_rows_kernel(const float *__restrict__ in, float *__restrict__ out, int B, int P, int L, int d4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * L * d4) return;
    const int c = (int)(i % d4), t_ = (int)((i / d4) % L), b = (int)(i / ((size_t)d4 * L));
    reinterpret_cast<float4 *>(out)[i] = reinterpret_cast<const float4 *>(in)[((size_t)b * (P + L) + P - 1 + t_) * d4 + c];
}
// its transpose: out[(b, s)] = (P - 1 <= s < P - 1 + L) ? in[(b, s - P + 1)] : 0
__global__ void put_loss_rows_kernel(const float *__restrict__ in, float *__restrict__ out, int B, int P, int L, int d4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int S = P + L;
    if (i >= (size_t)B * S * d4) return;
    const int c = (int)(i % d4), s_ = (int)((i / d4) % S), b = (int)(i / ((size_t)d4 * S));
    const int t_ = s_ - (P - 1);
    reinterpret_cast<float4 *>(out)[i] = (t_ >= 0 && t_ < L) ? reinterpret_cast<const float4 *>(in)[((size_t)b * L + t_) * d4 + c]
                                                             : make_float4(0.f, 0.f, 0.f, 0.f);
}
// the token lookup's share of the tied wte gradient: g_wte[tokens[b, t], :] += d embeds[(b, P + t), :]   (atomic: an id may repeat)
__global__ void embed_scatter_add_kernel(const float *__restrict__ dh, const int *__restrict__ tokens, float *__restrict__ gwte,
                                         int B, int P, int L, int d, int V) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * L * d) return;
    const int c = (int)(i % d), t_ = (int)((i / d) % L), b = (int)(i / ((size_t)d * L));
    const int tok = tokens[(size_t)b * L + t_];
    if (tok < 0 || tok >= V) return;
    atomicAdd(gwte + (size_t)tok * d + c, dh[((size_t)b * (P + L) + P + t_) * d + c]);
}
// g_wpe[s, :] = sum_b d h_0[(b, s), :] for s < S   (the rest of the table gets no gradient: the arena is zeroed)
__global__ void wpe_grad_kernel(const float *__restrict__ dh, float *__restrict__ gwpe, int B, int S, int d) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)S * d) return;
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dh[(size_t)b * S * d + i];
    gwpe[i] = a;
}
// d pe[b, p] = d embeds[(b, p)]   (the mapper's output gradient, un-normalised like everything in the backward pass)
__global__ void take_prefix_grad_kernel(const float *__restrict__ dh, float *__restrict__ dy, int B, int P, int L, int d) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * P * d) return;
    const int c = (int)(i % d), p_ = (int)((i / d) % P), b = (int)(i / ((size_t)d * P));
    dy[i] = dh[((size_t)b * (P + L) + p_) * d + c];
}

// ---------------------------------------------------------------------------------------------- cross-entropy
// logits [rows, ld] (columns >= V are padding) -> in place: (softmax - onehot) x ls for rows whose label != ignore, 0 for
// the others and for the padding; row_loss[r] = lse - logit[label] (0 for ignored rows).  One block per row.
__global__ __launch_bounds__(256) void ce_bwd_kernel(float *__restrict__ logits, int ld, const int *__restrict__ labels,
                                                     int V, int ignore_index, float *__restrict__ row_loss, float ls) {
    __shared__ float red[4];
    const int row = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float *lr = logits + (size_t)row * ld;
    const int lab = labels[row];
    const bool ignored = lab == ignore_index || lab < 0 || lab >= V;      // (an out-of-range label cannot be scored)
    if (ignored) {
        for (int c = t; c < ld; c += 256) lr[c] = 0.f;
        if (t == 0) row_loss[row] = 0.f;
        return;
    }
    float mx = -INFINITY;
    for (int c = t; c < V; c += 256) mx = fmaxf(mx, lr[c]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = t; c < V; c += 256) s += expf(lr[c] - mx);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float lse = mx + logf((red[0] + red[1]) + (red[2] + red[3]));
    if (t == 0) row_loss[row] = lse - lr[lab];
    __syncthreads();                                                       // (lr[lab] is read before anyone rewrites it)
    for (int c = t; c < ld; c += 256) lr[c] = c < V ? (expf(lr[c] - lse) - (c == lab ? 1.f : 0.f)) * ls : 0.f;
}
// the step's scalars: count = labels != ignore (and in range), loss = sum(row_loss) / count, gscale = 1 / (count ls);
// an id outside [0, V) (torch raises on it: the embedding lookup and the loss both index with it) flags the step bad --
// loss NaN, no update   (one block)
__global__ __launch_bounds__(256) void ce_finish_kernel(const float *__restrict__ row_loss, const int *__restrict__ labels,
                                                        int rows, int V, int ignore_index, StepScalars *__restrict__ sc, float ls) {
    __shared__ float rs[4];
    __shared__ int rc[4], rb[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float s = 0.f;
    int n = 0, bad = 0;
    for (int r = t; r < rows; r += 256) {
        const int lab = labels[r];
        if (lab < 0 || lab >= V) bad = 1;
        else if (lab != ignore_index) { s += row_loss[r]; ++n; }
    }
    s = wave_sum(s);
    const float nf = wave_sum((float)n), bf = wave_sum((float)bad);
    if (lane == 0) { rs[wave] = s; rc[wave] = (int)nf; rb[wave] = bf > 0.f; }
    __syncthreads();
    if (t == 0) {
        const int c = rc[0] + rc[1] + rc[2] + rc[3];
        const bool b = rb[0] | rb[1] | rb[2] | rb[3];
        const float loss = b ? __builtin_nanf("") : ((rs[0] + rs[1]) + (rs[2] + rs[3])) / (float)max(c, 1);
        sc->count = c;
        sc->loss = loss;
        sc->gscale = 1.0f / ((float)max(c, 1) * ls);
        sc->bad = b;
        if (!b) {               // (a flagged step neither updates nor counts: the running mean stays a mean of real losses)
            sc->loss_sum += loss;
            sc->loss_steps += 1;
        }
    }
}

// the keep-masks of this step: injected ones (consumed once) or the Philox stream of (seed, step)
static int prepare_dropout_masks(capdec_ctx *c, TrainState &t, size_t n) {
    const size_t n4 = (n + 3) / 4;
    CAPDEC_TRY(t.dmask.ensure(n4 * 4));
    if (t.dinj_n) {
        CAPDEC_CHECK(t.dinj_n == n, "train_step: the injected dropout masks do not have this batch's size "
                                     "(B S d + n_layer (B H S S + 2 B S d) bytes, S = prefix_length + length)");
        CAPDEC_HIP(hipMemcpyAsync(t.dmask.p, t.dinj.p, n, hipMemcpyDeviceToDevice, c->stream));
        t.dinj_n = 0;
    } else {
        hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, c->stream, t.dmask.as<uint32_t>(), n4,
                           c->train_drop_p, c->train_drop_seed, t.draws);
        CAPDEC_HIP(hipGetLastError());
        t.draws += 1;
    }
    t.dmask_n = n;
    return 0;
}
static inline void dropout_apply(hipStream_t st, const float *y, const uint8_t *mask, const float *resid, float *out, size_t n,
                                 float inv_keep) {
    hipLaunchKernelGGL(dropout_apply_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const float4 *>(y),
                       reinterpret_cast<const uint32_t *>(mask), reinterpret_cast<const float4 *>(resid),
                       reinterpret_cast<float4 *>(out), n / 4, inv_keep);
}

// the whole step; see capdec.h: capdec_train_step
static int train_step(capdec_ctx *c, const float *prefix, const int *tokens, int B, int L, float lr, float b1, float b2,
                      float eps, float weight_decay, int apply_update, float *loss_host) {
    const Gpt2 &g = c->gpt;
    Mapper &m = c->map;
    CAPDEC_CHECK(g.loaded && (m.kind == 1 || m.kind == 2), "train_step: needs GPT-2 weights and a mapper");
    CAPDEC_CHECK(g.d == 768 && g.d / g.n_head == 64 && m.d == g.d, "train_step: d = 768, head_dim = 64");
    const int d = g.d, P = m.P, S = P + L, R = B * S, Rl = B * L, O = P * d, D = m.D;
    CAPDEC_CHECK(B >= 1 && L >= 1 && S <= 256 && S <= g.n_pos, "train_step: bad batch geometry (prefix_length + L <= 256)");
    CAPDEC_CHECK(D % 32 == 0 && O % 32 == 0, "train_step: mapper dims must be multiples of 32");
    if (!c->train) { c->train = new TrainState(); c->train->train_gpt = c->train_scope != 0; }
    TrainState &t = *c->train;
    CAPDEC_TRY(prepare_backward_weights(c, t));
    CAPDEC_TRY(build_slots(c, t));
    hipStream_t st = c->stream;
    const int nl = g.n_layer, Vp = t.Vp;
    const size_t Rd = (size_t)R * d;
    const bool full = t.train_gpt;                  // GPT-2 is trained too: weight gradients along the way
    // GPT-2's dropouts exist only when GPT-2 is in train() mode: the frozen scope keeps it in eval mode (train.py:283-287)
    const bool drop = full && (c->train_drop_p > 0.f || t.dinj_n);
    const float keep = 1.0f - c->train_drop_p, inv_keep = 1.0f / keep;
    const size_t mA = (size_t)B * g.n_head * S * S, mLayer = mA + 2 * Rd;       // mask stream: embd, then per layer attn, resid, mlp
    CAPDEC_CHECK(!t.dinj_n || full, "train_step: dropout masks were injected but GPT-2 is frozen (scope 0: eval mode, no dropout)");
    CAPDEC_CHECK(!t.dinj_n || c->train_drop_p > 0.f, "train_step: dropout masks were injected but the dropout probability is 0");
    // loss scale of the backward pass: a power of two that lifts softmax probabilities of a 50 257-entry vocabulary into
    // the two-fp16-plane format's full-precision range (>= 2^-14); exact, undone by StepScalars::gscale
    const float LS = c->tune.train_f16x2 ? 64.f : 1.f;
    // ---- buffers
    CAPDEC_TRY(t.pe.ensure((size_t)B * O * 4));
    CAPDEC_TRY(t.emb.ensure(Rd * 4));
    CAPDEC_TRY(t.hs.ensure(Rd * 4 * (nl + 1)));              // block inputs h_0 .. h_nl
    CAPDEC_TRY(t.a.ensure(Rd * 4));
    CAPDEC_TRY(t.qkv.ensure(Rd * 3 * 4 * nl));
    CAPDEC_TRY(t.att.ensure(Rd * 4 * nl));
    CAPDEC_TRY(t.hmid.ensure(Rd * 4 * nl));
    CAPDEC_TRY(t.fc.ensure(Rd * 4 * 4 * nl));
    CAPDEC_TRY(t.gl.ensure(Rd * 4 * 4));
    CAPDEC_TRY(t.hf.ensure(Rd * 4));
    CAPDEC_TRY(t.hfl.ensure((size_t)Rl * d * 4));
    CAPDEC_TRY(t.logits.ensure((size_t)Rl * Vp * 4));
    CAPDEC_TRY(t.rloss.ensure((size_t)Rl * 4));
    CAPDEC_TRY(t.cnt.ensure(sizeof(StepScalars)));
    CAPDEC_TRY(t.dh.ensure(Rd * 4));
    CAPDEC_TRY(t.dh2.ensure(Rd * 4));
    CAPDEC_TRY(t.da.ensure(Rd * 4));
    CAPDEC_TRY(t.dqkv.ensure(Rd * 3 * 4));
    CAPDEC_TRY(t.datt.ensure(Rd * 4));
    CAPDEC_TRY(t.dfc.ensure(Rd * 4 * 4));
    CAPDEC_TRY(t.dhfl.ensure((size_t)Rl * d * 4));
    CAPDEC_TRY(t.lse.ensure((size_t)B * g.n_head * S * 4));
    CAPDEC_TRY(t.dsum.ensure((size_t)B * g.n_head * S * 4));
    CAPDEC_TRY(t.dy.ensure((size_t)B * O * 4));
    if (drop) {
        CAPDEC_TRY(t.ytmp.ensure(Rd * 4));
        CAPDEC_TRY(t.dtmp.ensure(Rd * 4));
        CAPDEC_TRY(prepare_dropout_masks(c, t, Rd + (size_t)nl * mLayer));
    }
    if (!t.scalars_ready) {
        CAPDEC_HIP(hipMemsetAsync(t.cnt.p, 0, sizeof(StepScalars), st));
        t.scalars_ready = true;
    }
    float *pe = t.pe.as<float>(), *emb = t.emb.as<float>(), *hs = t.hs.as<float>(), *a = t.a.as<float>(), *gl = t.gl.as<float>(),
          *hf = t.hf.as<float>(), *hfl = t.hfl.as<float>(), *logits = t.logits.as<float>();
    StepScalars *sc = t.cnt.as<StepScalars>();
    const uint8_t *mk = t.dmask.as<uint8_t>();
    float *ytmp = t.ytmp.as<float>(), *dtmp = t.dtmp.as<float>();

    // ---- forward: mapper, then embeds = cat(pe.view(B, P, d), wte(tokens))
    CAPDEC_TRY(mapper_forward_saved(c, t, prefix, B, pe));
    {
        ProfScope ps(c, F_EMBED);
        hipLaunchKernelGGL(build_embeds_kernel, grid1(Rd / 4), dim3(256), 0, st, pe, g.wte, tokens, emb, B, P, L, d / 4, g.vocab);
        CAPDEC_TRY(launch_embed_prefix(st, emb, g.wpe, hs, B, S, 0, d));
        if (drop) dropout_apply(st, hs, mk, nullptr, hs, Rd, inv_keep);                 // self.drop(inputs_embeds + position_embeds)
    }
    KvCache kv;
    kv_geometry(kv, B, S, g.n_head, 64);
    kv.tune = &c->tune;
    for (int i = 0; i < nl; ++i) {
        const Gpt2Layer &w = g.layers[i];
        float *h = hs + Rd * i, *hn = hs + Rd * (i + 1);
        float *qkv = t.qkv.as<float>() + Rd * 3 * i, *att = t.att.as<float>() + Rd * i, *hmid = t.hmid.as<float>() + Rd * i,
              *fc = t.fc.as<float>() + Rd * 4 * i;
        const uint8_t *m_att = mk + Rd + (size_t)i * mLayer, *m_res = m_att + mA, *m_mlp = m_res + Rd;
        { ProfScope ps(c, F_LN); CAPDEC_TRY(launch_layernorm(st, h, d, w.ln1w, w.ln1b, g.eps, a, d, R, d)); }
        CAPDEC_TRY(gemm(c, a, d, w.wqkv, d, qkv, 3 * d, R, 3 * d, d, w.bqkv, CAPDEC_ACT_NONE, nullptr, 0, !full));
        if (drop) {
            ProfScope ps(c, F_ATTN_PRE);
            CAPDEC_TRY(train_attn_fwd(c, qkv, att, B, S, g.n_head, 64, true, 0.125f, m_att, inv_keep));
            CAPDEC_TRY(gemm(c, att, d, w.wproj, d, ytmp, d, R, d, d, w.bproj, CAPDEC_ACT_NONE, nullptr, 0, false));
            dropout_apply(st, ytmp, m_res, h, hmid, Rd, inv_keep);                      // h + resid_dropout(c_proj(att))
        } else {
            { ProfScope ps(c, F_ATTN_PRE); CAPDEC_TRY(launch_attn_prefill(st, qkv, kv, i, B, S, 1, att, true)); }
            CAPDEC_TRY(gemm(c, att, d, w.wproj, d, hmid, d, R, d, d, w.bproj, CAPDEC_ACT_NONE, h, d, !full));
        }
        { ProfScope ps(c, F_LN); CAPDEC_TRY(launch_layernorm(st, hmid, d, w.ln2w, w.ln2b, g.eps, a, d, R, d)); }
        CAPDEC_TRY(gemm(c, a, d, w.wfc, d, fc, 4 * d, R, 4 * d, d, w.bfc, CAPDEC_ACT_NONE, nullptr, 0, !full));
        hipLaunchKernelGGL(gelu_new_fwd_kernel, grid1(Rd * 4), dim3(256), 0, st, fc, gl, Rd * 4);
        if (drop) {
            CAPDEC_TRY(gemm(c, gl, 4 * d, w.wproj2, 4 * d, ytmp, d, R, d, 4 * d, w.bproj2, CAPDEC_ACT_NONE, nullptr, 0, false));
            dropout_apply(st, ytmp, m_mlp, hmid, hn, Rd, inv_keep);                     // h_mid + dropout(mlp.c_proj(...))
        } else
            CAPDEC_TRY(gemm(c, gl, 4 * d, w.wproj2, 4 * d, hn, d, R, d, 4 * d, w.bproj2, CAPDEC_ACT_NONE, hmid, d, !full));
    }
    float *hL = hs + Rd * nl;
    { ProfScope ps(c, F_LN); CAPDEC_TRY(launch_layernorm(st, hL, d, g.lnfw, g.lnfb, g.eps, hf, d, R, d)); }
    // the rows the loss reads: logits[:, P-1:-1]  ->  row (b, P - 1 + t) predicts tokens[b, t]
    hipLaunchKernelGGL(take_loss_rows_kernel, grid1((size_t)Rl * (d / 4)), dim3(256), 0, st, hf, hfl, B, P, L, d / 4);
    CAPDEC_HIP(hipMemsetAsync(logits, 0, (size_t)Rl * Vp * 4, st));
    CAPDEC_TRY(gemm(c, hfl, d, g.wte, d, logits, Vp, Rl, g.vocab, d, nullptr, CAPDEC_ACT_NONE, nullptr, 0, !full));
    // ---- loss + d logits (un-normalised: (softmax - onehot) LS; the factor 1 / (count LS) is applied where gradients are
    // consumed, so every backward GEMM sees operands of order one)
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(Rl), dim3(256), 0, st, logits, Vp, tokens, g.vocab, 0, t.rloss.as<float>(), LS);
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(256), 0, st, t.rloss.as<float>(), tokens, Rl, g.vocab, 0, sc, LS);
    CAPDEC_HIP(hipGetLastError());
    // ---- backward through the lm_head and ln_f
    float *dh = t.dh.as<float>(), *dh2 = t.dh2.as<float>(), *da = t.da.as<float>(), *dqkv = t.dqkv.as<float>(),
          *datt = t.datt.as<float>(), *dfc = t.dfc.as<float>(), *dhfl = t.dhfl.as<float>();
    const int gs = t.gpt_slot0;
    CAPDEC_HIP(hipMemsetAsync(t.G.p, 0, t.n_params * 4, st));        // (bias / LayerNorm / wte gradients are accumulated)
    if (full) {
        // the lm_head's share of the tied wte: d logits^T hf  ([V, d]; K = the loss rows)
        if (c->tune.train_f16x2) {
            CAPDEC_TRY(gemm_tn(c, logits, Vp, hfl, d, Rl, g.vocab, d, t.grad(gs), d));      // (logits rows are padded to Vp columns)
        } else {
            const int Kp = pad_rows(c, Rl);
            CAPDEC_TRY(t.tA.ensure((size_t)Vp * Kp * 4));
            CAPDEC_TRY(t.tB.ensure((size_t)d * Kp * 4));
            CAPDEC_TRY(transpose_pad(c, logits, Rl, Vp, t.tA.as<float>(), Kp));
            CAPDEC_TRY(transpose_pad(c, hfl, Rl, d, t.tB.as<float>(), Kp));
            CAPDEC_TRY(gemm_fp32(c, t.tA.as<float>(), Kp, t.tB.as<float>(), Kp, t.grad(gs), d, g.vocab, d, Kp));
        }
    }
    CAPDEC_TRY(gemm_fp32(c, logits, Vp, t.wte_t, Vp, dhfl, d, Rl, d, Vp, !full));
    hipLaunchKernelGGL(put_loss_rows_kernel, grid1(Rd / 4), dim3(256), 0, st, dhfl, da, B, P, L, d / 4);
    CAPDEC_TRY(ln_bwd(c, hL, g.lnfw, da, nullptr, dh, R, d, g.eps, full ? t.grad(gs + 2 + 12 * nl) : nullptr,
                      full ? t.grad(gs + 3 + 12 * nl) : nullptr));
    // ---- backward through the blocks (frozen scope: dX only)
    for (int i = nl - 1; i >= 0; --i) {
        const Gpt2Layer &w = g.layers[i];
        const TrainState::LayerT &wt = t.lt[i];
        float *h = hs + Rd * i;
        float *qkv = t.qkv.as<float>() + Rd * 3 * i, *hmid = t.hmid.as<float>() + Rd * i, *fc = t.fc.as<float>() + Rd * 4 * i;
        const uint8_t *m_att = mk + Rd + (size_t)i * mLayer, *m_res = m_att + mA, *m_mlp = m_res + Rd;
        const int s0 = gs + 2 + 12 * i;               // (full scope) this layer's slots: ln_1 w b, c_attn w b, c_proj w b, ln_2 w b, c_fc w b, mlp.c_proj w b
        const float *dy2 = dh;                        // d (mlp.c_proj output): dh through the MLP's dropout
        if (drop) { dropout_apply(st, dh, m_mlp, nullptr, dtmp, Rd, inv_keep); dy2 = dtmp; }
        if (full) {                                   // mlp.c_proj: y = gelu(fc) W + b
            hipLaunchKernelGGL(gelu_new_fwd_kernel, grid1(Rd * 4), dim3(256), 0, st, fc, gl, Rd * 4);
            CAPDEC_TRY(linear_dw(c, t, dy2, gl, R, d, 4 * d, t.grad(s0 + 10), t.grad(s0 + 11)));
        }
        CAPDEC_TRY(gemm_fp32(c, dy2, d, wt.wproj2_t, d, dfc, 4 * d, R, 4 * d, d, !full));         // d gelu_out = dy2 Wproj2^T
        hipLaunchKernelGGL(gelu_new_bwd_kernel, grid1(Rd * 4), dim3(256), 0, st, fc, dfc, dfc, Rd * 4);
        if (full) {                                   // mlp.c_fc: input ln_2(h_mid)
            CAPDEC_TRY(launch_layernorm(st, hmid, d, w.ln2w, w.ln2b, g.eps, a, d, R, d));
            CAPDEC_TRY(linear_dw(c, t, dfc, a, R, 4 * d, d, t.grad(s0 + 8), t.grad(s0 + 9)));
        }
        CAPDEC_TRY(gemm_fp32(c, dfc, 4 * d, wt.wfc_t, 4 * d, da, d, R, d, 4 * d, !full));         // d a2
        CAPDEC_TRY(ln_bwd(c, hmid, w.ln2w, da, dh, dh2, R, d, g.eps, full ? t.grad(s0 + 6) : nullptr,
                          full ? t.grad(s0 + 7) : nullptr));                                       // dh_mid = dh + LN'(..)
        const float *dy1 = dh2;                       // d (attn.c_proj output): dh_mid through resid_dropout
        if (drop) { dropout_apply(st, dh2, m_res, nullptr, dtmp, Rd, inv_keep); dy1 = dtmp; }
        if (full) CAPDEC_TRY(linear_dw(c, t, dy1, t.att.as<float>() + Rd * i, R, d, d, t.grad(s0 + 4), t.grad(s0 + 5)));   // attn.c_proj
        CAPDEC_TRY(gemm_fp32(c, dy1, d, wt.wproj_t, d, datt, d, R, d, d, !full));                 // d att
        CAPDEC_TRY(train_attn_bwd(c, t, qkv, datt, dqkv, B, S, g.n_head, 64, true, 0.125f, drop ? m_att : nullptr, inv_keep));
        if (full) {                                   // attn.c_attn: input ln_1(h)
            CAPDEC_TRY(launch_layernorm(st, h, d, w.ln1w, w.ln1b, g.eps, a, d, R, d));
            CAPDEC_TRY(linear_dw(c, t, dqkv, a, R, 3 * d, d, t.grad(s0 + 2), t.grad(s0 + 3)));
        }
        CAPDEC_TRY(gemm_fp32(c, dqkv, 3 * d, wt.wqkv_t, 3 * d, da, d, R, d, 3 * d, !full));       // d a1
        CAPDEC_TRY(ln_bwd(c, h, w.ln1w, da, dh2, dh, R, d, g.eps, full ? t.grad(s0 + 0) : nullptr,
                          full ? t.grad(s0 + 1) : nullptr));                                       // dh = dh_mid + LN'(..)
    }
    if (drop) dropout_apply(st, dh, mk, nullptr, dh, Rd, inv_keep);                                // through the embedding dropout
    if (full) {       // d inputs_embeds: the token rows feed the tied wte (added to the lm_head's share), every row feeds wpe
        hipLaunchKernelGGL(embed_scatter_add_kernel, grid1((size_t)Rl * d), dim3(256), 0, st, dh, tokens, t.grad(gs), B, P, L, d, g.vocab);
        hipLaunchKernelGGL(wpe_grad_kernel, grid1((size_t)S * d), dim3(256), 0, st, dh, t.grad(gs + 1), B, S, d);
    }
    CAPDEC_HIP(hipGetLastError());
    // ---- the mapper: dY = d embeds[:, :P]
    float *dy = t.dy.as<float>();
    hipLaunchKernelGGL(take_prefix_grad_kernel, grid1((size_t)B * O), dim3(256), 0, st, dh, dy, B, P, L, d);
    CAPDEC_TRY(mapper_backward(c, t, prefix, dy, B));
    CAPDEC_HIP(hipGetLastError());
    t.have_grads = true;
    // ---- AdamW (transformers 4.24 semantics) on arena x gscale
    if (apply_update) CAPDEC_TRY(train_apply_update(c, t, lr, b1, b2, eps, weight_decay));
    // loss == nullptr: nothing waits for the device (capdec_train_loss reads the step's loss, and their running sum, later)
    if (loss_host) {
        CAPDEC_HIP(hipMemcpyAsync(loss_host, &sc->loss, sizeof(float), hipMemcpyDeviceToHost, st));
        CAPDEC_HIP(hipStreamSynchronize(st));
    }
    return 0;
}

}  // namespace capdec

using namespace capdec;

extern "C" {

int capdec_train_step(capdec_ctx *c, const float *d_prefix, const int32_t *d_tokens, int batch, int length, float lr,
                      float beta1, float beta2, float eps, float weight_decay, int apply_update, float *loss) {
    CAPDEC_CHECK(c && d_prefix && d_tokens, "train_step: null argument");
    CAPDEC_HIP(hipSetDevice(c->device));
    return train_step(c, d_prefix, d_tokens, batch, length, lr, beta1, beta2, eps, weight_decay, apply_update, loss);
}

}  // extern "C"
