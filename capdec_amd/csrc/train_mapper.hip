// Train step, the mapping networks (train.h): the MLP (reference gpt2_prefix.py:114-126) and the TransformerMapper
// (transformer_mapper.py:113-127) -- forward keeping what the backward needs, backward into the gradient arena.
#include "train.h"

namespace capdec {

// dx = dy * (1 - y^2), y = tanh(.)
__global__ void tanh_bwd_kernel(const float *__restrict__ y, const float *dy, float *dx, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dx[i] = dy[i] * (1.0f - y[i] * y[i]);
}
// dx = dy where y > 0 (y = relu(.)), else 0
__global__ void relu_bwd_kernel(const float *__restrict__ y, const float *dy, float *dx, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}
// TransformerMapper output = rows clip_len.. of the sequence: dseq[b, s] = s >= clip_len ? dout[b, s - clip_len] : 0
__global__ void tmapper_put_kernel(const float *__restrict__ dout, float *__restrict__ dseq, int n, int clip_len, int P, int d) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int S = clip_len + P;
    if (i >= (size_t)n * S * d) return;
    const int c = (int)(i % d), s_ = (int)((i / d) % S), b = (int)(i / ((size_t)d * S));
    dseq[i] = s_ >= clip_len ? dout[((size_t)b * P + (s_ - clip_len)) * d + c] : 0.f;
}
// the sequence's first layer input = cat(linear(x).view(B, clip_len, d), prefix_const): dlin[b, s, :] = dseq[b, s < clip_len],
// g_prefix_const[p, :] = sum_b dseq[b, clip_len + p, :]
__global__ void tmapper_split_grad_kernel(const float *__restrict__ dseq, float *__restrict__ dlin, float *__restrict__ gpc,
                                          int n, int clip_len, int P, int d) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int S = clip_len + P;
    if (i < (size_t)n * clip_len * d) {
        const int c = (int)(i % d), s_ = (int)((i / d) % clip_len), b = (int)(i / ((size_t)d * clip_len));
        dlin[i] = dseq[((size_t)b * S + s_) * d + c];
    }
    if (i < (size_t)P * d) {
        const int c = (int)(i % d), pp = (int)(i / d);
        float a = 0.f;
        for (int b = 0; b < n; ++b) a += dseq[((size_t)b * S + clip_len + pp) * d + c];
        gpc[i] = a;
    }
}

// ---- the mapper's forward with everything its backward needs; out = pe [B, P d]
int mapper_forward_saved(capdec_ctx *c, TrainState &t, const float *x, int B, float *pe) {
    Mapper &m = c->map;
    const int d = m.d, D = m.D, O = m.P * d;
    hipStream_t st = c->stream;
    if (m.kind == 1) {
        const int H = m.hidden;
        CAPDEC_TRY(t.hid.ensure((size_t)B * H * 4));
        // (current weights: never the cached planes of an earlier step)
        CAPDEC_TRY(gemm(c, x, D, m.w1, D, t.hid.as<float>(), H, B, H, D, m.b1, CAPDEC_ACT_TANH, nullptr, 0, false));
        return gemm(c, t.hid.as<float>(), H, m.w2, H, pe, O, B, O, H, m.b2, CAPDEC_ACT_NONE, nullptr, 0, false);
    }
    const int S = m.clip_len + m.P, M = B * S, hd = d / m.heads, hid = m.mlp_hidden, nl = m.n_layers;
    const size_t Md = (size_t)M * d;
    CAPDEC_TRY(t.t_lin.ensure((size_t)B * m.clip_len * d * 4));
    CAPDEC_TRY(t.t_seq.ensure(Md * 4 * (nl + 1)));
    CAPDEC_TRY(t.t_a1.ensure(Md * 4 * nl));
    CAPDEC_TRY(t.t_qkv.ensure(Md * 3 * 4 * nl));
    CAPDEC_TRY(t.t_att.ensure(Md * 4 * nl));
    CAPDEC_TRY(t.t_mid.ensure(Md * 4 * nl));
    CAPDEC_TRY(t.t_a2.ensure(Md * 4 * nl));
    CAPDEC_TRY(t.t_r.ensure((size_t)M * hid * 4 * nl));
    float *seq = t.t_seq.as<float>();
    CAPDEC_TRY(gemm(c, x, D, m.lin_w, D, t.t_lin.as<float>(), m.clip_len * d, B, m.clip_len * d, D, m.lin_b, CAPDEC_ACT_NONE,
                    nullptr, 0, false));
    { ProfScope ps(c, F_OTHER); CAPDEC_TRY(launch_tmapper_concat(st, t.t_lin.as<float>(), m.prefix_const, seq, B, m.clip_len, m.P, d)); }
    for (int l = 0; l < nl; ++l) {
        const TMapLayer &w = m.layers[l];
        float *h = seq + Md * l, *hn = seq + Md * (l + 1), *a1 = t.t_a1.as<float>() + Md * l, *qkv = t.t_qkv.as<float>() + Md * 3 * l,
              *att = t.t_att.as<float>() + Md * l, *mid = t.t_mid.as<float>() + Md * l, *a2 = t.t_a2.as<float>() + Md * l,
              *r = t.t_r.as<float>() + (size_t)M * hid * l;
        { ProfScope ps(c, F_LN); CAPDEC_TRY(launch_layernorm(st, h, d, w.n1w, w.n1b, 1e-5f, a1, d, M, d)); }
        CAPDEC_TRY(gemm(c, a1, d, w.wqkv, d, qkv, 3 * d, M, 3 * d, d, nullptr, CAPDEC_ACT_NONE, nullptr, 0, false));
        {
            ProfScope ps(c, F_MAP_ATTN);
            if (hd == 96)       // (block-per-(sample, head) kernel when the head fits the LDS; the inference kernel otherwise)
                CAPDEC_TRY(train_attn_fwd(c, qkv, att, B, S, m.heads, 96, false, (float)pow(96.0, -0.5), nullptr, 1.f));
            else
                CAPDEC_TRY(launch_attn_mapper(st, qkv, 3 * d, qkv + d, qkv + 2 * d, 3 * d, att, B, S, m.heads, hd));
        }
        CAPDEC_TRY(gemm(c, att, d, w.wproj, d, mid, d, M, d, d, w.bproj, CAPDEC_ACT_NONE, h, d, false));
        { ProfScope ps(c, F_LN); CAPDEC_TRY(launch_layernorm(st, mid, d, w.n2w, w.n2b, 1e-5f, a2, d, M, d)); }
        CAPDEC_TRY(gemm(c, a2, d, w.wfc1, d, r, hid, M, hid, d, w.bfc1, CAPDEC_ACT_RELU, nullptr, 0, false));
        CAPDEC_TRY(gemm(c, r, hid, w.wfc2, hid, hn, d, M, d, hid, w.bfc2, CAPDEC_ACT_NONE, mid, d, false));
    }
    ProfScope ps(c, F_OTHER);
    return launch_tmapper_take(st, seq + Md * nl, pe, B, m.clip_len, m.P, d);
}

// ---- the mapper's backward: dy [B, P d] = d loss / d pe  ->  every slot's gradient in the arena G
int mapper_backward(capdec_ctx *c, TrainState &t, const float *x, const float *dy, int B) {
    Mapper &m = c->map;
    const int d = m.d, D = m.D, O = m.P * d;
    hipStream_t st = c->stream;
    if (m.kind == 1) {
        const int H = m.hidden;
        CAPDEC_TRY(t.dhid.ensure((size_t)B * H * 4));
        float *hid = t.hid.as<float>(), *dhid = t.dhid.as<float>();
        CAPDEC_TRY(linear_dw(c, t, dy, hid, B, O, H, t.grad(2), t.grad(3)));
        CAPDEC_TRY(linear_dx(c, t, dy, m.w2, dhid, B, O, H));
        hipLaunchKernelGGL(tanh_bwd_kernel, grid1((size_t)B * H), dim3(256), 0, st, hid, dhid, dhid, (size_t)B * H);
        return linear_dw(c, t, dhid, x, B, H, D, t.grad(0), t.grad(1));
    }
    const int S = m.clip_len + m.P, M = B * S, hid = m.mlp_hidden, nl = m.n_layers, HD = d / m.heads;
    CAPDEC_CHECK(HD == 96 && d == 768, "train: the TransformerMapper backward is instantiated for d = 768, 8 heads of 96");
    const size_t Md = (size_t)M * d;
    CAPDEC_TRY(t.t_ds.ensure(Md * 4));
    CAPDEC_TRY(t.t_ds2.ensure(Md * 4));
    CAPDEC_TRY(t.t_da.ensure(Md * 4));
    CAPDEC_TRY(t.t_dr.ensure((size_t)M * hid * 4));
    CAPDEC_TRY(t.t_dqkv.ensure(Md * 3 * 4));
    CAPDEC_TRY(t.t_datt.ensure(Md * 4));
    CAPDEC_TRY(t.t_dlin.ensure((size_t)B * m.clip_len * d * 4));
    CAPDEC_TRY(t.lse.ensure((size_t)B * m.heads * S * 4));
    CAPDEC_TRY(t.dsum.ensure((size_t)B * m.heads * S * 4));
    float *ds = t.t_ds.as<float>(), *ds2 = t.t_ds2.as<float>(), *da = t.t_da.as<float>(), *dr = t.t_dr.as<float>(),
          *dqkv = t.t_dqkv.as<float>(), *datt = t.t_datt.as<float>(), *dlin = t.t_dlin.as<float>();
    const float *seq = t.t_seq.as<float>();
    const float scale = (float)pow((double)HD, -0.5);
    hipLaunchKernelGGL(tmapper_put_kernel, grid1(Md), dim3(256), 0, st, dy, ds, B, m.clip_len, m.P, d);
    for (int l = nl - 1; l >= 0; --l) {
        const TMapLayer &w = m.layers[l];
        const int s0 = 3 + 12 * l;
        const float *h = seq + Md * l, *a1 = t.t_a1.as<float>() + Md * l, *qkv = t.t_qkv.as<float>() + Md * 3 * l,
                    *att = t.t_att.as<float>() + Md * l, *mid = t.t_mid.as<float>() + Md * l, *a2 = t.t_a2.as<float>() + Md * l,
                    *r = t.t_r.as<float>() + (size_t)M * hid * l;
        // mlp: out = mid + fc2(relu(fc1(a2)))
        CAPDEC_TRY(linear_dw(c, t, ds, r, M, d, hid, t.grad(s0 + 10), t.grad(s0 + 11)));
        CAPDEC_TRY(linear_dx(c, t, ds, w.wfc2, dr, M, d, hid));
        hipLaunchKernelGGL(relu_bwd_kernel, grid1((size_t)M * hid), dim3(256), 0, st, r, dr, dr, (size_t)M * hid);
        CAPDEC_TRY(linear_dw(c, t, dr, a2, M, hid, d, t.grad(s0 + 8), t.grad(s0 + 9)));
        CAPDEC_TRY(linear_dx(c, t, dr, w.wfc1, da, M, hid, d));
        CAPDEC_TRY(ln_bwd(c, mid, w.n2w, da, ds, ds2, M, d, 1e-5f, t.grad(s0 + 6), t.grad(s0 + 7)));        // ds2 = d mid
        // attention: mid = h + project(att)
        CAPDEC_TRY(linear_dw(c, t, ds2, att, M, d, d, t.grad(s0 + 4), t.grad(s0 + 5)));
        CAPDEC_TRY(linear_dx(c, t, ds2, w.wproj, datt, M, d, d));
        CAPDEC_TRY(train_attn_bwd(c, t, qkv, datt, dqkv, B, S, m.heads, 96, false, scale, nullptr, 1.f));
        CAPDEC_TRY(linear_dw(c, t, dqkv, a1, M, 3 * d, d, t.grad(s0 + 2), nullptr));      // [to_queries ; to_keys_values]: no bias
        CAPDEC_TRY(linear_dx(c, t, dqkv, w.wqkv, da, M, 3 * d, d));
        CAPDEC_TRY(ln_bwd(c, h, w.n1w, da, ds2, ds, M, d, 1e-5f, t.grad(s0 + 0), t.grad(s0 + 1)));          // ds = d h
    }
    hipLaunchKernelGGL(tmapper_split_grad_kernel, grid1(std::max((size_t)B * m.clip_len * d, (size_t)m.P * d)), dim3(256), 0, st,
                       ds, dlin, t.grad(2), B, m.clip_len, m.P, d);
    CAPDEC_HIP(hipGetLastError());
    return linear_dw(c, t, dlin, x, B, m.clip_len * d, D, t.grad(0), t.grad(1));
}

}  // namespace capdec
