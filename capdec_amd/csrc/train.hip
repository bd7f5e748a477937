// The train step -- reference train.py:344-354.  Scope 0: run with --only_prefix (ClipCaptionPrefix, train.py:279-287:
// parameters() = the mapper's, GPT-2 in eval mode => no dropout, a deterministic step).  Scope 1: the reference's DEFAULT
// run (ClipCaptionModel, :306-308: GPT-2 is trained too and, being in train() mode, applies transformers' dropouts --
// embd / attention weights / both residual branches, p = 0.1 -- from a Philox keep-mask stream or from injected masks):
//
//     prefix -> MLP mapper -> cat(prefix rows, wte(tokens)) -> GPT-2 -> logits[:, P-1:-1] -> cross_entropy(ignore_index=0)
//     -> backward down to the mapper's four tensors -> transformers-4.24 AdamW
//
// Gradients travel UN-NORMALISED (d logits = (softmax - onehot) x LS, LS a power of two) through every backward GEMM and
// the factor 1 / (count LS) is applied where a gradient is consumed (AdamW, capdec_train_get): the GEMM operands then sit
// in the range where the two-fp16-plane format is fp32-accurate, whatever the number of scored labels.
//
// Both mapping networks: the MLP (gpt2_prefix.py:114-126) and the TransformerMapper (transformer_mapper.py:113-127: every
// one of its 3 + 12 n_layers tensors).  Structure: the forward keeps every activation the backward needs
// (fp32, per layer: block input, qkv, attention output, mid-block residual, c_fc pre-activation: 30 KB per token and
// layer -- 1.2 GB for the reference's default batch of 34 captions x (40 + ~20) positions x 12 layers); the backward is
// dX-only through GPT-2 (its weights are frozen: no weight gradients, no optimizer state for 124 M parameters) on the
// NATIVE fp32 MFMA GEMM (launch_gemm_f32: gradients span many binades, the two-fp16-plane format of the inference path is
// only fp32-accurate above 2^-14) against transposed copies of the weights made once on the device; LayerNorm / GELU /
// tanh / attention / cross-entropy backward are small HBM-bound kernels below; the mapper's weight gradients are
// dY^T X products with K = batch (the same GEMM on transposed, zero-padded activations), and AdamW is one elementwise
// pass over (p, g, m, v).  Parity: tests/test_hip_parity.py against gradients the reference's own loss.backward()
// produced (tests/golden/train_step_*.npz).
#include "context.h"

#include <algorithm>
#include <cmath>
#include <vector>

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
// causal attention with dropout on the softmax weights (GPT-2 in train() mode: attn_dropout): one wavefront per
// (sample, head, query i), HD = 64: lane = head dimension; out_i = sum_j softmax_j(q_i . k_j / 8) mask_ij / keep v_j
__global__ __launch_bounds__(256) void attn_fwd_drop_kernel(const float *__restrict__ qkv, const uint8_t *__restrict__ mask,
                                                            float *__restrict__ out, int total, int S, int heads,
                                                            float scale, float inv_keep) {
    extern __shared__ float sh[];                     // [4 waves][S]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave;
    if (gw >= total) return;
    const int i = gw % S, bh = gw / S, h = bh % heads, b = bh / heads;
    const int d = heads * 64;
    float *sc = sh + (size_t)wave * S;
    const size_t row = (size_t)b * S + i;
    const float q = qkv[row * 3 * d + h * 64 + lane];
    for (int j = 0; j <= i; ++j) {
        const float a = wave_sum(q * qkv[((size_t)b * S + j) * 3 * d + d + h * 64 + lane]) * scale;
        if (lane == 0) sc[j] = a;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float mx = -INFINITY;
    for (int j = lane; j <= i; j += 64) mx = fmaxf(mx, sc[j]);
    mx = wave_max(mx);
    float l = 0.f;
    for (int j = lane; j <= i; j += 64) l += expf(sc[j] - mx);
    l = wave_sum(l);
    const float inv = 1.0f / l;
    const uint8_t *mr = mask + ((size_t)bh * S + i) * S;
    float o = 0.f;
    for (int j = 0; j <= i; ++j) {
        const float pj = expf(sc[j] - mx) * inv * (mr[j] ? inv_keep : 0.f);
        o += pj * qkv[((size_t)b * S + j) * 3 * d + 2 * d + h * 64 + lane];
    }
    out[row * d + h * 64 + lane] = o;
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
// ---- full-model scope (GPT-2 trained too) only
// x[i] *= *scale   (capdec_train_get: a gradient leaves the arena normalised)
__global__ void scale_by_kernel(float *x, size_t n, const float *__restrict__ scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= *scale;
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
// dst[c][r] = src[r][c] for r < rows, 0 for rows <= r < ld   (dst [cols][ld])
__global__ void transpose_pad_kernel(const float *__restrict__ src, int rows, int cols, float *__restrict__ dst, int ld) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (c < cols && r < ld) dst[(size_t)c * ld + r] = tile[tx][k];
    }
}
// out[j] += sum over rows of x[r][j]: block (x, y) sums rows [64 y, 64 y + 64) of 256 columns and adds its partial sum
// atomically (`out` is zeroed by the caller: the gradient arena is cleared once per step)
constexpr int COLSUM_ROWS = 64;
__global__ void colsum_kernel(const float *__restrict__ x, int rows, int n, float *__restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int r0 = blockIdx.y * COLSUM_ROWS, r1 = min(rows, r0 + COLSUM_ROWS);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
        s0 += x[(size_t)r * n + j];
        s1 += x[(size_t)(r + 1) * n + j];
        s2 += x[(size_t)(r + 2) * n + j];
        s3 += x[(size_t)(r + 3) * n + j];
    }
    for (; r < r1; ++r) s0 += x[(size_t)r * n + j];
    atomicAdd(out + j, (s0 + s1) + (s2 + s3));
}
// The step's device-side scalars (one 64-byte record; TrainState::cnt)
struct StepScalars {
    int count;              // labels scored by the loss (!= ignore_index)
    float loss;             // mean over them (NaN when an id lies outside the vocabulary)
    float gscale;           // 1 / (max(count, 1) LS): what turns an arena entry into d loss / d tensor
    int bad;                // an id outside [0, vocab): the forward looked row 0 up instead; the update is skipped
    long long updates;      // AdamW updates applied (bias correction uses updates + 1)
    float step_size;        // lr sqrt(1 - b2^t) / (1 - b1^t) of the update being applied
    float loss_sum;         // sum of the losses since the last capdec_train_loss(reset) ...
    int loss_steps;         // ... and how many
};
// one thread: the update's step size from the device's own update counter (a bad step neither counts nor updates)
__global__ void adam_prepare_kernel(StepScalars *s, float lr, float b1, float b2) {
    if (s->bad) return;
    const double t = (double)(s->updates + 1);
    s->step_size = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
    s->updates += 1;
}
// transformers-4.24 AdamW (optimization.py AdamW.step): m, v updated in place; p -= step_size * m / (sqrt(v) + eps);
// then p -= decay * p (decay = lr * weight_decay, 0 by default).  g = arena entry x gscale;
// the same update for EVERY tensor of the scope in one launch (the full model has 247 of them: one launch each cost 1.8 ms
// of a 26 ms step): block b takes chunk b of the table -- (slot, chunk within the slot) -- and walks its 16 384 elements
struct SlotDev { float *p; unsigned long long off, n; };
constexpr int ADAMW_CHUNK = 16384;
__global__ __launch_bounds__(256) void adamw_multi_kernel(const SlotDev *__restrict__ slots, const int2 *__restrict__ chunks,
                                                          const float *__restrict__ G, float *__restrict__ Mo, float *__restrict__ Vo,
                                                          const StepScalars *__restrict__ sc, float b1, float b2, float eps, float decay) {
    if (sc->bad) return;
    const int2 ch = chunks[blockIdx.x];
    const SlotDev sl = slots[ch.x];
    const float step_size = sc->step_size, gs = sc->gscale;
    const unsigned long long i0 = (unsigned long long)ch.y * ADAMW_CHUNK, i1 = min(sl.n, i0 + ADAMW_CHUNK);
    const float *g = G + sl.off;
    float *m = Mo + sl.off, *v = Vo + sl.off, *p = sl.p;
    for (unsigned long long i = i0 + threadIdx.x; i < i1; i += 256) {
        const float gi = g[i] * gs;
        const float mi = m[i] * b1 + gi * (1.0f - b1);
        const float vi = v[i] * b2 + gi * gi * (1.0f - b2);
        m[i] = mi;
        v[i] = vi;
        float pi = p[i] - step_size * (mi / (sqrtf(vi) + eps));
        if (decay > 0.f) pi -= decay * pi;
        p[i] = pi;
    }
}

// ---------------------------------------------------------------------------------------------- LayerNorm backward
// dx = add + rstd (g - mean(g) - xhat mean(g xhat)), g = dy w; one wavefront per row, d = 64 * NPL
// (stats != nullptr: the row's (mean, rstd) are kept for ln_param_grad_kernel)
template <int NPL>
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                        const float *__restrict__ dy, const float *add, float *dx,
                                                        int rows, float eps, float2 *stats = nullptr) {
    constexpr int d = 64 * NPL;
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + (size_t)row * d, *dyr = dy + (size_t)row * d;
    float xv[NPL], gv[NPL];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NPL; ++k) { xv[k] = xr[lane + 64 * k]; s += xv[k]; }
    const float mu = wave_sum(s) / d;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NPL; ++k) { xv[k] -= mu; q += xv[k] * xv[k]; }
    const float rstd = rsqrtf(wave_sum(q) / d + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        xv[k] *= rstd;                                       // xhat
        gv[k] = dyr[lane + 64 * k] * w[lane + 64 * k];
        sg += gv[k];
        sgx += gv[k] * xv[k];
    }
    const float mg = wave_sum(sg) / d, mgx = wave_sum(sgx) / d;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const float r = rstd * (gv[k] - mg - xv[k] * mgx);
        dx[(size_t)row * d + lane + 64 * k] = add ? add[(size_t)row * d + lane + 64 * k] + r : r;
    }
    if (stats && lane == 0) stats[row] = make_float2(mu, rstd);
}
// the LayerNorm's own gradients: gw[c] += sum_r dy[r, c] (x[r, c] - mean_r) rstd_r, gb[c] += sum_r dy[r, c].  Block (x, y):
// 64 columns x rows [128 y, 128 y + 128), four row lanes per column reduced through LDS, one atomic add per column
__global__ __launch_bounds__(256) void ln_param_grad_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                            const float2 *__restrict__ stats, int rows, int d,
                                                            float *__restrict__ gw, float *__restrict__ gb) {
    __shared__ float sw[4][64], sb[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * 128, r1 = min(rows, r0 + 128);
    float a = 0.f, b = 0.f;
    if (c < d)
        for (int r = r0 + rl; r < r1; r += 4) {
            const float2 st = stats[r];
            const float g = dy[(size_t)r * d + c];
            a += g * (x[(size_t)r * d + c] - st.x) * st.y;
            b += g;
        }
    sw[rl][cl] = a;
    sb[rl][cl] = b;
    __syncthreads();
    if (rl == 0 && c < d) {
        atomicAdd(gw + c, (sw[0][cl] + sw[1][cl]) + (sw[2][cl] + sw[3][cl]));
        atomicAdd(gb + c, (sb[0][cl] + sb[1][cl]) + (sb[2][cl] + sb[3][cl]));
    }
}

// ---------------------------------------------------------------------------------------------- attention backward
// Softmax attention, rows = (sample, position) with S positions per sample, qkv rows [q | k | v] of 3 d floats, head h at
// columns h HD; CAUSAL (GPT-2, HD = 64, scale 1/8) or over all S keys (TransformerMapper, HD = 96, scale HD^-0.5).
// One wavefront per (sample, head, query i); a lane owns the head dimensions lane and lane + 64 (< HD).
//   s_j = q_i . k_j scale, p = softmax_j(s), dP_j = dO_i . v_j, D = sum_j p_j dP_j, dS_j = p_j (dP_j - D)
//   dq_i = sum_j dS_j k_j scale;  lse_i and D_i are kept for the key-side kernel
// mask != nullptr (GPT-2's attn_dropout, [B, H, S, S] keep bytes): the weights that multiplied V were p_j m_ij / keep,
// so dP_j = (dO_i . v_j) m_ij / keep -- everything else as above
template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const float *__restrict__ qkv, const float *__restrict__ dout,
                                                         float *__restrict__ dqkv, float *__restrict__ lse_out,
                                                         float *__restrict__ dsum_out, int total, int S, int heads,
                                                         float scale, const uint8_t *__restrict__ mask = nullptr,
                                                         float inv_keep = 1.f) {
    constexpr int NE = (HD + 63) / 64;
    extern __shared__ float sh[];                     // [4 waves][2][S]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave;
    if (gw >= total) return;
    const int i = gw % S, bh = gw / S, h = bh % heads, b = bh / heads;
    const int d = heads * HD;
    float *sc = sh + (size_t)wave * 2 * S, *dp = sc + S;
    const size_t row = (size_t)b * S + i;
    const int nk = CAUSAL ? i + 1 : S;
    float q[NE], go[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const bool ok = lane + 64 * e < HD;
        q[e] = ok ? qkv[row * 3 * d + h * HD + lane + 64 * e] : 0.f;
        go[e] = ok ? dout[row * d + h * HD + lane + 64 * e] : 0.f;
    }
    for (int j = 0; j < nk; ++j) {
        const float *kr = qkv + ((size_t)b * S + j) * 3 * d + d + h * HD;
        float a = 0.f, t = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e)
            if (lane + 64 * e < HD) { a += q[e] * kr[lane + 64 * e]; t += go[e] * kr[d + lane + 64 * e]; }
        a = wave_sum(a) * scale;
        t = wave_sum(t);
        if (mask) t *= mask[((size_t)bh * S + i) * S + j] ? inv_keep : 0.f;
        if (lane == 0) { sc[j] = a; dp[j] = t; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float mx = -INFINITY;
    for (int j = lane; j < nk; j += 64) mx = fmaxf(mx, sc[j]);
    mx = wave_max(mx);
    float l = 0.f;
    for (int j = lane; j < nk; j += 64) l += expf(sc[j] - mx);
    l = wave_sum(l);
    const float lse = mx + logf(l);
    float D = 0.f;
    for (int j = lane; j < nk; j += 64) D += expf(sc[j] - lse) * dp[j];
    D = wave_sum(D);
    for (int j = lane; j < nk; j += 64) sc[j] = expf(sc[j] - lse) * (dp[j] - D);       // dS_j
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float dq[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) dq[e] = 0.f;
    for (int j = 0; j < nk; ++j) {
        const float *kr = qkv + ((size_t)b * S + j) * 3 * d + d + h * HD;
#pragma unroll
        for (int e = 0; e < NE; ++e)
            if (lane + 64 * e < HD) dq[e] += sc[j] * kr[lane + 64 * e];
    }
#pragma unroll
    for (int e = 0; e < NE; ++e)
        if (lane + 64 * e < HD) dqkv[row * 3 * d + h * HD + lane + 64 * e] = dq[e] * scale;
    if (lane == 0) { lse_out[gw] = lse; dsum_out[gw] = D; }
}
// one wavefront per (sample, head, key j): dk_j = sum_i dS_ij q_i scale, dv_j = sum_i p_ij dO_i  (i >= j when CAUSAL)
template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(const float *__restrict__ qkv, const float *__restrict__ dout,
                                                          float *__restrict__ dqkv, const float *__restrict__ lse_in,
                                                          const float *__restrict__ dsum_in, int total, int S, int heads,
                                                          float scale, const uint8_t *__restrict__ mask = nullptr,
                                                          float inv_keep = 1.f) {
    constexpr int NE = (HD + 63) / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave;
    if (gw >= total) return;
    const int j = gw % S, bh = gw / S, h = bh % heads, b = bh / heads;
    const int d = heads * HD;
    const size_t rowj = (size_t)b * S + j;
    float k[NE], v[NE], dk[NE], dv[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const bool ok = lane + 64 * e < HD;
        k[e] = ok ? qkv[rowj * 3 * d + d + h * HD + lane + 64 * e] : 0.f;
        v[e] = ok ? qkv[rowj * 3 * d + 2 * d + h * HD + lane + 64 * e] : 0.f;
        dk[e] = dv[e] = 0.f;
    }
    for (int i = CAUSAL ? j : 0; i < S; ++i) {
        const size_t rowi = (size_t)b * S + i;
        float q[NE], go[NE];
        float a = 0.f, t = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const bool ok = lane + 64 * e < HD;
            q[e] = ok ? qkv[rowi * 3 * d + h * HD + lane + 64 * e] : 0.f;
            go[e] = ok ? dout[rowi * d + h * HD + lane + 64 * e] : 0.f;
            a += q[e] * k[e];
            t += go[e] * v[e];
        }
        const int gi = bh * S + i;
        const float mk = mask ? (mask[(size_t)gi * S + j] ? inv_keep : 0.f) : 1.f;
        const float p = expf(wave_sum(a) * scale - lse_in[gi]);
        const float ds = p * (wave_sum(t) * mk - dsum_in[gi]);
#pragma unroll
        for (int e = 0; e < NE; ++e) { dk[e] += ds * q[e]; dv[e] += p * mk * go[e]; }
    }
#pragma unroll
    for (int e = 0; e < NE; ++e)
        if (lane + 64 * e < HD) {
            dqkv[rowj * 3 * d + d + h * HD + lane + 64 * e] = dk[e] * scale;
            dqkv[rowj * 3 * d + 2 * d + h * HD + lane + 64 * e] = dv[e];
        }
}

// ---------------------------------------------------------------------------------------------- attention, block form
// The train step's sequences are short (S = prefix_length + caption <= ~100; the TransformerMapper's 80): ONE BLOCK per
// (sample, head) stages the head's K and V (later Q and dO) in LDS once and its eight wavefronts walk the queries (keys) --
// where the per-(sample, head, query) wavefronts above re-read every key from L2 and spend two 64-lane reductions per
// (query, key) pair (TransformerMapper, S = 80, 8 x 96: 319 + 173 us per layer backward, 310 us forward).  Lane-per-key
// for the scores (a key's row of the LDS tile per lane, row stride HD + 1: conflict-free; the query is broadcast from a
// per-wavefront buffer), lane-per-dimension for the products with V / K / Q / dO (the weight is the broadcast).
//   forward:  P = softmax(q K^T scale) [mask / keep], out = P V
//   backward: phase 1 per query i -- p, dP = (dO_i . v_j) [mask / keep], D = sum p dP, dS = p (dP - D), dq_i = scale dS K;
//             P[i][j] (as it multiplied V) and dS[i][j] stay in LDS;  phase 2 per key j -- Q / dO take K / V's place:
//             dk_j = scale sum_i dS[i][j] q_i,  dv_j = sum_i P[i][j] dO_i
// mask: GPT-2's attn_dropout keep bytes [B, H, S, S] (nullptr: none).  Layout of qkv / dqkv as above.
constexpr int ATTN_BLK_NW = 16;                                     // wavefronts per block (1024 threads: the loops are LDS-latency-bound)
template <int HD> struct AttnBlk {
    static constexpr int LD = HD + 1;
    static size_t fwd_bytes(int S) { return ((size_t)2 * S * LD + ATTN_BLK_NW * (HD + S)) * sizeof(float); }
    static size_t bwd_bytes(int S) {
        return ((size_t)2 * S * LD + (size_t)2 * S * (S + 1) + ATTN_BLK_NW * (2 * HD + S)) * sizeof(float);
    }
};
template <int HD>
__device__ __forceinline__ void attn_blk_stage(float *__restrict__ dst, const float *__restrict__ src, int ld, int S) {
    // dst[j][e] = src[j * ld + e], j < S, e < HD   (consecutive threads read consecutive e: coalesced rows)
    for (int i = threadIdx.x; i < S * HD; i += 64 * ATTN_BLK_NW) {
        const int j = i / HD, e = i - j * HD;
        dst[j * (HD + 1) + e] = src[(size_t)j * ld + e];
    }
}
// sum_e x[e] y[e] over HD (a multiple of 16): the loop is bound by LDS latency, not by its FMAs -- left alone the compiler
// reuses one register pair per step and waits for every read (load, s_waitcnt lgkmcnt(0), fma, ...), so 16 elements of each
// vector are read into registers FIRST (the sched_barrier keeps the reads ahead of the arithmetic), then four independent
// chains consume them
template <int HD>
__device__ __forceinline__ float dot_lds(const float *__restrict__ x, const float *__restrict__ y) {
    static_assert(HD % 16 == 0, "head dimension");
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 2
    for (int e0 = 0; e0 < HD; e0 += 16) {
        float xv[16], yv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { xv[u] = x[e0 + u]; yv[u] = y[e0 + u]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 16; u += 4) {
            a0 += xv[u] * yv[u];
            a1 += xv[u + 1] * yv[u + 1];
            a2 += xv[u + 2] * yv[u + 2];
            a3 += xv[u + 3] * yv[u + 3];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return (a0 + a1) + (a2 + a3);
}
// acc[e] += sum_{j in [j0, j1)} w[j ws] M[j LD + lane + 64 e]: rows of an LDS tile weighted by a broadcast column, eight
// rows read before they are used
template <int HD>
__device__ __forceinline__ void wsum_rows(const float *__restrict__ w, int ws, const float *__restrict__ M, int j0, int j1, int lane,
                                          float (&acc)[(HD + 63) / 64]) {
    constexpr int LD = HD + 1, NE = (HD + 63) / 64, U = 8;
    float p[2][NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) p[0][e] = p[1][e] = 0.f;
    int j = j0;
    for (; j + U - 1 < j1; j += U) {
        float wv[U], mv[U][NE];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            wv[u] = w[(j + u) * ws];
#pragma unroll
            for (int e = 0; e < NE; ++e) mv[u][e] = lane + 64 * e < HD ? M[(j + u) * LD + lane + 64 * e] : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < NE; ++e) p[u & 1][e] += wv[u] * mv[u][e];
        __builtin_amdgcn_sched_barrier(0);
    }
    for (; j < j1; ++j) {
        const float wj = w[j * ws];
#pragma unroll
        for (int e = 0; e < NE; ++e)
            if (lane + 64 * e < HD) p[0][e] += wj * M[j * LD + lane + 64 * e];
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) acc[e] += p[0][e] + p[1][e];
}
#define ATTN_WAVE_SYNC()                                        \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      \
    __builtin_amdgcn_wave_barrier();                            \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
template <int HD, bool CAUSAL>
__global__ __launch_bounds__(64 * ATTN_BLK_NW) void attn_blk_fwd_kernel(const float *__restrict__ qkv, float *__restrict__ out, int S,
                                                                       int heads, float scale,
                                                                       const uint8_t *__restrict__ mask, float inv_keep) {
    constexpr int LD = HD + 1, NE = (HD + 63) / 64, NW = ATTN_BLK_NW;
    static_assert(HD % 4 == 0, "head dimension");
    extern __shared__ float sh[];
    float *Ks = sh, *Vs = Ks + S * LD, *qb = Vs + S * LD, *pb = qb + NW * HD;     // [S][LD] x 2, [NW][HD], [NW][S]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.x, h = bh % heads, b = bh / heads, d = heads * HD;
    const float *base = qkv + (size_t)b * S * 3 * d + h * HD;
    attn_blk_stage<HD>(Ks, base + d, 3 * d, S);
    attn_blk_stage<HD>(Vs, base + 2 * d, 3 * d, S);
    __syncthreads();
    float *q = qb + wave * HD, *p = pb + wave * S;
    for (int i = wave; i < S; i += NW) {
        const int nk = CAUSAL ? i + 1 : S;
#pragma unroll
        for (int e = 0; e < NE; ++e)
            if (lane + 64 * e < HD) q[lane + 64 * e] = base[(size_t)i * 3 * d + lane + 64 * e] * scale;
        ATTN_WAVE_SYNC()
        float sc[2], mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = lane + 64 * r;
            sc[r] = j < nk ? dot_lds<HD>(q, Ks + j * LD) : -INFINITY;
            mx = fmaxf(mx, sc[r]);
        }
        mx = wave_max(mx);
        float l = 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r) { sc[r] = lane + 64 * r < nk ? expf(sc[r] - mx) : 0.f; l += sc[r]; }
        const float inv = 1.0f / wave_sum(l);
        const uint8_t *mr = mask ? mask + ((size_t)bh * S + i) * S : nullptr;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = lane + 64 * r;
            if (j < nk) p[j] = sc[r] * inv * (mr ? (mr[j] ? inv_keep : 0.f) : 1.f);
        }
        ATTN_WAVE_SYNC()
        float o[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) o[e] = 0.f;
        wsum_rows<HD>(p, 1, Vs, 0, nk, lane, o);
#pragma unroll
        for (int e = 0; e < NE; ++e)
            if (lane + 64 * e < HD) out[((size_t)b * S + i) * d + h * HD + lane + 64 * e] = o[e];
    }
}
template <int HD, bool CAUSAL>
__global__ __launch_bounds__(64 * ATTN_BLK_NW) void attn_blk_bwd_kernel(const float *__restrict__ qkv, const float *__restrict__ dout,
                                                                       float *__restrict__ dqkv, int S, int heads, float scale,
                                                                       const uint8_t *__restrict__ mask, float inv_keep) {
    constexpr int LD = HD + 1, NE = (HD + 63) / 64, NW = ATTN_BLK_NW;
    extern __shared__ float sh[];
    const int SP = S + 1;
    float *A = sh, *Bm = A + S * LD, *P = Bm + S * LD, *dS = P + S * SP, *qb = dS + S * SP, *gb = qb + NW * HD,
          *sb = gb + NW * HD;                                       // [S][LD] x 2, [S][S + 1] x 2, [NW][HD] x 2, [NW][S]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.x, h = bh % heads, b = bh / heads, d = heads * HD;
    const float *base = qkv + (size_t)b * S * 3 * d + h * HD;
    const float *gbase = dout + (size_t)b * S * d + h * HD;
    float *obase = dqkv + (size_t)b * S * 3 * d + h * HD;
    attn_blk_stage<HD>(A, base + d, 3 * d, S);                      // K
    attn_blk_stage<HD>(Bm, base + 2 * d, 3 * d, S);                 // V
    __syncthreads();
    float *q = qb + wave * HD, *g = gb + wave * HD, *ds = sb + wave * S;
    // ---- phase 1: one query per wavefront pass
    for (int i = wave; i < S; i += NW) {
        const int nk = CAUSAL ? i + 1 : S;
#pragma unroll
        for (int e = 0; e < NE; ++e)
            if (lane + 64 * e < HD) {
                q[lane + 64 * e] = base[(size_t)i * 3 * d + lane + 64 * e] * scale;
                g[lane + 64 * e] = gbase[(size_t)i * d + lane + 64 * e];
            }
        ATTN_WAVE_SYNC()
        const uint8_t *mr = mask ? mask + ((size_t)bh * S + i) * S : nullptr;
        float sc[2], dp[2], mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = lane + 64 * r;
            float a = -INFINITY, t = 0.f;
            if (j < nk) {
                a = dot_lds<HD>(q, A + j * LD);
                t = dot_lds<HD>(g, Bm + j * LD);
                if (mr) t *= mr[j] ? inv_keep : 0.f;
            }
            sc[r] = a;
            dp[r] = t;
            mx = fmaxf(mx, sc[r]);
        }
        mx = wave_max(mx);
        float l = 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r) { sc[r] = lane + 64 * r < nk ? expf(sc[r] - mx) : 0.f; l += sc[r]; }
        const float inv = 1.0f / wave_sum(l);
        float D = 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r) { sc[r] *= inv; D += sc[r] * dp[r]; }
        D = wave_sum(D);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = lane + 64 * r;
            if (j < S) {            // (columns past the causal limit hold zeros)
                const float w = j < nk ? sc[r] * (dp[r] - D) : 0.f;
                ds[j] = w;
                dS[i * SP + j] = w;
                P[i * SP + j] = j < nk ? sc[r] * (mr ? (mr[j] ? inv_keep : 0.f) : 1.f) : 0.f;
            }
        }
        ATTN_WAVE_SYNC()
        float dq[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) dq[e] = 0.f;
        wsum_rows<HD>(ds, 1, A, 0, nk, lane, dq);
#pragma unroll
        for (int e = 0; e < NE; ++e)
            if (lane + 64 * e < HD) obase[(size_t)i * 3 * d + lane + 64 * e] = dq[e] * scale;
    }
    __syncthreads();
    // ---- phase 2: Q and dO take the place of K and V; one key per wavefront pass
    attn_blk_stage<HD>(A, base, 3 * d, S);                          // Q
    attn_blk_stage<HD>(Bm, gbase, d, S);                            // dO
    __syncthreads();
    for (int j = wave; j < S; j += NW) {
        float dk[NE], dv[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) dk[e] = dv[e] = 0.f;
        const int i0 = CAUSAL ? j : 0;
        wsum_rows<HD>(dS + j, SP, A, i0, S, lane, dk);              // column j of dS against the rows of Q
        wsum_rows<HD>(P + j, SP, Bm, i0, S, lane, dv);              // column j of P against the rows of dO
#pragma unroll
        for (int e = 0; e < NE; ++e)
            if (lane + 64 * e < HD) {
                obase[(size_t)j * 3 * d + d + lane + 64 * e] = dk[e] * scale;
                obase[(size_t)j * 3 * d + 2 * d + lane + 64 * e] = dv[e];
            }
    }
}
#undef ATTN_WAVE_SYNC
// the block kernels when a head's tiles fit the CU's LDS (S <= 128), the per-query wavefront kernels otherwise
constexpr size_t ATTN_BLK_LDS_MAX = 150 * 1024;
template <int HD, bool CAUSAL>
static int attn_blk_fwd(hipStream_t st, const float *qkv, float *out, int B, int S, int heads, float scale, const uint8_t *mask,
                        float inv_keep) {
    const size_t lds = AttnBlk<HD>::fwd_bytes(S);
    static bool attr = false;
    if (!attr) {
        CAPDEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_blk_fwd_kernel<HD, CAUSAL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATTN_BLK_LDS_MAX));
        attr = true;
    }
    hipLaunchKernelGGL((attn_blk_fwd_kernel<HD, CAUSAL>), dim3(B * heads), dim3(64 * ATTN_BLK_NW), lds, st, qkv, out, S, heads, scale, mask, inv_keep);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}
template <int HD, bool CAUSAL>
static int attn_blk_bwd(hipStream_t st, const float *qkv, const float *dout, float *dqkv, int B, int S, int heads, float scale,
                        const uint8_t *mask, float inv_keep) {
    const size_t lds = AttnBlk<HD>::bwd_bytes(S);
    static bool attr = false;
    if (!attr) {
        CAPDEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_blk_bwd_kernel<HD, CAUSAL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATTN_BLK_LDS_MAX));
        attr = true;
    }
    hipLaunchKernelGGL((attn_blk_bwd_kernel<HD, CAUSAL>), dim3(B * heads), dim3(64 * ATTN_BLK_NW), lds, st, qkv, dout, dqkv, S, heads, scale, mask,
                       inv_keep);
    CAPDEC_HIP(hipGetLastError());
    return 0;
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

// ---------------------------------------------------------------------------------------------- workspace
// One trainable tensor: where it lives and where its gradient / AdamW moments sit in the three arenas (same offset in each)
struct Slot {
    float *p;
    size_t n, off;
    int rows = 0, cols = 0;      // rows > 0: a GPT-2 Conv1D weight, kept [out = rows, in = cols] on the device, [in, out] in checkpoints
};
struct TrainState {
    // transposed copies of the frozen GPT-2 weights: the "[N, K]" operand of dX = dY W^T (= the checkpoint's own Conv1D
    // layout [in, out]); wte_t [d][Vp] zero-padded to a multiple of 64 columns
    struct LayerT { float *wqkv_t, *wproj_t, *wfc_t, *wproj2_t; };
    std::vector<LayerT> lt;
    float *wte_t = nullptr;
    int Vp = 0;
    std::vector<void *> owned;
    bool weights_ready = false;
    // the mapper's trainable tensors (build_slots) + gradient and moment arenas
    std::vector<Slot> slots;
    size_t n_params = 0;
    bool train_gpt = false;                  // scope (capdec_ctx::train_scope at creation): 0 the mapper (GPT-2 frozen), 1 GPT-2 as well
    int gpt_slot0 = -1;                      // first GPT-2 slot: wte, wpe, 12 per layer, ln_f weight / bias
    DBuf G, Mo, Vo;
    DBuf slotdev, chunkdev;                  // adamw_multi_kernel's tables (built with the slots)
    int n_chunks = 0;
    // saved activations + gradient scratch (grow-only)
    DBuf pe, emb, hs, a, qkv, att, hmid, fc, gl, hf, hfl, logits, rloss, cnt;
    DBuf dh, dh2, da, dqkv, datt, dfc, dhfl, lse, dsum, dy, tA, tB, wT, lnstat;
    DBuf dmask, dinj, ytmp, dtmp;            // scope 1 with dropout: this step's keep-masks, masks injected for the next
                                             // step, a Conv1D output before its dropout, a gradient after one
    size_t dinj_n = 0;                       // bytes waiting in dinj (0: the next step draws its masks from Philox)
    size_t dmask_n = 0;                      // bytes of the last step's mask stream (capdec_train_get_dropout_masks)
    unsigned long long draws = 0;            // mask streams drawn from Philox so far (the counter's high half)
    DBuf hid, dhid;                          // MLP mapper: tanh output, its gradient
    DBuf t_lin, t_seq, t_a1, t_qkv, t_att, t_mid, t_a2, t_r;      // TransformerMapper: per-layer saved activations
    DBuf t_ds, t_ds2, t_da, t_dr, t_dqkv, t_datt, t_dlin;         // ... gradient scratch
    long long step = 0;                      // train steps run with apply_update (the mask stream's counter; the AdamW
                                             // update counter lives on the device: StepScalars::updates)
    bool have_grads = false;
    bool scalars_ready = false;
    void release() {
        for (void *p : owned) (void)hipFree(p);
        owned.clear();
        lt.clear();
        slots.clear();
        wte_t = nullptr;
        weights_ready = false;
        DBuf *bufs[] = {&G, &Mo, &Vo, &slotdev, &chunkdev, &pe, &emb, &hs, &a, &qkv, &att, &hmid, &fc, &gl, &hf, &hfl, &logits, &rloss, &cnt,
                        &dh, &dh2, &da, &dqkv, &datt, &dfc, &dhfl, &lse, &dsum, &dy, &tA, &tB, &wT, &lnstat, &hid, &dhid,
                        &dmask, &dinj, &ytmp, &dtmp,
                        &t_lin, &t_seq, &t_a1, &t_qkv, &t_att, &t_mid, &t_a2, &t_r, &t_ds, &t_ds2, &t_da, &t_dr, &t_dqkv,
                        &t_datt, &t_dlin};
        for (DBuf *b : bufs) b->release();
        step = 0;
        have_grads = false;
        scalars_ready = false;
        dinj_n = dmask_n = 0;
        draws = 0;
    }
    float *grad(int slot) { return G.as<float>() + slots[slot].off; }
};

void train_release(capdec_ctx *c) {
    if (!c->train) return;
    for (void *p : c->train->owned) drop_planes_of(c, p);      // planes packed from the transposed copies die with them
    c->train->release();
    delete c->train;
    c->train = nullptr;
}

// Slot order (capdec.h: capdec_train_get).  MLP: model.0.weight, model.0.bias, model.2.weight, model.2.bias.
// TransformerMapper: linear.weight, linear.bias, prefix_const, then per layer norm1.weight, norm1.bias,
// attn.to_queries.weight, attn.to_keys_values.weight (adjacent halves of the fused [3d, d] projection on the device),
// attn.project.weight, attn.project.bias, norm2.weight, norm2.bias, mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias
static int build_slots(capdec_ctx *c, TrainState &t) {
    if (!t.slots.empty()) return 0;
    Mapper &m = c->map;
    const size_t d = m.d;
    auto add = [&](float *p, size_t n) { t.slots.push_back(Slot{p, n, t.n_params}); t.n_params += n; };
    auto add_t = [&](float *p, int rows, int cols) {
        Slot sl{p, (size_t)rows * cols, t.n_params};
        sl.rows = rows;
        sl.cols = cols;
        t.slots.push_back(sl);
        t.n_params += sl.n;
    };
    t.n_params = 0;
    if (m.kind == 1) {
        const size_t O = (size_t)m.P * d;
        add(m.w1, (size_t)m.hidden * m.D); add(m.b1, m.hidden); add(m.w2, O * m.hidden); add(m.b2, O);
    } else {
        add(m.lin_w, (size_t)m.clip_len * d * m.D); add(m.lin_b, (size_t)m.clip_len * d); add(m.prefix_const, (size_t)m.P * d);
        for (TMapLayer &l : m.layers) {
            add(l.n1w, d); add(l.n1b, d);
            add(l.wqkv, d * d); add(l.wqkv + d * d, 2 * d * d);
            add(l.wproj, d * d); add(l.bproj, d);
            add(l.n2w, d); add(l.n2b, d);
            add(l.wfc1, (size_t)m.mlp_hidden * d); add(l.bfc1, m.mlp_hidden);
            add(l.wfc2, d * m.mlp_hidden); add(l.bfc2, d);
        }
    }
    if (t.train_gpt) {
        // the reference's default run: AdamW(model.parameters()) (train.py:326) -- every GPT-2 tensor (the tied lm_head is wte)
        Gpt2 &g = c->gpt;
        const int gd = g.d;
        t.gpt_slot0 = (int)t.slots.size();
        add(g.wte, (size_t)g.vocab * gd); add(g.wpe, (size_t)g.n_pos * gd);
        for (Gpt2Layer &l : g.layers) {
            add(l.ln1w, gd); add(l.ln1b, gd);
            add_t(l.wqkv, 3 * gd, gd); add(l.bqkv, 3 * gd);
            add_t(l.wproj, gd, gd); add(l.bproj, gd);
            add(l.ln2w, gd); add(l.ln2b, gd);
            add_t(l.wfc, 4 * gd, gd); add(l.bfc, 4 * gd);
            add_t(l.wproj2, gd, 4 * gd); add(l.bproj2, gd);
        }
        add(g.lnfw, gd); add(g.lnfb, gd);
    }
    CAPDEC_TRY(t.G.ensure(t.n_params * 4));
    CAPDEC_TRY(t.Mo.ensure(t.n_params * 4));
    CAPDEC_TRY(t.Vo.ensure(t.n_params * 4));
    CAPDEC_HIP(hipMemsetAsync(t.Mo.p, 0, t.n_params * 4, c->stream));
    CAPDEC_HIP(hipMemsetAsync(t.Vo.p, 0, t.n_params * 4, c->stream));
    {   // the update's tables: one entry per slot, one per 16 384-element chunk
        std::vector<SlotDev> sd;
        std::vector<int2> ch;
        for (size_t i = 0; i < t.slots.size(); ++i) {
            sd.push_back(SlotDev{t.slots[i].p, (unsigned long long)t.slots[i].off, (unsigned long long)t.slots[i].n});
            for (size_t k = 0; k * ADAMW_CHUNK < t.slots[i].n; ++k) ch.push_back(make_int2((int)i, (int)k));
        }
        t.n_chunks = (int)ch.size();
        CAPDEC_TRY(t.slotdev.ensure(sd.size() * sizeof(SlotDev)));
        CAPDEC_TRY(t.chunkdev.ensure(ch.size() * sizeof(int2)));
        CAPDEC_HIP(hipMemcpyAsync(t.slotdev.p, sd.data(), sd.size() * sizeof(SlotDev), hipMemcpyHostToDevice, c->stream));
        CAPDEC_HIP(hipMemcpyAsync(t.chunkdev.p, ch.data(), ch.size() * sizeof(int2), hipMemcpyHostToDevice, c->stream));
        CAPDEC_HIP(hipStreamSynchronize(c->stream));              // (the host vectors die here; once per optimizer)
    }
    return 0;
}

static int dev_alloc(std::vector<void *> &owned, size_t bytes, float **out) {
    void *p = nullptr;
    CAPDEC_HIP(hipMalloc(&p, bytes));
    owned.push_back(p);
    *out = reinterpret_cast<float *>(p);
    return 0;
}
static int transpose_pad(capdec_ctx *c, const float *src, int rows, int cols, float *dst, int ld) {
    hipLaunchKernelGGL(transpose_pad_kernel, dim3((cols + 31) / 32, (ld + 31) / 32), dim3(256), 0, c->stream, src, rows, cols,
                       dst, ld);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}
// the forward weights are [out, in] on the device (weights.hip: upload_transposed); dX needs [in, out]
static int prepare_backward_weights(capdec_ctx *c, TrainState &t) {
    if (t.weights_ready) return 0;
    const Gpt2 &g = c->gpt;
    const int d = g.d;
    t.lt.resize(g.n_layer);
    for (int i = 0; i < g.n_layer; ++i) {
        const Gpt2Layer &w = g.layers[i];
        TrainState::LayerT &lt = t.lt[i];
        CAPDEC_TRY(dev_alloc(t.owned, (size_t)d * 3 * d * 4, &lt.wqkv_t));
        CAPDEC_TRY(transpose_pad(c, w.wqkv, 3 * d, d, lt.wqkv_t, 3 * d));          // [3d, d] -> [d, 3d]
        CAPDEC_TRY(dev_alloc(t.owned, (size_t)d * d * 4, &lt.wproj_t));
        CAPDEC_TRY(transpose_pad(c, w.wproj, d, d, lt.wproj_t, d));
        CAPDEC_TRY(dev_alloc(t.owned, (size_t)d * 4 * d * 4, &lt.wfc_t));
        CAPDEC_TRY(transpose_pad(c, w.wfc, 4 * d, d, lt.wfc_t, 4 * d));            // [4d, d] -> [d, 4d]
        CAPDEC_TRY(dev_alloc(t.owned, (size_t)4 * d * d * 4, &lt.wproj2_t));
        CAPDEC_TRY(transpose_pad(c, w.wproj2, d, 4 * d, lt.wproj2_t, d));          // [d, 4d] -> [4d, d]
    }
    t.Vp = (g.vocab + 63) / 64 * 64;
    CAPDEC_TRY(dev_alloc(t.owned, (size_t)d * t.Vp * 4, &t.wte_t));
    CAPDEC_TRY(transpose_pad(c, g.wte, g.vocab, d, t.wte_t, t.Vp));               // [V, d] -> [d, Vp], zero padding
    t.weights_ready = true;
    return 0;
}
// full-model scope: the GPT-2 weights moved -- the transposed copies follow (same buffers)
static int refresh_backward_weights(capdec_ctx *c, TrainState &t) {
    const Gpt2 &g = c->gpt;
    const int d = g.d;
    for (int i = 0; i < g.n_layer; ++i) {
        const Gpt2Layer &w = g.layers[i];
        TrainState::LayerT &lt = t.lt[i];
        CAPDEC_TRY(transpose_pad(c, w.wqkv, 3 * d, d, lt.wqkv_t, 3 * d));
        CAPDEC_TRY(transpose_pad(c, w.wproj, d, d, lt.wproj_t, d));
        CAPDEC_TRY(transpose_pad(c, w.wfc, 4 * d, d, lt.wfc_t, 4 * d));
        CAPDEC_TRY(transpose_pad(c, w.wproj2, d, 4 * d, lt.wproj2_t, d));
    }
    return transpose_pad(c, g.wte, g.vocab, d, t.wte_t, t.Vp);
}

// C[M, N] = A[M, K] . Bt[N, K]^T on the fp32-accurate two-fp16-plane kernels of the inference path (the backward pass runs
// on un-normalised, loss-scaled gradients for exactly that: see the header); CAPDEC_TRAIN_F16X2=0: the native fp32 MFMA
// GEMM instead (39.4 vs 23.6 ms per step at the reference's default geometry, profiles/r5_first_call.txt).
// static_weight: Bt never changes, its packed planes may be cached
static int gemm_fp32(capdec_ctx *c, const float *A, int lda, const float *Bt, int ldb, float *C, int ldc, int M, int N, int K,
                     bool static_weight = false) {
    if (c->tune.train_f16x2 && K % 64 == 0 && ldb == K && lda % 4 == 0)
        return gemm(c, A, lda, Bt, ldb, C, ldc, M, N, K, nullptr, CAPDEC_ACT_NONE, nullptr, 0, static_weight);
    GemmEpilogue e;
    e.tune = &c->tune;
    ProfScope ps(c, F_GEMM, 2.0 * M * (double)N * K);
    return launch_gemm_f32(c->stream, A, lda, Bt, ldb, C, ldc, M, N, K, e);
}
static inline dim3 grid1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }
static inline int pad32(int n) { return (n + 31) / 32 * 32; }
// K of a weight-gradient product = its row count, zero-padded to what the GEMM in use wants (32; 64 for the f16x2 kernels)
static inline int pad_rows(const capdec_ctx *c, int n) { return c->tune.train_f16x2 ? (n + 63) / 64 * 64 : pad32(n); }

static int ln_bwd(capdec_ctx *c, const float *x, const float *w, const float *dy, const float *add, float *dx, int rows,
                  int d, float eps, float *gw = nullptr, float *gb = nullptr) {
    CAPDEC_CHECK(d == 768, "train: LayerNorm backward is instantiated for d = 768");
    float2 *stats = nullptr;
    if (gw) {
        CAPDEC_TRY(c->train->lnstat.ensure((size_t)rows * sizeof(float2)));
        stats = c->train->lnstat.as<float2>();
    }
    hipLaunchKernelGGL(ln_bwd_dx_kernel<12>, dim3((rows + 3) / 4), dim3(256), 0, c->stream, x, w, dy, add, dx, rows, eps, stats);
    if (gw)
        hipLaunchKernelGGL(ln_param_grad_kernel, dim3((d + 63) / 64, (rows + 127) / 128), dim3(256), 0, c->stream, x, dy, stats, rows,
                           d, gw, gb);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}
// dX = dY W for an nn.Linear weight W [out, in] that CHANGES every step: transposed into the scratch `wT` first
static int linear_dx(capdec_ctx *c, TrainState &t, const float *dy, const float *W, float *dx, int M, int out, int in) {
    CAPDEC_TRY(t.wT.ensure((size_t)out * in * 4));
    CAPDEC_TRY(transpose_pad(c, W, out, in, t.wT.as<float>(), out));               // [in][out]
    return gemm_fp32(c, dy, out, t.wT.as<float>(), out, dx, in, M, in, out);
}
// dW = dY^T X ([out, in]; dY [rows, out], X [rows, in]; the GEMM's K = rows, zero-padded to a multiple of 32), db = colsum(dY)
static int linear_dw(capdec_ctx *c, TrainState &t, const float *dy, const float *x, int rows, int out, int in, float *gW,
                     float *gb) {
    if (c->tune.train_f16x2 && in % 4 == 0) {
        // both operands packed transposed from their row-major form: no fp32 transposed copies (a "TN" operand loader)
        CAPDEC_TRY(gemm_tn(c, dy, out, x, in, rows, out, in, gW, in));
    } else {
        const int Kp = pad_rows(c, rows);
        CAPDEC_TRY(t.tA.ensure((size_t)out * Kp * 4));
        CAPDEC_TRY(t.tB.ensure((size_t)in * Kp * 4));
        CAPDEC_TRY(transpose_pad(c, dy, rows, out, t.tA.as<float>(), Kp));
        CAPDEC_TRY(transpose_pad(c, x, rows, in, t.tB.as<float>(), Kp));
        CAPDEC_TRY(gemm_fp32(c, t.tA.as<float>(), Kp, t.tB.as<float>(), Kp, gW, in, out, in, Kp));
    }
    if (gb)
        hipLaunchKernelGGL(colsum_kernel, dim3((out + 255) / 256, (rows + COLSUM_ROWS - 1) / COLSUM_ROWS), dim3(256), 0, c->stream,
                           dy, rows, out, gb);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// ---- the mapper's forward with everything its backward needs; out = pe [B, P d]
static int mapper_forward_saved(capdec_ctx *c, TrainState &t, const float *x, int B, float *pe) {
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
            if (c->tune.train_attn_blk && hd == 96 && S <= 128 && AttnBlk<96>::fwd_bytes(S) <= ATTN_BLK_LDS_MAX)
                CAPDEC_TRY((attn_blk_fwd<96, false>(st, qkv, att, B, S, m.heads, (float)pow(96.0, -0.5), nullptr, 1.f)));
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
static int mapper_backward(capdec_ctx *c, TrainState &t, const float *x, const float *dy, int B) {
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
    const int nbh = B * m.heads * S;
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
        if (c->tune.train_attn_blk && S <= 128 && AttnBlk<96>::bwd_bytes(S) <= ATTN_BLK_LDS_MAX) {
            CAPDEC_TRY((attn_blk_bwd<96, false>(st, qkv, datt, dqkv, B, S, m.heads, scale, nullptr, 1.f)));
        } else {
            hipLaunchKernelGGL((attn_bwd_q_kernel<96, false>), dim3((nbh + 3) / 4), dim3(256), (size_t)4 * 2 * S * sizeof(float), st,
                               qkv, datt, dqkv, t.lse.as<float>(), t.dsum.as<float>(), nbh, S, m.heads, scale);
            hipLaunchKernelGGL((attn_bwd_kv_kernel<96, false>), dim3((nbh + 3) / 4), dim3(256), 0, st, qkv, datt, dqkv,
                               t.lse.as<float>(), t.dsum.as<float>(), nbh, S, m.heads, scale);
        }
        CAPDEC_TRY(linear_dw(c, t, dqkv, a1, M, 3 * d, d, t.grad(s0 + 2), nullptr));      // [to_queries ; to_keys_values]: no bias
        CAPDEC_TRY(linear_dx(c, t, dqkv, w.wqkv, da, M, 3 * d, d));
        CAPDEC_TRY(ln_bwd(c, h, w.n1w, da, ds2, ds, M, d, 1e-5f, t.grad(s0 + 0), t.grad(s0 + 1)));          // ds = d h
    }
    hipLaunchKernelGGL(tmapper_split_grad_kernel, grid1(std::max((size_t)B * m.clip_len * d, (size_t)m.P * d)), dim3(256), 0, st,
                       ds, dlin, t.grad(2), B, m.clip_len, m.P, d);
    CAPDEC_HIP(hipGetLastError());
    return linear_dw(c, t, dlin, x, B, m.clip_len * d, D, t.grad(0), t.grad(1));
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
    const int nbh = B * g.n_head * S;
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
            if (c->tune.train_attn_blk && S <= 128 && AttnBlk<64>::fwd_bytes(S) <= ATTN_BLK_LDS_MAX)
                CAPDEC_TRY((attn_blk_fwd<64, true>(st, qkv, att, B, S, g.n_head, 0.125f, m_att, inv_keep)));
            else
                hipLaunchKernelGGL(attn_fwd_drop_kernel, dim3((nbh + 3) / 4), dim3(256), (size_t)4 * S * sizeof(float), st, qkv, m_att, att,
                                   nbh, S, g.n_head, 0.125f, inv_keep);
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
        if (c->tune.train_attn_blk && S <= 128 && AttnBlk<64>::bwd_bytes(S) <= ATTN_BLK_LDS_MAX) {
            CAPDEC_TRY((attn_blk_bwd<64, true>(st, qkv, datt, dqkv, B, S, g.n_head, 0.125f, drop ? m_att : nullptr, inv_keep)));
        } else {
            hipLaunchKernelGGL((attn_bwd_q_kernel<64, true>), dim3((nbh + 3) / 4), dim3(256), (size_t)4 * 2 * S * sizeof(float), st, qkv,
                               datt, dqkv, t.lse.as<float>(), t.dsum.as<float>(), nbh, S, g.n_head, 0.125f, drop ? m_att : nullptr, inv_keep);
            hipLaunchKernelGGL((attn_bwd_kv_kernel<64, true>), dim3((nbh + 3) / 4), dim3(256), 0, st, qkv, datt, dqkv,
                               t.lse.as<float>(), t.dsum.as<float>(), nbh, S, g.n_head, 0.125f, drop ? m_att : nullptr, inv_keep);
        }
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
    if (apply_update) {
        hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, st, sc, lr, b1, b2);
        hipLaunchKernelGGL(adamw_multi_kernel, dim3(t.n_chunks), dim3(256), 0, st, t.slotdev.as<SlotDev>(), t.chunkdev.as<int2>(),
                           t.G.as<float>(), t.Mo.as<float>(), t.Vo.as<float>(), sc, b1, b2, eps, lr * weight_decay);
        CAPDEC_HIP(hipGetLastError());
        t.step += 1;
        for (const Slot &sl : t.slots) drop_planes_of(c, sl.p);      // inference must never see planes packed from old values
        if (full) CAPDEC_TRY(refresh_backward_weights(c, t));
    }
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

static TrainState &train_state(capdec_ctx *c) {
    if (!c->train) { c->train = new TrainState(); c->train->train_gpt = c->train_scope != 0; }
    return *c->train;
}

extern "C" {

int capdec_train_get(capdec_ctx *c, int kind, int which, float *d_out, size_t n) {
    CAPDEC_CHECK(c && d_out, "train_get: null argument");
    CAPDEC_CHECK(c->gpt.loaded && (c->map.kind == 1 || c->map.kind == 2), "train_get: needs GPT-2 weights and a mapper");
    CAPDEC_CHECK(kind == 0 || kind == 1, "train_get: kind must be 0 (parameter) or 1 (gradient)");
    CAPDEC_HIP(hipSetDevice(c->device));
    TrainState &t = train_state(c);
    CAPDEC_TRY(build_slots(c, t));
    CAPDEC_CHECK(which >= 0 && which < (int)t.slots.size(), "train_get: tensor index out of range");
    CAPDEC_CHECK(n == t.slots[which].n, "train_get: wrong element count");
    CAPDEC_CHECK(kind == 0 || t.have_grads, "train_get: no gradients yet (run capdec_train_step)");
    const float *src = kind == 0 ? t.slots[which].p : t.grad(which);
    const Slot &sl = t.slots[which];
    if (sl.rows > 0)       // a GPT-2 Conv1D weight: [out, in] on the device, [in, out] in the checkpoint (and for its gradient)
        CAPDEC_TRY(transpose_pad(c, src, sl.rows, sl.cols, d_out, sl.rows));
    else
        CAPDEC_HIP(hipMemcpyAsync(d_out, src, n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    if (kind == 1) {       // the arena holds d loss / d tensor x count x LS (train_step)
        hipLaunchKernelGGL(scale_by_kernel, grid1(n), dim3(256), 0, c->stream, d_out, n, &t.cnt.as<StepScalars>()->gscale);
        CAPDEC_HIP(hipGetLastError());
    }
    CAPDEC_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int capdec_train_set_scope(capdec_ctx *c, int train_gpt) {
    CAPDEC_CHECK(c, "null context");
    CAPDEC_CHECK(train_gpt == 0 || train_gpt == 1, "train_set_scope: 0 (mapper, GPT-2 frozen) or 1 (GPT-2 as well)");
    CAPDEC_HIP(hipSetDevice(c->device));
    if (c->train_scope == train_gpt && (!c->train || c->train->train_gpt == (train_gpt != 0))) return 0;
    c->train_scope = train_gpt;             // (kept by capdec_train_reset and by weight reloads)
    train_release(c);                       // another parameter set: new slots, fresh optimizer state
    return 0;
}

int capdec_train_set_dropout(capdec_ctx *c, float p, uint64_t seed) {
    CAPDEC_CHECK(c, "null context");
    CAPDEC_CHECK(p >= 0.f && p < 1.f, "train_set_dropout: p must lie in [0, 1)");
    c->train_drop_p = p;
    c->train_drop_seed = seed;
    if (c->train) { c->train->draws = 0; c->train->dinj_n = 0; }
    return 0;
}

int capdec_train_set_dropout_masks(capdec_ctx *c, const uint8_t *d_masks, size_t n) {
    CAPDEC_CHECK(c && d_masks && n > 0, "train_set_dropout_masks: null argument");
    CAPDEC_CHECK(c->train_scope == 1 && c->train_drop_p > 0.f, "train_set_dropout_masks: needs scope 1 (GPT-2 in train mode) and a "
                                                                 "dropout probability > 0 (capdec_train_set_dropout)");
    CAPDEC_HIP(hipSetDevice(c->device));
    TrainState &t = train_state(c);
    CAPDEC_TRY(t.dinj.ensure((n + 3) / 4 * 4));
    CAPDEC_HIP(hipMemcpyAsync(t.dinj.p, d_masks, n, hipMemcpyDeviceToDevice, c->stream));
    t.dinj_n = n;
    return 0;
}

int capdec_train_get_dropout_masks(capdec_ctx *c, uint8_t *d_out, size_t n) {
    CAPDEC_CHECK(c && d_out, "train_get_dropout_masks: null argument");
    CAPDEC_CHECK(c->train && c->train->dmask_n > 0, "train_get_dropout_masks: the last train step used no dropout");
    CAPDEC_CHECK(n == c->train->dmask_n, "train_get_dropout_masks: wrong byte count");
    CAPDEC_HIP(hipSetDevice(c->device));
    CAPDEC_HIP(hipMemcpyAsync(d_out, c->train->dmask.p, n, hipMemcpyDeviceToDevice, c->stream));
    CAPDEC_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int capdec_train_loss(capdec_ctx *c, float *last, double *sum, long long *steps, int reset) {
    CAPDEC_CHECK(c, "null context");
    if (!c->train || !c->train->scalars_ready) {       // no step since the last reset: nothing accumulated
        if (last) *last = 0.f;
        if (sum) *sum = 0.0;
        if (steps) *steps = 0;
        return 0;
    }
    CAPDEC_HIP(hipSetDevice(c->device));
    StepScalars h;
    CAPDEC_HIP(hipMemcpyAsync(&h, c->train->cnt.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    CAPDEC_HIP(hipStreamSynchronize(c->stream));
    if (last) *last = h.loss;
    if (sum) *sum = (double)h.loss_sum;
    if (steps) *steps = h.loss_steps;
    if (reset) {
        StepScalars *d = c->train->cnt.as<StepScalars>();
        CAPDEC_HIP(hipMemsetAsync(&d->loss_sum, 0, sizeof(float) + sizeof(int), c->stream));
    }
    return 0;
}

int capdec_train_reset(capdec_ctx *c) {
    CAPDEC_CHECK(c, "null context");
    CAPDEC_HIP(hipSetDevice(c->device));
    train_release(c);                       // (the scope and the dropout setting live in the context: they stay)
    return 0;
}

}  // extern "C"
