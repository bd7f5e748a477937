// The train step with a frozen GPT-2 -- reference train.py:344-354 run with --only_prefix (ClipCaptionPrefix,
// train.py:279-287: parameters() = the mapper's, GPT-2 in eval mode => no dropout, a deterministic step):
//
//     prefix -> MLP mapper -> cat(prefix rows, wte(tokens)) -> GPT-2 -> logits[:, P-1:-1] -> cross_entropy(ignore_index=0)
//     -> backward down to the mapper's four tensors -> transformers-4.24 AdamW
//
// Scope: the MLP mapper (gpt2_prefix.py:114-126).  Structure: the forward keeps every activation the backward needs
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
// x[i] *= 1 / *count   (count > 0)
__global__ void scale_by_count_kernel(float *x, size_t n, const int *__restrict__ count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = x[i] / (float)max(*count, 1);
}
// out[ids[r]] = in[r]   (rows of d floats, d % 4 == 0)
__global__ void scatter_rows_kernel(const float *__restrict__ in, const int *__restrict__ ids, float *__restrict__ out,
                                    int rows, int d4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * d4) return;
    const int r = i / d4, c = i - r * d4;
    reinterpret_cast<float4 *>(out)[(size_t)ids[r] * d4 + c] = reinterpret_cast<const float4 *>(in)[i];
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
// out[j] = sum over rows of x[r][j]
__global__ void colsum_kernel(const float *__restrict__ x, int rows, int n, float *__restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += x[(size_t)r * n + j];
    out[j] = s;
}
// transformers-4.24 AdamW (optimization.py AdamW.step): m, v updated in place; p -= step_size * m / (sqrt(v) + eps);
// then p -= decay * p (decay = lr * weight_decay, 0 by default).  step_size = lr * sqrt(1 - b2^t) / (1 - b1^t) (host)
__global__ void adamw_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                             float *__restrict__ v, size_t n, float step_size, float b1, float b2, float eps, float decay) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = m[i] * b1 + gi * (1.0f - b1);
    const float vi = v[i] * b2 + gi * gi * (1.0f - b2);
    m[i] = mi;
    v[i] = vi;
    float pi = p[i] - step_size * (mi / (sqrtf(vi) + eps));
    if (decay > 0.f) pi -= decay * pi;
    p[i] = pi;
}

// ---------------------------------------------------------------------------------------------- LayerNorm backward
// dx = add + rstd (g - mean(g) - xhat mean(g xhat)), g = dy w; one wavefront per row, d = 64 * NPL
template <int NPL>
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                        const float *__restrict__ dy, const float *add, float *dx,
                                                        int rows, float eps) {
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
}

// ---------------------------------------------------------------------------------------------- attention backward
// Causal softmax attention, head_dim 64, rows = (sample, position) with S positions per sample, qkv rows [q | k | v] of
// 3 d floats.  One wavefront per (sample, head, query i): lane = head dimension.
//   s_j = q_i . k_j / 8, p = softmax_j<=i(s), dP_j = dO_i . v_j, D = sum_j p_j dP_j, dS_j = p_j (dP_j - D)
//   dq_i = sum_j dS_j k_j / 8;  lse_i and D_i are kept for the key-side kernel
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const float *__restrict__ qkv, const float *__restrict__ dout,
                                                         float *__restrict__ dqkv, float *__restrict__ lse_out,
                                                         float *__restrict__ dsum_out, int total, int S, int heads) {
    extern __shared__ float sh[];                     // [4 waves][2][S]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave;
    if (gw >= total) return;
    const int i = gw % S, bh = gw / S, h = bh % heads, b = bh / heads;
    const int d = heads * 64;
    float *sc = sh + (size_t)wave * 2 * S, *dp = sc + S;
    const size_t row = (size_t)b * S + i;
    const float q = qkv[row * 3 * d + h * 64 + lane], go = dout[row * d + h * 64 + lane];
    for (int j = 0; j <= i; ++j) {
        const float *kr = qkv + ((size_t)b * S + j) * 3 * d + d + h * 64;
        const float s = wave_sum(q * kr[lane]) * 0.125f, t = wave_sum(go * kr[d + lane]);
        if (lane == 0) { sc[j] = s; dp[j] = t; }
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
    const float lse = mx + logf(l);
    float D = 0.f;
    for (int j = lane; j <= i; j += 64) D += expf(sc[j] - lse) * dp[j];
    D = wave_sum(D);
    for (int j = lane; j <= i; j += 64) sc[j] = expf(sc[j] - lse) * (dp[j] - D);       // dS_j
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float dq = 0.f;
    for (int j = 0; j <= i; ++j) dq += sc[j] * qkv[((size_t)b * S + j) * 3 * d + d + h * 64 + lane];
    dqkv[row * 3 * d + h * 64 + lane] = dq * 0.125f;
    if (lane == 0) { lse_out[gw] = lse; dsum_out[gw] = D; }
}
// one wavefront per (sample, head, key j): dk_j = sum_{i>=j} dS_ij q_i / 8, dv_j = sum_{i>=j} p_ij dO_i
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(const float *__restrict__ qkv, const float *__restrict__ dout,
                                                          float *__restrict__ dqkv, const float *__restrict__ lse_in,
                                                          const float *__restrict__ dsum_in, int total, int S, int heads) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave;
    if (gw >= total) return;
    const int j = gw % S, bh = gw / S, h = bh % heads, b = bh / heads;
    const int d = heads * 64;
    const size_t rowj = (size_t)b * S + j;
    const float k = qkv[rowj * 3 * d + d + h * 64 + lane], v = qkv[rowj * 3 * d + 2 * d + h * 64 + lane];
    float dk = 0.f, dv = 0.f;
    for (int i = j; i < S; ++i) {
        const size_t rowi = (size_t)b * S + i;
        const float q = qkv[rowi * 3 * d + h * 64 + lane], go = dout[rowi * d + h * 64 + lane];
        const int gi = (bh * S) + i;
        const float p = expf(wave_sum(q * k) * 0.125f - lse_in[gi]);
        const float ds = p * (wave_sum(go * v) - dsum_in[gi]);
        dk += ds * q;
        dv += p * go;
    }
    dqkv[rowj * 3 * d + d + h * 64 + lane] = dk * 0.125f;
    dqkv[rowj * 3 * d + 2 * d + h * 64 + lane] = dv;
}

// ---------------------------------------------------------------------------------------------- cross-entropy
// logits [rows, ld] (columns >= V are padding) -> in place: (softmax - onehot) for rows whose label != ignore, 0 for the
// others and for the padding; row_loss[r] = lse - logit[label] (0 for ignored rows).  One block per row.
__global__ __launch_bounds__(256) void ce_bwd_kernel(float *__restrict__ logits, int ld, const int *__restrict__ labels,
                                                     int V, int ignore_index, float *__restrict__ row_loss) {
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
    for (int c = t; c < ld; c += 256) lr[c] = c < V ? expf(lr[c] - lse) - (c == lab ? 1.f : 0.f) : 0.f;
}
// *count = number of labels != ignore (and in range); *loss = sum(row_loss) / count   (one block)
__global__ __launch_bounds__(256) void ce_finish_kernel(const float *__restrict__ row_loss, const int *__restrict__ labels,
                                                        int rows, int V, int ignore_index, int *__restrict__ count,
                                                        float *__restrict__ loss) {
    __shared__ float rs[4];
    __shared__ int rc[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float s = 0.f;
    int n = 0;
    for (int r = t; r < rows; r += 256) {
        const int lab = labels[r];
        if (lab != ignore_index && lab >= 0 && lab < V) { s += row_loss[r]; ++n; }
    }
    s = wave_sum(s);
    const float nf = wave_sum((float)n);
    if (lane == 0) { rs[wave] = s; rc[wave] = (int)nf; }
    __syncthreads();
    if (t == 0) {
        const int c = rc[0] + rc[1] + rc[2] + rc[3];
        *count = c;
        *loss = ((rs[0] + rs[1]) + (rs[2] + rs[3])) / (float)max(c, 1);
    }
}

// ---------------------------------------------------------------------------------------------- workspace
struct TrainState {
    // transposed copies of the frozen GPT-2 weights: the "[N, K]" operand of dX = dY W^T (= the checkpoint's own Conv1D
    // layout [in, out]); wte_t [d][Vp] zero-padded to a multiple of 64 columns
    struct LayerT { float *wqkv_t, *wproj_t, *wfc_t, *wproj2_t; };
    std::vector<LayerT> lt;
    float *wte_t = nullptr;
    int Vp = 0;
    std::vector<void *> owned;
    bool weights_ready = false;
    // saved activations + gradients (grow-only)
    DBuf pe, emb, hs, a, qkv, att, hmid, fc, gl, hf, hfl, logits, ids, labels, rloss, cnt, loss_dev;
    DBuf dh, dh2, da, dqkv, datt, dfc, dhfl, lse, dsum, dy, dhid, tmp_t, tmp_t2, w2_t;
    DBuf hid;                                // mapper hidden (tanh output) [Bp, hidden]
    DBuf g_w1, g_b1, g_w2, g_b2;             // gradients of the last step
    DBuf m_w1, m_b1, m_w2, m_b2, v_w1, v_b1, v_w2, v_b2;   // AdamW moments
    long long step = 0;                      // updates applied (bias correction uses step + 1)
    bool have_grads = false;
    void release() {
        for (void *p : owned) (void)hipFree(p);
        owned.clear();
        lt.clear();
        wte_t = nullptr;
        weights_ready = false;
        DBuf *bufs[] = {&pe, &emb, &hs, &a, &qkv, &att, &hmid, &fc, &gl, &hf, &hfl, &logits, &ids, &labels, &rloss, &cnt,
                        &loss_dev, &dh, &dh2, &da, &dqkv, &datt, &dfc, &dhfl, &lse, &dsum, &dy, &dhid, &tmp_t, &tmp_t2, &w2_t,
                        &hid, &g_w1, &g_b1, &g_w2, &g_b2, &m_w1, &m_b1, &m_w2, &m_b2, &v_w1, &v_b1, &v_w2, &v_b2};
        for (DBuf *b : bufs) b->release();
        step = 0;
        have_grads = false;
    }
};

void train_release(capdec_ctx *c) {
    if (!c->train) return;
    c->train->release();
    delete c->train;
    c->train = nullptr;
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

// C[M, N] = A[M, K] . Bt[N, K]^T (+ resid) on the native fp32 MFMA GEMM
static int gemm_fp32(capdec_ctx *c, const float *A, int lda, const float *Bt, int ldb, float *C, int ldc, int M, int N, int K,
                     const float *resid = nullptr, int ldr = 0) {
    GemmEpilogue e;
    e.tune = &c->tune;
    e.resid = resid;
    e.ldr = ldr;
    ProfScope ps(c, F_GEMM, 2.0 * M * (double)N * K);
    return launch_gemm_f32(c->stream, A, lda, Bt, ldb, C, ldc, M, N, K, e);
}
static inline dim3 grid1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

static int ln_bwd(capdec_ctx *c, const float *x, const float *w, const float *dy, const float *add, float *dx, int rows,
                  int d, float eps) {
    CAPDEC_CHECK(d == 768, "train: LayerNorm backward is instantiated for d = 768");
    hipLaunchKernelGGL(ln_bwd_dx_kernel<12>, dim3((rows + 3) / 4), dim3(256), 0, c->stream, x, w, dy, add, dx, rows, eps);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// the whole step; see capdec.h: capdec_train_step
static int train_step(capdec_ctx *c, const float *prefix, const int *tokens, int B, int L, float lr, float b1, float b2,
                      float eps, float weight_decay, int apply_update, float *loss_host) {
    const Gpt2 &g = c->gpt;
    Mapper &m = c->map;
    CAPDEC_CHECK(g.loaded && m.kind == 1, "train_step: needs GPT-2 weights and an MLP mapper (capdec_load_mapper_mlp)");
    CAPDEC_CHECK(g.d == 768 && g.d / g.n_head == 64, "train_step: d = 768, head_dim = 64");
    const int d = g.d, P = m.P, S = P + L, R = B * S, Rl = B * L, H = m.hidden, O = P * d, D = m.D;
    CAPDEC_CHECK(B >= 1 && L >= 1 && S <= 256 && S <= g.n_pos, "train_step: bad batch geometry (prefix_length + L <= 256)");
    CAPDEC_CHECK(D % 32 == 0 && H % 32 == 0 && O % 32 == 0, "train_step: mapper dims must be multiples of 32");
    if (!c->train) c->train = new TrainState();
    TrainState &t = *c->train;
    CAPDEC_TRY(prepare_backward_weights(c, t));
    hipStream_t st = c->stream;
    const int nl = g.n_layer, Vp = t.Vp, Bp = (B + 31) / 32 * 32;
    const size_t Rd = (size_t)R * d;
    // ---- buffers
    CAPDEC_TRY(t.hid.ensure((size_t)Bp * H * 4));
    CAPDEC_TRY(t.pe.ensure((size_t)Bp * O * 4));
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
    CAPDEC_TRY(t.ids.ensure((size_t)(Rl + R) * 4));
    CAPDEC_TRY(t.rloss.ensure((size_t)Rl * 4));
    CAPDEC_TRY(t.cnt.ensure(16));
    CAPDEC_TRY(t.dh.ensure(Rd * 4));
    CAPDEC_TRY(t.dh2.ensure(Rd * 4));
    CAPDEC_TRY(t.da.ensure(Rd * 4));
    CAPDEC_TRY(t.dqkv.ensure(Rd * 3 * 4));
    CAPDEC_TRY(t.datt.ensure(Rd * 4));
    CAPDEC_TRY(t.dfc.ensure(Rd * 4 * 4));
    CAPDEC_TRY(t.dhfl.ensure((size_t)Rl * d * 4));
    CAPDEC_TRY(t.lse.ensure((size_t)B * g.n_head * S * 4));
    CAPDEC_TRY(t.dsum.ensure((size_t)B * g.n_head * S * 4));
    CAPDEC_TRY(t.dy.ensure((size_t)Bp * O * 4));
    CAPDEC_TRY(t.dhid.ensure((size_t)Bp * H * 4));
    CAPDEC_TRY(t.tmp_t.ensure((size_t)std::max(O, H) * Bp * 4));
    CAPDEC_TRY(t.tmp_t2.ensure((size_t)std::max(H, D) * Bp * 4));
    CAPDEC_TRY(t.w2_t.ensure((size_t)O * H * 4));
    const size_t n_w1 = (size_t)H * D, n_w2 = (size_t)O * H;
    CAPDEC_TRY(t.g_w1.ensure(n_w1 * 4));
    CAPDEC_TRY(t.g_b1.ensure((size_t)H * 4));
    CAPDEC_TRY(t.g_w2.ensure(n_w2 * 4));
    CAPDEC_TRY(t.g_b2.ensure((size_t)O * 4));
    float *hid = t.hid.as<float>(), *pe = t.pe.as<float>(), *emb = t.emb.as<float>(), *hs = t.hs.as<float>(),
          *a = t.a.as<float>(), *gl = t.gl.as<float>(), *hf = t.hf.as<float>(), *hfl = t.hfl.as<float>(),
          *logits = t.logits.as<float>();
    int *row_ids = t.ids.as<int>();
    int *cnt = t.cnt.as<int>();
    float *loss_dev = reinterpret_cast<float *>(cnt + 1);

    // ---- forward: mapper (current weights: never the cached planes of an earlier step)
    CAPDEC_TRY(gemm(c, prefix, D, m.w1, D, hid, H, B, H, D, m.b1, CAPDEC_ACT_TANH, nullptr, 0, false));
    CAPDEC_TRY(gemm(c, hid, H, m.w2, H, pe, O, B, O, H, m.b2, CAPDEC_ACT_NONE, nullptr, 0, false));
    // embeds = cat(pe.view(B, P, d), wte(tokens)): built on the host side of this function from two row maps
    {
        // rows of `emb`: (b, p < P) <- pe[b, p]; (b, P + t) <- wte[tokens[b, t]]
        for (int b = 0; b < B; ++b) {
            CAPDEC_HIP(hipMemcpyAsync(emb + (size_t)(b * S) * d, pe + (size_t)b * O, (size_t)O * 4, hipMemcpyDeviceToDevice, st));
            ProfScope ps(c, F_EMBED);
            CAPDEC_TRY(launch_gather_rows(st, g.wte, tokens + (size_t)b * L, emb + (size_t)(b * S + P) * d, L, d));
        }
    }
    { ProfScope ps(c, F_EMBED); CAPDEC_TRY(launch_embed_prefix(st, emb, g.wpe, hs, B, S, 0, d)); }
    KvCache kv;
    kv_geometry(kv, B, S, g.n_head, 64);
    kv.tune = &c->tune;
    for (int i = 0; i < nl; ++i) {
        const Gpt2Layer &w = g.layers[i];
        float *h = hs + Rd * i, *hn = hs + Rd * (i + 1);
        float *qkv = t.qkv.as<float>() + Rd * 3 * i, *att = t.att.as<float>() + Rd * i, *hmid = t.hmid.as<float>() + Rd * i,
              *fc = t.fc.as<float>() + Rd * 4 * i;
        { ProfScope ps(c, F_LN); CAPDEC_TRY(launch_layernorm(st, h, d, w.ln1w, w.ln1b, g.eps, a, d, R, d)); }
        CAPDEC_TRY(gemm(c, a, d, w.wqkv, d, qkv, 3 * d, R, 3 * d, d, w.bqkv, CAPDEC_ACT_NONE, nullptr, 0, true));
        { ProfScope ps(c, F_ATTN_PRE); CAPDEC_TRY(launch_attn_prefill(st, qkv, kv, i, B, S, 1, att, true)); }
        CAPDEC_TRY(gemm(c, att, d, w.wproj, d, hmid, d, R, d, d, w.bproj, CAPDEC_ACT_NONE, h, d, true));
        { ProfScope ps(c, F_LN); CAPDEC_TRY(launch_layernorm(st, hmid, d, w.ln2w, w.ln2b, g.eps, a, d, R, d)); }
        CAPDEC_TRY(gemm(c, a, d, w.wfc, d, fc, 4 * d, R, 4 * d, d, w.bfc, CAPDEC_ACT_NONE, nullptr, 0, true));
        hipLaunchKernelGGL(gelu_new_fwd_kernel, grid1(Rd * 4), dim3(256), 0, st, fc, gl, Rd * 4);
        CAPDEC_TRY(gemm(c, gl, 4 * d, w.wproj2, 4 * d, hn, d, R, d, 4 * d, w.bproj2, CAPDEC_ACT_NONE, hmid, d, true));
    }
    float *hL = hs + Rd * nl;
    { ProfScope ps(c, F_LN); CAPDEC_TRY(launch_layernorm(st, hL, d, g.lnfw, g.lnfb, g.eps, hf, d, R, d)); }
    // the rows the loss reads: logits[:, P-1:-1]  ->  row (b, P - 1 + t) predicts tokens[b, t]
    {
        std::vector<int> ids((size_t)Rl);
        for (int b = 0; b < B; ++b)
            for (int tt = 0; tt < L; ++tt) ids[(size_t)b * L + tt] = b * S + P - 1 + tt;
        CAPDEC_HIP(hipMemcpyAsync(row_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, st));
        CAPDEC_HIP(hipStreamSynchronize(st));                       // (`ids` leaves scope)
    }
    { ProfScope ps(c, F_EMBED); CAPDEC_TRY(launch_gather_rows(st, hf, row_ids, hfl, Rl, d)); }
    CAPDEC_HIP(hipMemsetAsync(logits, 0, (size_t)Rl * Vp * 4, st));
    CAPDEC_TRY(gemm(c, hfl, d, g.wte, d, logits, Vp, Rl, g.vocab, d, nullptr, CAPDEC_ACT_NONE, nullptr, 0, true));
    // ---- loss + d logits (unnormalised: softmax - onehot; the 1 / count factor is applied where the mapper's gradient
    // starts, so every GPT-2 backward GEMM sees values of order one)
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(Rl), dim3(256), 0, st, logits, Vp, tokens, g.vocab, 0, t.rloss.as<float>());
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(256), 0, st, t.rloss.as<float>(), tokens, Rl, g.vocab, 0, cnt, loss_dev);
    CAPDEC_HIP(hipGetLastError());
    // ---- backward through the lm_head and ln_f
    float *dh = t.dh.as<float>(), *dh2 = t.dh2.as<float>(), *da = t.da.as<float>(), *dqkv = t.dqkv.as<float>(),
          *datt = t.datt.as<float>(), *dfc = t.dfc.as<float>(), *dhfl = t.dhfl.as<float>();
    CAPDEC_TRY(gemm_fp32(c, logits, Vp, t.wte_t, Vp, dhfl, d, Rl, d, Vp));
    CAPDEC_HIP(hipMemsetAsync(da, 0, Rd * 4, st));
    hipLaunchKernelGGL(scatter_rows_kernel, grid1((size_t)Rl * (d / 4)), dim3(256), 0, st, dhfl, row_ids, da, Rl, d / 4);
    CAPDEC_TRY(ln_bwd(c, hL, g.lnfw, da, nullptr, dh, R, d, g.eps));
    // ---- backward through the blocks (dX only: the GPT-2 weights are frozen)
    const int nbh = B * g.n_head * S;
    for (int i = nl - 1; i >= 0; --i) {
        const Gpt2Layer &w = g.layers[i];
        const TrainState::LayerT &wt = t.lt[i];
        float *h = hs + Rd * i;
        float *qkv = t.qkv.as<float>() + Rd * 3 * i, *hmid = t.hmid.as<float>() + Rd * i, *fc = t.fc.as<float>() + Rd * 4 * i;
        CAPDEC_TRY(gemm_fp32(c, dh, d, wt.wproj2_t, d, dfc, 4 * d, R, 4 * d, d));                 // d gelu_out = dh Wproj2^T
        hipLaunchKernelGGL(gelu_new_bwd_kernel, grid1(Rd * 4), dim3(256), 0, st, fc, dfc, dfc, Rd * 4);
        CAPDEC_TRY(gemm_fp32(c, dfc, 4 * d, wt.wfc_t, 4 * d, da, d, R, d, 4 * d));                // d a2
        CAPDEC_TRY(ln_bwd(c, hmid, w.ln2w, da, dh, dh2, R, d, g.eps));                            // dh_mid = dh + LN'(..)
        CAPDEC_TRY(gemm_fp32(c, dh2, d, wt.wproj_t, d, datt, d, R, d, d));                        // d att
        hipLaunchKernelGGL(attn_bwd_q_kernel, dim3((nbh + 3) / 4), dim3(256), (size_t)4 * 2 * S * sizeof(float), st, qkv, datt,
                           dqkv, t.lse.as<float>(), t.dsum.as<float>(), nbh, S, g.n_head);
        hipLaunchKernelGGL(attn_bwd_kv_kernel, dim3((nbh + 3) / 4), dim3(256), 0, st, qkv, datt, dqkv, t.lse.as<float>(),
                           t.dsum.as<float>(), nbh, S, g.n_head);
        CAPDEC_TRY(gemm_fp32(c, dqkv, 3 * d, wt.wqkv_t, 3 * d, da, d, R, d, 3 * d));              // d a1
        CAPDEC_TRY(ln_bwd(c, h, w.ln1w, da, dh2, dh, R, d, g.eps));                               // dh = dh_mid + LN'(..)
    }
    CAPDEC_HIP(hipGetLastError());
    // ---- the mapper: dY = d embeds[:, :P] / count  (rows b of [Bp, O]; the padding rows stay zero)
    float *dy = t.dy.as<float>(), *dhid = t.dhid.as<float>(), *tA = t.tmp_t.as<float>(), *tB = t.tmp_t2.as<float>();
    CAPDEC_HIP(hipMemsetAsync(dy, 0, (size_t)Bp * O * 4, st));
    for (int b = 0; b < B; ++b)
        CAPDEC_HIP(hipMemcpyAsync(dy + (size_t)b * O, dh + (size_t)(b * S) * d, (size_t)O * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(scale_by_count_kernel, grid1((size_t)B * O), dim3(256), 0, st, dy, (size_t)B * O, cnt);
    // d W2 = dY^T hid  ([O, H], K = batch), d b2 = colsum(dY)
    if (Bp > B) CAPDEC_HIP(hipMemsetAsync(hid + (size_t)B * H, 0, (size_t)(Bp - B) * H * 4, st));
    CAPDEC_TRY(transpose_pad(c, dy, Bp, O, tA, Bp));                                             // [O, Bp]
    CAPDEC_TRY(transpose_pad(c, hid, Bp, H, tB, Bp));                                            // [H, Bp]
    CAPDEC_TRY(gemm_fp32(c, tA, Bp, tB, Bp, t.g_w2.as<float>(), H, O, H, Bp));
    hipLaunchKernelGGL(colsum_kernel, grid1(O), dim3(256), 0, st, dy, B, O, t.g_b2.as<float>());
    // d hid = dY W2 (W2 [O, H]: the "[N, K]" operand is W2^T [H, O]); d pre-tanh = d hid (1 - hid^2)
    CAPDEC_TRY(transpose_pad(c, m.w2, O, H, t.w2_t.as<float>(), O));                              // [H, O]
    CAPDEC_TRY(gemm_fp32(c, dy, O, t.w2_t.as<float>(), O, dhid, H, Bp, H, O));
    hipLaunchKernelGGL(tanh_bwd_kernel, grid1((size_t)Bp * H), dim3(256), 0, st, hid, dhid, dhid, (size_t)Bp * H);
    // d W1 = d pre^T x  ([H, D]), d b1 = colsum
    CAPDEC_TRY(transpose_pad(c, dhid, Bp, H, tA, Bp));                                           // [H, Bp]
    CAPDEC_TRY(transpose_pad(c, prefix, B, D, tB, Bp));                                          // [D, Bp] (zero padded)
    CAPDEC_TRY(gemm_fp32(c, tA, Bp, tB, Bp, t.g_w1.as<float>(), D, H, D, Bp));
    hipLaunchKernelGGL(colsum_kernel, grid1(H), dim3(256), 0, st, dhid, B, H, t.g_b1.as<float>());
    CAPDEC_HIP(hipGetLastError());
    t.have_grads = true;
    // ---- AdamW (transformers 4.24 semantics)
    if (apply_update) {
        const size_t sizes[4] = {n_w1, (size_t)H, n_w2, (size_t)O};
        DBuf *ms[4] = {&t.m_w1, &t.m_b1, &t.m_w2, &t.m_b2}, *vs[4] = {&t.v_w1, &t.v_b1, &t.v_w2, &t.v_b2};
        DBuf *gs[4] = {&t.g_w1, &t.g_b1, &t.g_w2, &t.g_b2};
        float *ps[4] = {m.w1, m.b1, m.w2, m.b2};
        const double tt = (double)(t.step + 1);
        const float step_size = (float)((double)lr * std::sqrt(1.0 - std::pow((double)b2, tt)) / (1.0 - std::pow((double)b1, tt)));
        for (int k = 0; k < 4; ++k) {
            if (ms[k]->cap < sizes[k] * 4) {                       // first update: zero moments
                CAPDEC_TRY(ms[k]->ensure(sizes[k] * 4));
                CAPDEC_TRY(vs[k]->ensure(sizes[k] * 4));
                CAPDEC_HIP(hipMemsetAsync(ms[k]->p, 0, sizes[k] * 4, st));
                CAPDEC_HIP(hipMemsetAsync(vs[k]->p, 0, sizes[k] * 4, st));
            }
            hipLaunchKernelGGL(adamw_kernel, grid1(sizes[k]), dim3(256), 0, st, ps[k], gs[k]->as<float>(), ms[k]->as<float>(),
                               vs[k]->as<float>(), sizes[k], step_size, b1, b2, eps, lr * weight_decay);
        }
        CAPDEC_HIP(hipGetLastError());
        t.step += 1;
        drop_planes_of(c, m.w1);              // inference must never see planes packed from the old weights
        drop_planes_of(c, m.w2);
    }
    if (loss_host) {
        CAPDEC_HIP(hipMemcpyAsync(loss_host, loss_dev, sizeof(float), hipMemcpyDeviceToHost, st));
    }
    CAPDEC_HIP(hipStreamSynchronize(st));
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

// which: 0 model.0.weight [hidden, D], 1 model.0.bias, 2 model.2.weight [P d, hidden], 3 model.2.bias
static int mapper_tensor(capdec_ctx *c, int which, float **p, size_t *n) {
    Mapper &m = c->map;
    CAPDEC_CHECK(m.kind == 1, "train: MLP mapper not loaded");
    const size_t O = (size_t)m.P * m.d;
    switch (which) {
        case 0: *p = m.w1; *n = (size_t)m.hidden * m.D; return 0;
        case 1: *p = m.b1; *n = (size_t)m.hidden; return 0;
        case 2: *p = m.w2; *n = O * m.hidden; return 0;
        case 3: *p = m.b2; *n = O; return 0;
        default: CAPDEC_CHECK(false, "train: tensor index must be 0..3");
    }
}

int capdec_train_get(capdec_ctx *c, int kind, int which, float *d_out, size_t n) {
    CAPDEC_CHECK(c && d_out, "train_get: null argument");
    CAPDEC_HIP(hipSetDevice(c->device));
    float *p = nullptr;
    size_t cnt = 0;
    CAPDEC_TRY(mapper_tensor(c, which, &p, &cnt));
    CAPDEC_CHECK(n == cnt, "train_get: wrong element count");
    const void *src = p;
    if (kind == 1) {
        CAPDEC_CHECK(c->train && c->train->have_grads, "train_get: no gradients yet (run capdec_train_step)");
        DBuf *gs[4] = {&c->train->g_w1, &c->train->g_b1, &c->train->g_w2, &c->train->g_b2};
        src = gs[which]->p;
    } else {
        CAPDEC_CHECK(kind == 0, "train_get: kind must be 0 (parameter) or 1 (gradient)");
    }
    CAPDEC_HIP(hipMemcpyAsync(d_out, src, n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    CAPDEC_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int capdec_train_reset(capdec_ctx *c) {
    CAPDEC_CHECK(c, "null context");
    CAPDEC_HIP(hipSetDevice(c->device));
    train_release(c);
    return 0;
}

}  // extern "C"
