// Train step, the pieces the mapper's and GPT-2's passes share (train.h): LayerNorm backward, attention backward (per-query
// wavefront kernels and the block-per-(sample, head) form), GPT-2's attention forward with dropout, transposes / column sums,
// and the dX / dW products on top of the inference path's GEMMs.
#include "train.h"

namespace capdec {

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


// ---------------------------------------------------------------------------------------------- host side
int transpose_pad(capdec_ctx *c, const float *src, int rows, int cols, float *dst, int ld) {
    hipLaunchKernelGGL(transpose_pad_kernel, dim3((cols + 31) / 32, (ld + 31) / 32), dim3(256), 0, c->stream, src, rows, cols,
                       dst, ld);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// C[M, N] = A[M, K] . Bt[N, K]^T on the fp32-accurate two-fp16-plane kernels of the inference path (the backward pass runs
// on un-normalised, loss-scaled gradients for exactly that: see the header); CAPDEC_TRAIN_F16X2=0: the native fp32 MFMA
// GEMM instead (39.4 vs 23.6 ms per step at the reference's default geometry, profiles/r5_first_call.txt).
// static_weight: Bt never changes, its packed planes may be cached
int gemm_fp32(capdec_ctx *c, const float *A, int lda, const float *Bt, int ldb, float *C, int ldc, int M, int N, int K,
                     bool static_weight) {
    if (c->tune.train_f16x2 && K % 64 == 0 && ldb == K && lda % 4 == 0)
        return gemm(c, A, lda, Bt, ldb, C, ldc, M, N, K, nullptr, CAPDEC_ACT_NONE, nullptr, 0, static_weight);
    GemmEpilogue e;
    e.tune = &c->tune;
    ProfScope ps(c, F_GEMM, 2.0 * M * (double)N * K);
    return launch_gemm_f32(c->stream, A, lda, Bt, ldb, C, ldc, M, N, K, e);
}

int ln_bwd(capdec_ctx *c, const float *x, const float *w, const float *dy, const float *add, float *dx, int rows,
                  int d, float eps, float *gw, float *gb) {
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
int linear_dx(capdec_ctx *c, TrainState &t, const float *dy, const float *W, float *dx, int M, int out, int in) {
    CAPDEC_TRY(t.wT.ensure((size_t)out * in * 4));
    CAPDEC_TRY(transpose_pad(c, W, out, in, t.wT.as<float>(), out));               // [in][out]
    return gemm_fp32(c, dy, out, t.wT.as<float>(), out, dx, in, M, in, out);
}
// dW = dY^T X ([out, in]; dY [rows, out], X [rows, in]; the GEMM's K = rows, zero-padded to a multiple of 32), db = colsum(dY)
int linear_dw(capdec_ctx *c, TrainState &t, const float *dy, const float *x, int rows, int out, int in, float *gW,
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

int train_attn_fwd(capdec_ctx *c, const float *qkv, float *out, int B, int S, int heads, int hd, bool causal, float scale,
                   const uint8_t *mask, float inv_keep) {
    hipStream_t st = c->stream;
    const int d = heads * hd;
    if (hd == 96 && !causal) {
        if (c->tune.train_attn_blk && S <= 128 && AttnBlk<96>::fwd_bytes(S) <= ATTN_BLK_LDS_MAX)
            return attn_blk_fwd<96, false>(st, qkv, out, B, S, heads, scale, nullptr, 1.f);
        return launch_attn_mapper(st, qkv, 3 * d, qkv + d, qkv + 2 * d, 3 * d, out, B, S, heads, hd);
    }
    CAPDEC_CHECK(hd == 64 && causal && mask, "train: attention forward forms are (96, bidirectional) and (64, causal, dropout)");
    if (c->tune.train_attn_blk && S <= 128 && AttnBlk<64>::fwd_bytes(S) <= ATTN_BLK_LDS_MAX)
        return attn_blk_fwd<64, true>(st, qkv, out, B, S, heads, scale, mask, inv_keep);
    const int nbh = B * heads * S;
    hipLaunchKernelGGL(attn_fwd_drop_kernel, dim3((nbh + 3) / 4), dim3(256), (size_t)4 * S * sizeof(float), st, qkv, mask, out, nbh, S,
                       heads, scale, inv_keep);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

int train_attn_bwd(capdec_ctx *c, TrainState &t, const float *qkv, const float *dout, float *dqkv, int B, int S, int heads, int hd,
                   bool causal, float scale, const uint8_t *mask, float inv_keep) {
    hipStream_t st = c->stream;
    const int nbh = B * heads * S;
    if (hd == 96 && !causal) {
        if (c->tune.train_attn_blk && S <= 128 && AttnBlk<96>::bwd_bytes(S) <= ATTN_BLK_LDS_MAX)
            return attn_blk_bwd<96, false>(st, qkv, dout, dqkv, B, S, heads, scale, nullptr, 1.f);
        hipLaunchKernelGGL((attn_bwd_q_kernel<96, false>), dim3((nbh + 3) / 4), dim3(256), (size_t)4 * 2 * S * sizeof(float), st, qkv, dout,
                           dqkv, t.lse.as<float>(), t.dsum.as<float>(), nbh, S, heads, scale);
        hipLaunchKernelGGL((attn_bwd_kv_kernel<96, false>), dim3((nbh + 3) / 4), dim3(256), 0, st, qkv, dout, dqkv, t.lse.as<float>(),
                           t.dsum.as<float>(), nbh, S, heads, scale);
        CAPDEC_HIP(hipGetLastError());
        return 0;
    }
    CAPDEC_CHECK(hd == 64 && causal, "train: attention backward forms are (96, bidirectional) and (64, causal)");
    if (c->tune.train_attn_blk && S <= 128 && AttnBlk<64>::bwd_bytes(S) <= ATTN_BLK_LDS_MAX)
        return attn_blk_bwd<64, true>(st, qkv, dout, dqkv, B, S, heads, scale, mask, inv_keep);
    hipLaunchKernelGGL((attn_bwd_q_kernel<64, true>), dim3((nbh + 3) / 4), dim3(256), (size_t)4 * 2 * S * sizeof(float), st, qkv, dout, dqkv,
                       t.lse.as<float>(), t.dsum.as<float>(), nbh, S, heads, scale, mask, inv_keep);
    hipLaunchKernelGGL((attn_bwd_kv_kernel<64, true>), dim3((nbh + 3) / 4), dim3(256), 0, st, qkv, dout, dqkv, t.lse.as<float>(),
                       t.dsum.as<float>(), nbh, S, heads, scale, mask, inv_keep);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

}  // namespace capdec
