// Memory-bound row kernels: LayerNorm, embedding gathers, prefix normalise / noise injection,
// TransformerMapper sequence assembly, weight transposes.  One wavefront per row wherever a
// row reduction is needed (64-lane shuffle reductions, float4 accesses).
#include "bf16x3.h"
#include "common.h"

namespace capdec {

// ---------------------------------------------------------------------------- LayerNorm
// y = (x - mean) * rsqrt(var + eps) * w + b, biased variance, two-pass in registers.
// One wavefront per row; d <= 64 * 4 * LN_MAXV.
constexpr int LN_MAXV = 4;   // float4s per lane -> d <= 1024
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x, int ldx,
                                                        const float *__restrict__ w, const float *__restrict__ b,
                                                        float eps, float *__restrict__ y, int ldy, int rows, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + (size_t)row * ldx;
    const int nv = d >> 2;   // float4 count
    float4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            v[i] = reinterpret_cast<const float4 *>(xr)[idx];
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
            q += (a * a + bb * bb) + (c * c + e * e);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
    float *yr = y + (size_t)row * ldy;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float4 ww = reinterpret_cast<const float4 *>(w)[idx];
            const float4 bb = reinterpret_cast<const float4 *>(b)[idx];
            float4 o;
            o.x = (v[i].x - mean) * rstd * ww.x + bb.x;
            o.y = (v[i].y - mean) * rstd * ww.y + bb.y;
            o.z = (v[i].z - mean) * rstd * ww.z + bb.z;
            o.w = (v[i].w - mean) * rstd * ww.w + bb.w;
            reinterpret_cast<float4 *>(yr)[idx] = o;
        }
    }
}

int launch_layernorm(hipStream_t st, const float *x, int ldx, const float *w, const float *b, float eps, float *y,
                     int ldy, int rows, int d) {
    CAPDEC_CHECK(d % 4 == 0 && d <= 256 * LN_MAXV && ldx % 4 == 0 && ldy % 4 == 0, "layernorm: unsupported width");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, ldx, w, b, eps, y, ldy, rows, d);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// Same LayerNorm, output written as the packed split-bf16 A operand of the next GEMM (bf16x3.h): the fp32
// normalised row never goes to HBM; float4 index idx of the row is k-step idx/4, quad idx%4.
// (each wavefront walks LN_RPW consecutive rows: the loads of the next row are in flight while the current one is
//  reduced and stored -- one row per wavefront left the kernel at 3.9 TB/s, short-lived waves and no overlap)
template <int LN_RPW>
__global__ __launch_bounds__(256) void layernorm_packed_kernel(const float *__restrict__ x, int ldx,
                                                               const float *__restrict__ w, const float *__restrict__ b,
                                                               float eps, char *__restrict__ packed, int rows, int d,
                                                               int fmt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = d >> 2, nk = d / X3_BK;
    const int row0 = blockIdx.x * 4 * LN_RPW + wave;     // rows row0, row0 + 4, ...: at any time the block's four waves
    if (row0 >= rows) return;                            // write four ADJACENT rows = whole 128-B lines of a packed plane
    float4 ww[LN_MAXV], bb[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int idx = lane + 64 * i;
        ww[i] = idx < nv ? reinterpret_cast<const float4 *>(w)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        bb[i] = idx < nv ? reinterpret_cast<const float4 *>(b)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 v[LN_MAXV], nx[LN_MAXV];
#define LN_LOAD(dst, r)                                                                              \
    _Pragma("unroll") for (int i = 0; i < LN_MAXV; ++i) {                                            \
        const int idx = lane + 64 * i;                                                               \
        dst[i] = (idx < nv && (r) < rows) ? reinterpret_cast<const float4 *>(x + (size_t)(r) * ldx)[idx] \
                                          : make_float4(0.f, 0.f, 0.f, 0.f);                         \
    }
    LN_LOAD(nx, row0)
#pragma unroll
    for (int rr = 0; rr < LN_RPW; ++rr) {
        const int row = row0 + 4 * rr;
        if (row >= rows) break;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) v[i] = nx[i];
        if (rr + 1 < LN_RPW) { LN_LOAD(nx, row + 4) }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);     // (lanes past nv hold zeros)
        const float mean = wave_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int idx = lane + 64 * i;
            if (idx < nv) {
                const float a = v[i].x - mean, b2 = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
                q += (a * a + b2 * b2) + (c * c + e * e);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int idx = lane + 64 * i;
            if (idx < nv) {
                float4 o;
                o.x = (v[i].x - mean) * rstd * ww[i].x + bb[i].x;
                o.y = (v[i].y - mean) * rstd * ww[i].y + bb[i].y;
                o.z = (v[i].z - mean) * rstd * ww[i].z + bb[i].z;
                o.w = (v[i].w - mean) * rstd * ww[i].w + bb[i].w;
                x3_store_quad(packed, nk, row, idx >> 2, idx & 3, o, fmt);
            }
        }
    }
#undef LN_LOAD
}

int launch_layernorm_packed(hipStream_t st, const float *x, int ldx, const float *w, const float *b, float eps,
                            void *packed, int rows, int d, int fmt) {
    CAPDEC_CHECK(d % 16 == 0 && d <= 256 * LN_MAXV && ldx % 4 == 0, "layernorm_packed: unsupported width");
    if (rows <= 0) return 0;
    if (rows >= 16384)      // 4 rows per wavefront once that still leaves >> 256 blocks (25 000 rows: 39.5 -> 37.5 us);
        hipLaunchKernelGGL(layernorm_packed_kernel<4>, dim3((rows + 15) / 16), dim3(256), 0, st, x, ldx, w, b, eps,
                           (char *)packed, rows, d, fmt);
    else                    // small batches need every block they can get (3125 rows: 9.8 us with 1, 12.7 us with 4)
        hipLaunchKernelGGL(layernorm_packed_kernel<1>, dim3((rows + 3) / 4), dim3(256), 0, st, x, ldx, w, b, eps,
                           (char *)packed, rows, d, fmt);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------- embeddings
// h[row] = wte[tok[row]] + wpe_row   (all rows of a decode step share one position)
// (cmap: compact activation row -> the original caption whose token it carries, see launch_attn_decode)
__global__ void embed_tokens_kernel(const int *__restrict__ tok, const float *__restrict__ wte,
                                    const float *__restrict__ wpe_row, float *__restrict__ h, int rows, int nv,
                                    const int *__restrict__ cmap, int beam) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * nv) return;
    const int row = i / nv, c = i - row * nv;
    const int srow = cmap ? cmap[row / beam] * beam + row % beam : row;
    const float4 a = reinterpret_cast<const float4 *>(wte + (size_t)tok[srow] * nv * 4)[c];
    const float4 p = reinterpret_cast<const float4 *>(wpe_row)[c];
    reinterpret_cast<float4 *>(h)[i] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
}
int launch_embed_tokens(hipStream_t st, const int *tok, const float *wte, const float *wpe_row, float *h, int rows,
                        int d, const int *cmap, int beam) {
    if (rows <= 0) return 0;
    const int n = rows * (d / 4);
    hipLaunchKernelGGL(embed_tokens_kernel, dim3((n + 255) / 256), dim3(256), 0, st, tok, wte, wpe_row, h, rows,
                       d / 4, cmap, beam);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// h[(c, i)] = prefix[(c, i)] + wpe[pos0 + i]
__global__ void embed_prefix_kernel(const float *__restrict__ prefix, const float *__restrict__ wpe,
                                    float *__restrict__ h, int n, int P, int pos0, int nv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * P * nv) return;
    const int c = i % nv, p = (i / nv) % P;
    const float4 a = reinterpret_cast<const float4 *>(prefix)[i];
    const float4 w = reinterpret_cast<const float4 *>(wpe + (size_t)(pos0 + p) * nv * 4)[c];
    reinterpret_cast<float4 *>(h)[i] = make_float4(a.x + w.x, a.y + w.y, a.z + w.z, a.w + w.w);
}
int launch_embed_prefix(hipStream_t st, const float *prefix, const float *wpe, float *h, int n, int P, int pos0,
                        int d) {
    const int tot = n * P * (d / 4);
    if (tot <= 0) return 0;
    hipLaunchKernelGGL(embed_prefix_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, prefix, wpe, h, n, P, pos0,
                       d / 4);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

__global__ void gather_rows_kernel(const float *__restrict__ table, const int *__restrict__ ids,
                                   float *__restrict__ out, int rows, int nv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * nv) return;
    const int row = i / nv, c = i - row * nv;
    reinterpret_cast<float4 *>(out)[i] = reinterpret_cast<const float4 *>(table + (size_t)ids[row] * nv * 4)[c];
}
int launch_gather_rows(hipStream_t st, const float *table, const int *ids, float *out, int rows, int d) {
    if (rows <= 0) return 0;
    const int n = rows * (d / 4);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, st, table, ids, out, rows, d / 4);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------- prefix stage
// One wavefront per row; dim arbitrary (scalar strided loop; rows are 512 / 640 floats).
__global__ __launch_bounds__(256) void normalize_prefix_kernel(const float *__restrict__ x, int n, int dim,
                                                               int normalize, const float *__restrict__ offset,
                                                               float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float *xr = x + (size_t)row * dim;
    float s = 0.f;
    for (int i = lane; i < dim; i += 64) s += xr[i] * xr[i];
    const float nrm = sqrtf(wave_sum(s));
    for (int i = lane; i < dim; i += 64) {
        float v = xr[i];
        if (normalize) v = v / nrm;               // reference: prefix / prefix.norm(2, -1), no eps
        if (offset) v += offset[i];
        out[(size_t)row * dim + i] = v;
    }
}
int launch_normalize_prefix(hipStream_t st, const float *x, int n, int dim, int normalize, const float *offset,
                            float *out) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(normalize_prefix_kernel, dim3((n + 3) / 4), dim3(256), 0, st, x, n, dim, normalize, offset,
                       out);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// (philox4x32: common.h)
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }
__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t idx, uint32_t stream) {
    uint32_t o[4];
    philox4x32((uint32_t)idx, (uint32_t)(idx >> 32), stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    const float r = sqrtf(-2.0f * logf(u01(o[0])));
    return r * cosf(6.283185307179586f * u01(o[1]));   // Box-Muller
}
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t idx, uint32_t stream) {
    uint32_t o[4];
    philox4x32((uint32_t)idx, (uint32_t)(idx >> 32), stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    return ((float)(o[0] >> 8)) * (1.0f / 16777216.0f);   // [0, 1) like torch.rand
}

// reference train.py:27-39 (variance != 0): normalise -> + noise -> + offset -> normalise.
// F.normalize: x / max(||x||, 1e-12).  One wavefront per row, row kept in LDS-free registers
// by re-reading x (rows are <= 2.5 KB, L1-resident).
__global__ __launch_bounds__(256) void noise_inject_kernel(const float *__restrict__ x, int n, int dim, float std,
                                                           const float *__restrict__ offset, int uniform,
                                                           int dont_norm, uint64_t seed,
                                                           const float *__restrict__ noise,
                                                           const float *__restrict__ u, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float *xr = x + (size_t)row * dim;
    float inv1 = 1.f;
    if (!dont_norm) {
        float s = 0.f;
        for (int i = lane; i < dim; i += 64) s += xr[i] * xr[i];
        inv1 = 1.0f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    }
    // noise scale: Gaussian -> std; uniform ball -> u^(1/dim) * std / max(||g||, 1e-12)
    float nscale = std;
    if (uniform) {
        float s = 0.f;
        for (int i = lane; i < dim; i += 64) {
            const float g = noise ? noise[(size_t)row * dim + i] : philox_normal(seed, (uint64_t)row * dim + i, 0u);
            s += g * g;
        }
        const float gn = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
        const float uu = u ? u[row] : philox_uniform(seed, (uint64_t)row, 1u);
        nscale = powf(uu, 1.0f / (float)dim) * std / gn;
    }
    float s2 = 0.f;
    for (int i = lane; i < dim; i += 64) {
        const float g = noise ? noise[(size_t)row * dim + i] : philox_normal(seed, (uint64_t)row * dim + i, 0u);
        float v = (dont_norm ? xr[i] : xr[i] * inv1);
        v = v + g * nscale;
        if (offset) v += offset[i];
        out[(size_t)row * dim + i] = v;
        s2 += v * v;
    }
    const float inv2 = 1.0f / fmaxf(sqrtf(wave_sum(s2)), 1e-12f);
    for (int i = lane; i < dim; i += 64) out[(size_t)row * dim + i] *= inv2;
}
int launch_noise_inject(hipStream_t st, const float *x, int n, int dim, float variance, const float *offset,
                        int uniform, int dont_norm, uint64_t seed, const float *noise, const float *u, float *out) {
    if (n <= 0) return 0;
    if (variance == 0.0f) {   // reference returns x unchanged (train.py:28-29)
        if (out != x) CAPDEC_HIP(hipMemcpyAsync(out, x, (size_t)n * dim * sizeof(float), hipMemcpyDeviceToDevice, st));
        return 0;
    }
    CAPDEC_CHECK(out != x, "noise_inject: in-place not supported");
    hipLaunchKernelGGL(noise_inject_kernel, dim3((n + 3) / 4), dim3(256), 0, st, x, n, dim, sqrtf(variance), offset,
                       uniform, dont_norm, seed, noise, u, out);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------- TransformerMapper glue
// seq[c, 0:clip_len] = lin[c].view(clip_len, d); seq[c, clip_len:] = prefix_const
__global__ void tmapper_concat_kernel(const float *__restrict__ lin, const float *__restrict__ pc,
                                      float *__restrict__ seq, int n, int clip_len, int P, int nv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int S = clip_len + P;
    if (i >= n * S * nv) return;
    const int c = i % nv, s = (i / nv) % S, cap = i / (nv * S);
    float4 v;
    if (s < clip_len) v = reinterpret_cast<const float4 *>(lin)[((size_t)cap * clip_len + s) * nv + c];
    else v = reinterpret_cast<const float4 *>(pc)[(size_t)(s - clip_len) * nv + c];
    reinterpret_cast<float4 *>(seq)[i] = v;
}
int launch_tmapper_concat(hipStream_t st, const float *lin, const float *prefix_const, float *seq, int n,
                          int clip_len, int P, int d) {
    const int tot = n * (clip_len + P) * (d / 4);
    if (tot <= 0) return 0;
    hipLaunchKernelGGL(tmapper_concat_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, lin, prefix_const, seq, n,
                       clip_len, P, d / 4);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}
__global__ void tmapper_take_kernel(const float *__restrict__ seq, float *__restrict__ out, int n, int clip_len,
                                    int P, int nv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * P * nv) return;
    const int c = i % nv, p = (i / nv) % P, cap = i / (nv * P);
    reinterpret_cast<float4 *>(out)[i] =
        reinterpret_cast<const float4 *>(seq)[((size_t)cap * (clip_len + P) + clip_len + p) * nv + c];
}
int launch_tmapper_take(hipStream_t st, const float *seq, float *out, int n, int clip_len, int P, int d) {
    const int tot = n * P * (d / 4);
    if (tot <= 0) return 0;
    hipLaunchKernelGGL(tmapper_take_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, seq, out, n, clip_len, P,
                       d / 4);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// out[c][r] = in[r][c], 32x32 LDS tiles (+1 pad), used once per Conv1D weight at load time.
__global__ void transpose_kernel(const float *__restrict__ in, float *__restrict__ out, int rows, int cols) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int r = by + j, c = bx + tx;
        if (r < rows && c < cols) tile[j][tx] = in[(size_t)r * cols + c];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = bx + j, r = by + tx;
        if (r < rows && c < cols) out[(size_t)c * rows + r] = tile[tx][j];
    }
}
int launch_transpose(hipStream_t st, const float *in, float *out, int rows, int cols) {
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, st, in, out, rows,
                       cols);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------- CLIP glue
// h[(i, t)] = token_embedding[tokens[cap(i), t]] + positional_embedding[t] for t < P (P <= L: the positions up to the chunk's
// last EOT -- nothing behind a caption's EOT can reach its feature through causal attention); cap(i) = perm ? perm[i] : i
__global__ void clip_text_embed_kernel(const int *__restrict__ tokens, const float *__restrict__ tok_emb,
                                       const float *__restrict__ pos_emb, float *__restrict__ h, int n, int L, int P,
                                       const int *__restrict__ perm, int nv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * P * nv) return;
    const int c = i % nv, row = i / nv, t = row % P, cap = row / P;
    const int tok = tokens[(size_t)(perm ? perm[cap] : cap) * L + t];
    const float4 a = reinterpret_cast<const float4 *>(tok_emb + (size_t)tok * nv * 4)[c];
    const float4 p = reinterpret_cast<const float4 *>(pos_emb + (size_t)t * nv * 4)[c];
    reinterpret_cast<float4 *>(h)[i] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
}
int launch_clip_text_embed(hipStream_t st, const int *tokens, const float *tok_emb, const float *pos_emb, float *h,
                           int n, int L, int d, int P, const int *perm) {
    if (P <= 0) P = L;
    const int tot = n * P * (d / 4);
    if (tot <= 0) return 0;
    hipLaunchKernelGGL(clip_text_embed_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, tokens, tok_emb, pos_emb, h, n,
                       L, P, perm, d / 4);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// flat_idx[n] = n*L + argmax_t tokens[n, t]  (first maximum, like torch.argmax: the EOT position); flat = 0: the bare position
__global__ void eot_index_kernel(const int *__restrict__ tokens, int *__restrict__ flat_idx, int n, int L, int flat) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int best = tokens[(size_t)r * L], bi = 0;
    for (int t = 1; t < L; ++t) {
        const int v = tokens[(size_t)r * L + t];
        if (v > best) { best = v; bi = t; }
    }
    flat_idx[r] = (flat ? r * L : 0) + bi;
}
int launch_eot_index(hipStream_t st, const int *tokens, int *flat_idx, int n, int L, int flat) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(eot_index_kernel, dim3((n + 255) / 256), dim3(256), 0, st, tokens, flat_idx, n, L, flat);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}
// rows[i] = i P + pos[perm[i]]: the EOT row of caption perm[i] inside a chunk computed with P positions per caption
__global__ void eot_rows_kernel(const int *__restrict__ pos, const int *__restrict__ perm, int *__restrict__ rows, int m, int P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) rows[i] = i * P + pos[perm[i]];
}
int launch_eot_rows(hipStream_t st, const int *pos, const int *perm, int *rows, int m, int P) {
    if (m <= 0) return 0;
    hipLaunchKernelGGL(eot_rows_kernel, dim3((m + 255) / 256), dim3(256), 0, st, pos, perm, rows, m, P);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}
// dst[perm[i], :] = src[i, :]   (d a multiple of 4)
__global__ void scatter_rows_kernel(const float *__restrict__ src, const int *__restrict__ perm, float *__restrict__ dst, int rows, int nv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * nv) return;
    const int row = i / nv, c = i - row * nv;
    reinterpret_cast<float4 *>(dst + (size_t)perm[row] * nv * 4)[c] = reinterpret_cast<const float4 *>(src)[i];
}
int launch_scatter_rows(hipStream_t st, const float *src, const int *perm, float *dst, int rows, int d) {
    if (rows <= 0) return 0;
    const int n = rows * (d / 4);
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, st, src, perm, dst, rows, d / 4);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// conv(kernel = stride = patch, no bias) as a GEMM: patches[(n, py, px)][(c, ky, kx)] = pixels[n, c, py*p+ky, px*p+kx]
__global__ void im2col_patches_kernel(const float *__restrict__ pixels, float *__restrict__ patches, int n, int S,
                                      int patch) {
    const int g = S / patch, kq = patch / 4;           // float4 along kx
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t tot = (size_t)n * g * g * 3 * patch * kq;
    if (i >= tot) return;
    const int x4 = i % kq;
    size_t r = i / kq;
    const int ky = r % patch; r /= patch;
    const int c = r % 3; r /= 3;
    const int px = r % g; r /= g;
    const int py = r % g;
    const int img = r / g;
    const float4 v = *reinterpret_cast<const float4 *>(
        pixels + (((size_t)img * 3 + c) * S + (py * patch + ky)) * S + px * patch + x4 * 4);
    reinterpret_cast<float4 *>(patches)[i] = v;
}
int launch_im2col_patches(hipStream_t st, const float *pixels, float *patches, int n, int S, int patch) {
    CAPDEC_CHECK(S % patch == 0 && patch % 4 == 0, "im2col: bad patch geometry");
    const int g = S / patch;
    const size_t tot = (size_t)n * g * g * 3 * patch * (patch / 4);
    if (tot == 0) return 0;
    hipLaunchKernelGGL(im2col_patches_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, pixels, patches, n,
                       S, patch);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// seq[n, 0] = class_embedding + pos[0]; seq[n, 1 + p] = patch_out[n*(ntok-1) + p] + pos[1 + p]
__global__ void vision_assemble_kernel(const float *__restrict__ patch_out, const float *__restrict__ cls,
                                       const float *__restrict__ pos, float *__restrict__ seq, int n, int ntok,
                                       int nv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * ntok * nv) return;
    const int c = i % nv, t = (i / nv) % ntok, img = i / (nv * ntok);
    const float4 p = reinterpret_cast<const float4 *>(pos)[(size_t)t * nv + c];
    const float4 a = t == 0 ? reinterpret_cast<const float4 *>(cls)[c]
                            : reinterpret_cast<const float4 *>(patch_out)[((size_t)img * (ntok - 1) + t - 1) * nv + c];
    reinterpret_cast<float4 *>(seq)[i] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
}
int launch_vision_assemble(hipStream_t st, const float *patch_out, const float *cls, const float *pos, float *seq,
                           int n, int ntok, int d) {
    const int tot = n * ntok * (d / 4);
    if (tot <= 0) return 0;
    hipLaunchKernelGGL(vision_assemble_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, patch_out, cls, pos, seq, n,
                       ntok, d / 4);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// rows src_rows[i] (i < *count_dev) of a packed f16x2 operand -> rows i of `out`: one wavefront per row, a lane moves
// (k-step, quad) pieces of 8 bytes per plane (the swizzle of a piece depends on its row: x3_group_offset)
template <int PLANES>
__global__ __launch_bounds__(256) void gather_packed_rows_kernel(const char *__restrict__ packed, int nk,
                                                                 const int *__restrict__ src_rows,
                                                                 const int *__restrict__ count_dev, char *__restrict__ out) {
    constexpr int BLOCK_B = PLANES * X3_PLANE_B;
    const int lane = threadIdx.x & 63;
    const int n = *count_dev;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += gridDim.x * 4) {
        const int r = src_rows[i];
        for (int q = lane; q < nk * 4; q += 64) {
            const int ks = q >> 2, quad = q & 3;
            const char *sp = packed + ((size_t)(r >> 7) * nk + ks) * BLOCK_B + x3_group_offset(r & 127, quad);
            char *dp = out + ((size_t)(i >> 7) * nk + ks) * BLOCK_B + x3_group_offset(i & 127, quad);
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl)
                *reinterpret_cast<uint2 *>(dp + pl * X3_PLANE_B) = *reinterpret_cast<const uint2 *>(sp + pl * X3_PLANE_B);
        }
    }
}
int launch_gather_packed_rows(hipStream_t st, const void *packed, int K, const int *src_rows, const int *count_dev,
                              int rows_cap, void *out, int fmt) {
    CAPDEC_CHECK(K % 16 == 0 && rows_cap > 0, "gather_packed_rows: bad sizes");
    CAPDEC_CHECK(fmt == PK_F16X2 || fmt == PK_F16X1 || fmt == PK_BF16X1, "gather_packed_rows: f16x2 or one-plane operands");
    const dim3 grid(std::min((rows_cap + 3) / 4, 1024));
    if (fmt == PK_F16X2)
        hipLaunchKernelGGL(gather_packed_rows_kernel<2>, grid, dim3(256), 0, st, (const char *)packed, K / X3_BK, src_rows,
                           count_dev, (char *)out);
    else
        hipLaunchKernelGGL(gather_packed_rows_kernel<1>, grid, dim3(256), 0, st, (const char *)packed, K / X3_BK, src_rows,
                           count_dev, (char *)out);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

CAPDEC_SAT_ACCESSOR(sat_count_elementwise)

}  // namespace capdec
