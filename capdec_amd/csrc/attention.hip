// Attention kernels (HBM-bound: one query row against a short cached context).
//
// GPT-2 decode / prefill: head_dim 64.  One wavefront per (row, head).  The wavefront is four
// 16-lane groups; a group owns one key position per iteration and its 16 lanes each hold a
// float4 of the 64-wide head (so a key is one 256-byte coalesced read), dot products are
// reduced with 4 xor-shuffles, the softmax over <= 256 positions with 64-lane shuffles.
// The KV cache is [layer][phys_row][head][ctx][64]: consecutive positions of a head are
// contiguous.  Beam search never copies K/V: row r reads position p from physical row
// caption*beam + anc[r][p] (ancestor table maintained by the beam-step kernel).
#include "bf16x3.h"
#include "common.h"

namespace capdec {

constexpr int ATT_CTX_MAX = 256;

__device__ __forceinline__ float dot4(const float4 &a, const float4 &b) {
    return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
}
__device__ __forceinline__ float group16_sum(float v) { return row16_sum(v); }
// lane l (0..15 of every row) <- sum over the four rows of lane l: row_bcast-free, two ds_swizzle-free steps
__device__ __forceinline__ float groups4_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// DECODE = true : rows = captions*beam, every row at context length L; own k/v (position L-1)
//                 taken from qkv and appended to the cache at phys row r.
// DECODE = false: prefill rows (caption, i), L = i + 1, everything read from the cache at phys
//                 row caption*beam (written by kv_scatter_prefill beforehand).
template <bool DECODE>
__global__ __launch_bounds__(256) void attn_gpt2_kernel(const float *__restrict__ qkv, float *__restrict__ kc,
                                                        float *__restrict__ vc, int total, int heads, int ctx,
                                                        int d, int beam, int Lparam, int P, int causal,
                                                        const uint8_t *__restrict__ anc, int anc_stride,
                                                        float *__restrict__ out, char *__restrict__ packed_out,
                                                        const int *__restrict__ cmap, int fmt) {
    __shared__ float sc[4][ATT_CTX_MAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, sub = lane & 15;
    const int gw = blockIdx.x * 4 + wave;
    const bool active = gw < total;
    const int row = active ? gw / heads : 0;
    const int head = active ? gw - row * heads : 0;
    // state row: after finished captions were compacted away, activation row r belongs to caption cmap[r / beam]
    // (KV cache, ancestor table and beam state keep the ORIGINAL caption indexing)
    const int srow = (DECODE && cmap) ? cmap[row / beam] * beam + row % beam : row;
    int L, phys_self;
    if (DECODE) {
        L = Lparam;
        phys_self = srow;
    } else {
        const int cap = row / P, i = row - cap * P;
        L = causal ? i + 1 : P;                  // CLIP's vision tower attends to the whole sequence
        phys_self = cap * beam;
    }
    const int cap_base = DECODE ? (srow / beam) * beam : phys_self;
    const size_t hstride = (size_t)ctx * 64;
    const float *qrow = qkv + (size_t)row * 3 * d;
    float4 q = reinterpret_cast<const float4 *>(qrow + head * 64)[sub];
    q.x *= 0.125f; q.y *= 0.125f; q.z *= 0.125f; q.w *= 0.125f;     // 1/sqrt(64), exact
    float4 kcur, vcur;
    const int Lpast = DECODE ? L - 1 : L;
    if (DECODE) {
        kcur = reinterpret_cast<const float4 *>(qrow + d + head * 64)[sub];
        vcur = reinterpret_cast<const float4 *>(qrow + 2 * d + head * 64)[sub];
        if (active && grp == 0) {
            const size_t o = ((size_t)phys_self * heads + head) * hstride + (size_t)(L - 1) * 64;
            reinterpret_cast<float4 *>(kc + o)[sub] = kcur;
            reinterpret_cast<float4 *>(vc + o)[sub] = vcur;
        }
    }
    // ---- scores
    for (int p0 = 0; p0 < Lpast; p0 += 4) {
        const int p = p0 + grp;
        if (p < Lpast) {
            int phys = phys_self;
            if (DECODE && anc) phys = cap_base + anc[(size_t)srow * anc_stride + p];
            const float4 k = reinterpret_cast<const float4 *>(kc + ((size_t)phys * heads + head) * hstride + (size_t)p * 64)[sub];
            const float s = group16_sum(dot4(q, k));
            if (sub == 0) sc[wave][p] = s;
        }
    }
    if (DECODE) {
        const float s = group16_sum(dot4(q, kcur));
        if (lane == 0) sc[wave][L - 1] = s;
    }
    __syncthreads();
    // ---- softmax statistics over L positions
    float mx = -INFINITY;
    for (int p = lane; p < L; p += 64) mx = fmaxf(mx, sc[wave][p]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int p = lane; p < L; p += 64) {
        const float e = expf(sc[wave][p] - mx);
        sc[wave][p] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    __syncthreads();
    // ---- P.V
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p0 = 0; p0 < Lpast; p0 += 4) {
        const int p = p0 + grp;
        if (p < Lpast) {
            int phys = phys_self;
            if (DECODE && anc) phys = cap_base + anc[(size_t)srow * anc_stride + p];
            const float4 v = reinterpret_cast<const float4 *>(vc + ((size_t)phys * heads + head) * hstride + (size_t)p * 64)[sub];
            const float w = sc[wave][p];
            acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
        }
    }
    if (DECODE && grp == 0) {
        const float w = sc[wave][L - 1];
        acc.x += w * vcur.x; acc.y += w * vcur.y; acc.z += w * vcur.z; acc.w += w * vcur.w;
    }
    acc.x = groups4_sum(acc.x); acc.y = groups4_sum(acc.y); acc.z = groups4_sum(acc.z); acc.w = groups4_sum(acc.w);
    if (active && grp == 0) {
        const float inv = 1.0f / sum;
        const float4 o = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
        if (packed_out) x3_store_quad(packed_out, d >> 4, row, head * 4 + (sub >> 2), sub & 3, o, fmt);   // A operand of c_proj
        else reinterpret_cast<float4 *>(out + (size_t)row * d + head * 64)[sub] = o;
    }
}

// Beam-shared decode attention: one wavefront per (caption, head) serves all BEAM rows of the caption.
// The beams of a caption mostly share their ancestors (the whole prefix, and usually all but the last
// few generated positions), so K/V of position p are loaded once per DISTINCT physical slot among
// consecutive beams instead of once per row: the 5 q vectors / 5 accumulators live in registers, a
// loaded key or value is reused while anc[b][p] does not change from beam b-1 to beam b.
template <int BEAM>
__global__ __launch_bounds__(256, 4) void attn_decode_beams_kernel(const float *__restrict__ qkv, float *__restrict__ kc,
                                                                float *__restrict__ vc, int total, int heads,
                                                                int ctx, int d, int L,
                                                                const uint8_t *__restrict__ anc, int anc_stride,
                                                                float *__restrict__ out,
                                                                char *__restrict__ packed_out,
                                                                const int *__restrict__ cmap, int fmt) {
    extern __shared__ __attribute__((aligned(16))) float sc_all[];      // [4 waves][BEAM][L] scores + [4][BEAM][L] slots
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, sub = lane & 15;
    const int gw = blockIdx.x * 4 + wave;
    const bool active = gw < total;
    const int cap = active ? gw / heads : 0;
    const int head = active ? gw - cap * heads : 0;
    float *sc = sc_all + (size_t)wave * BEAM * L;
    int *sl = reinterpret_cast<int *>(sc_all + (size_t)4 * BEAM * L) + (size_t)wave * BEAM * L;
    const size_t hstride = (size_t)ctx * 64;
    const int row0 = cap * BEAM;                                   // activation rows (compact)
    const int srow0 = (cmap ? cmap[cap] : cap) * BEAM;             // state rows: KV cache / ancestor table (original)
    const int Lpast = L - 1;

    // ancestor slots of this caption -> LDS (removes the dependent byte load in front of every K/V load)
    for (int i = lane; i < BEAM * Lpast; i += 64) {
        const int b = i / Lpast, p = i - b * Lpast;
        sl[b * L + p] = anc[(size_t)(srow0 + b) * anc_stride + p];
    }

    float4 q[BEAM];
#pragma unroll
    for (int b = 0; b < BEAM; ++b) {
        const float *qrow = qkv + (size_t)(row0 + b) * 3 * d;
        q[b] = reinterpret_cast<const float4 *>(qrow + head * 64)[sub];
        q[b].x *= 0.125f; q[b].y *= 0.125f; q[b].z *= 0.125f; q[b].w *= 0.125f;
        // the row's own key / value: score from registers, appended to the cache at its own slot
        const float4 kcur = reinterpret_cast<const float4 *>(qrow + d + head * 64)[sub];
        const float4 vcur = reinterpret_cast<const float4 *>(qrow + 2 * d + head * 64)[sub];
        if (active && grp == 0) {
            const size_t o = ((size_t)(srow0 + b) * heads + head) * hstride + (size_t)Lpast * 64;
            reinterpret_cast<float4 *>(kc + o)[sub] = kcur;
            reinterpret_cast<float4 *>(vc + o)[sub] = vcur;
        }
        const float s = group16_sum(dot4(q[b], kcur));
        if (lane == 0) sc[b * L + Lpast] = s;
    }
    __syncthreads();
    const float *kbase = kc + ((size_t)srow0 * heads + head) * hstride + sub * 4;
    const float *vbase = vc + ((size_t)srow0 * heads + head) * hstride + sub * 4;
    const size_t slot_stride = (size_t)heads * hstride;
    // The beams of a caption are paths of one tree: if they all sit on the same node at position p they share every
    // earlier position too, so "all beams read the same slot" holds exactly on a PREFIX [0, nconv) of the history
    // (the CLIP prefix and the converged part of the generated text).  Phase A streams that prefix without any
    // per-beam slot logic, four positions per 16-lane group per iteration (4 KB per wavefront in flight); phase B
    // handles the diverged tail.
    int nconv = Lpast;
    for (int base = 0; base < Lpast; base += 64) {
        const int p = base + lane;
        bool differs = false;
        if (p < Lpast) {
#pragma unroll
            for (int b = 1; b < BEAM; ++b) differs = differs || sl[b * L + p] != sl[p];
        }
        const unsigned long long m = __ballot(differs);
        if (m) { nconv = base + __ffsll((long long)m) - 1; break; }
    }
    const float *dummy = qkv + (size_t)row0 * 3 * d + head * 64 + sub * 4;   // an L1-hot line for masked-off loads
    // ---- scores, phase A
    for (int p0 = 0; p0 < nconv; p0 += 16) {
        int pp[4];
        float4 kk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pp[j] = p0 + 4 * j + grp;
            const bool v = pp[j] < nconv;
            const int s0 = v ? sl[pp[j]] : 0;
            kk[j] = *reinterpret_cast<const float4 *>(v ? kbase + s0 * slot_stride + (size_t)pp[j] * 64 : dummy);
        }
#pragma unroll
        for (int b = 0; b < BEAM; ++b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float sv = group16_sum(dot4(q[b], kk[j]));
                if (sub == 0 && pp[j] < nconv) sc[b * L + pp[j]] = sv;
            }
        }
    }
    // ---- scores, phase B: two positions per group per iteration.  All 2 x BEAM loads are issued back to back (no
    // branch, no wait between them): a beam whose slot equals the previous beam's re-reads the L1-hot dummy line and
    // takes the previous beam's registers, so HBM traffic stays one load per DISTINCT consecutive slot while the
    // wavefront keeps 2 x BEAM requests in flight.
    for (int p0 = nconv; p0 < Lpast; p0 += 8) {
        const int pa = p0 + grp, pb = p0 + 4 + grp;
        const bool va = pa < Lpast, vb = pb < Lpast;
        int sa[BEAM], sb[BEAM];
        float4 ka[BEAM], kb[BEAM];
#pragma unroll
        for (int b = 0; b < BEAM; ++b) {
            sa[b] = va ? sl[b * L + pa] : 0;
            sb[b] = vb ? sl[b * L + pb] : 0;
        }
#pragma unroll
        for (int b = 0; b < BEAM; ++b) {
            const bool na = va && (b == 0 || sa[b] != sa[b - 1]), nb = vb && (b == 0 || sb[b] != sb[b - 1]);
            ka[b] = *reinterpret_cast<const float4 *>(na ? kbase + sa[b] * slot_stride + (size_t)pa * 64 : dummy);
            kb[b] = *reinterpret_cast<const float4 *>(nb ? kbase + sb[b] * slot_stride + (size_t)pb * 64 : dummy);
        }
#pragma unroll
        for (int b = 0; b < BEAM; ++b) {
            if (b > 0 && sa[b] == sa[b - 1]) ka[b] = ka[b - 1];
            if (b > 0 && sb[b] == sb[b - 1]) kb[b] = kb[b - 1];
            const float s0 = group16_sum(dot4(q[b], ka[b]));
            const float s1 = group16_sum(dot4(q[b], kb[b]));
            if (sub == 0) {
                if (va) sc[b * L + pa] = s0;
                if (vb) sc[b * L + pb] = s1;
            }
        }
    }
    __syncthreads();
    // ---- softmax statistics per beam
    float inv[BEAM];
#pragma unroll
    for (int b = 0; b < BEAM; ++b) {
        float mx = -INFINITY;
        for (int p = lane; p < L; p += 64) mx = fmaxf(mx, sc[b * L + p]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int p = lane; p < L; p += 64) {
            const float e = expf(sc[b * L + p] - mx);
            sc[b * L + p] = e;
            sum += e;
        }
        inv[b] = 1.0f / wave_sum(sum);
    }
    __syncthreads();
    // ---- P.V
    float4 acc[BEAM];
#pragma unroll
    for (int b = 0; b < BEAM; ++b) acc[b] = make_float4(0.f, 0.f, 0.f, 0.f);
    // phase A: converged prefix, four positions per group per iteration
    for (int p0 = 0; p0 < nconv; p0 += 16) {
        int pp[4];
        float4 xx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pp[j] = p0 + 4 * j + grp;
            const bool v = pp[j] < nconv;
            const int s0 = v ? sl[pp[j]] : 0;
            xx[j] = *reinterpret_cast<const float4 *>(v ? vbase + s0 * slot_stride + (size_t)pp[j] * 64 : dummy);
        }
#pragma unroll
        for (int b = 0; b < BEAM; ++b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float w = pp[j] < nconv ? sc[b * L + pp[j]] : 0.f;
                acc[b].x += w * xx[j].x; acc[b].y += w * xx[j].y; acc[b].z += w * xx[j].z; acc[b].w += w * xx[j].w;
            }
        }
    }
    // phase B: diverged tail
    for (int p0 = nconv; p0 < Lpast; p0 += 8) {
        const int pa = p0 + grp, pb = p0 + 4 + grp;
        const bool va = pa < Lpast, vb = pb < Lpast;
        int sa[BEAM], sb[BEAM];
        float4 xa[BEAM], xb[BEAM];
#pragma unroll
        for (int b = 0; b < BEAM; ++b) {
            sa[b] = va ? sl[b * L + pa] : 0;
            sb[b] = vb ? sl[b * L + pb] : 0;
        }
#pragma unroll
        for (int b = 0; b < BEAM; ++b) {
            const bool na = va && (b == 0 || sa[b] != sa[b - 1]), nb = vb && (b == 0 || sb[b] != sb[b - 1]);
            xa[b] = *reinterpret_cast<const float4 *>(na ? vbase + sa[b] * slot_stride + (size_t)pa * 64 : dummy);
            xb[b] = *reinterpret_cast<const float4 *>(nb ? vbase + sb[b] * slot_stride + (size_t)pb * 64 : dummy);
        }
#pragma unroll
        for (int b = 0; b < BEAM; ++b) {
            if (b > 0 && sa[b] == sa[b - 1]) xa[b] = xa[b - 1];
            if (b > 0 && sb[b] == sb[b - 1]) xb[b] = xb[b - 1];
            const float wa = va ? sc[b * L + pa] : 0.f, wb = vb ? sc[b * L + pb] : 0.f;
            acc[b].x += wa * xa[b].x + wb * xb[b].x; acc[b].y += wa * xa[b].y + wb * xb[b].y;
            acc[b].z += wa * xa[b].z + wb * xb[b].z; acc[b].w += wa * xa[b].w + wb * xb[b].w;
        }
    }
#pragma unroll
    for (int b = 0; b < BEAM; ++b) {
        if (grp == 0) {   // own token
            const float4 vcur = reinterpret_cast<const float4 *>(qkv + (size_t)(row0 + b) * 3 * d + 2 * d + head * 64)[sub];
            const float w = sc[b * L + Lpast];
            acc[b].x += w * vcur.x; acc[b].y += w * vcur.y; acc[b].z += w * vcur.z; acc[b].w += w * vcur.w;
        }
        acc[b].x = groups4_sum(acc[b].x); acc[b].y = groups4_sum(acc[b].y);
        acc[b].z = groups4_sum(acc[b].z); acc[b].w = groups4_sum(acc[b].w);
        if (active && grp == 0) {
            const float4 o = make_float4(acc[b].x * inv[b], acc[b].y * inv[b], acc[b].z * inv[b], acc[b].w * inv[b]);
            if (packed_out) x3_store_quad(packed_out, d >> 4, row0 + b, head * 4 + (sub >> 2), sub & 3, o, fmt);
            else reinterpret_cast<float4 *>(out + (size_t)(row0 + b) * d + head * 64)[sub] = o;
        }
    }
}


// Prefill / CLIP-tower attention: one wavefront per (caption, head, block of R query rows).  The R rows share every
// K / V row they read (taken straight from the fused qkv activations -- no cache round trip), so a key is loaded once
// per R queries instead of once per query: the per-row kernel above moved 61 GB per launch through L2 on the 77-token
// text tower (20 TB/s).  Same lane mapping and summation order as the per-row kernel: 16-lane group g owns the
// positions p = g (mod 4), four positions per group per iteration; scores through LDS; causal rows mask p > i.
template <int R>
__global__ __launch_bounds__(256) void attn_prefill_rows_kernel(const float *__restrict__ qkv, int total, int heads,
                                                                int P, int d, int causal, float *__restrict__ out,
                                                                char *__restrict__ packed_out, int fmt) {
    extern __shared__ __attribute__((aligned(16))) float sc_rows[];       // [4 waves][R][P]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, sub = lane & 15;
    const int gw = blockIdx.x * 4 + wave;
    const bool active = gw < total;
    const int nblk = (P + R - 1) / R;
    const int rb = active ? gw % nblk : 0;
    const int ch = active ? gw / nblk : 0;
    const int head = ch % heads, cap = ch / heads;
    const int i0 = rb * R, nr = min(R, P - i0);
    const int Lk = causal ? min(P, i0 + R) : P;                            // keys any of the R rows can see
    float *sc = sc_rows + (size_t)wave * R * P;
    const float *base = qkv + (size_t)cap * P * 3 * d + head * 64 + sub * 4;   // + row * 3d: q; + d: k; + 2d: v
    float4 q[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        q[r] = *reinterpret_cast<const float4 *>(base + (size_t)min(i0 + r, P - 1) * 3 * d);
        q[r].x *= 0.125f; q[r].y *= 0.125f; q[r].z *= 0.125f; q[r].w *= 0.125f;     // 1/sqrt(64), exact
    }
    for (int p0 = 0; p0 < Lk; p0 += 16) {
        float4 kk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            kk[j] = *reinterpret_cast<const float4 *>(base + (size_t)min(p0 + 4 * j + grp, P - 1) * 3 * d + d);
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pj = p0 + 4 * j + grp;
                const float sv = group16_sum(dot4(q[r], kk[j]));
                if (sub == 0 && pj < Lk) sc[r * P + pj] = (causal && pj > i0 + r) ? -INFINITY : sv;
            }
        }
    }
    __syncthreads();
    float inv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float mx = -INFINITY;
        for (int p = lane; p < Lk; p += 64) mx = fmaxf(mx, sc[r * P + p]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int p = lane; p < Lk; p += 64) {
            const float e = expf(sc[r * P + p] - mx);
            sc[r * P + p] = e;
            sum += e;
        }
        inv[r] = 1.0f / wave_sum(sum);
    }
    __syncthreads();
    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p0 = 0; p0 < Lk; p0 += 16) {
        float4 xx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            xx[j] = *reinterpret_cast<const float4 *>(base + (size_t)min(p0 + 4 * j + grp, P - 1) * 3 * d + 2 * d);
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pj = p0 + 4 * j + grp;
                const float w = pj < Lk ? sc[r * P + pj] : 0.f;
                acc[r].x += w * xx[j].x; acc[r].y += w * xx[j].y; acc[r].z += w * xx[j].z; acc[r].w += w * xx[j].w;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        acc[r].x = groups4_sum(acc[r].x); acc[r].y = groups4_sum(acc[r].y);
        acc[r].z = groups4_sum(acc[r].z); acc[r].w = groups4_sum(acc[r].w);
        if (active && grp == 0 && r < nr) {
            const int row = cap * P + i0 + r;
            const float4 o = make_float4(acc[r].x * inv[r], acc[r].y * inv[r], acc[r].z * inv[r], acc[r].w * inv[r]);
            if (packed_out) x3_store_quad(packed_out, d >> 4, row, head * 4 + (sub >> 2), sub & 3, o, fmt);
            else reinterpret_cast<float4 *>(out + (size_t)row * d + head * 64)[sub] = o;
        }
    }
}

// K/V of prefill row (cap, i) -> cache[phys = cap*beam][head][i][:]
__global__ void kv_scatter_prefill_kernel(const float *__restrict__ qkv, float *__restrict__ kc,
                                          float *__restrict__ vc, int ncap, int P, int beam, int heads, int ctx,
                                          int d) {
    const int nv = d / 4;                       // float4 per row per tensor
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncap * P * nv) return;
    const int c4 = i % nv, row = i / nv;
    const int cap = row / P, pos = row - cap * P;
    const int head = (c4 * 4) / 64, within = (c4 * 4) % 64;
    const size_t o = (((size_t)cap * beam * heads + head) * ctx + pos) * 64 + within;
    const float *r = qkv + (size_t)row * 3 * d;
    *reinterpret_cast<float4 *>(kc + o) = reinterpret_cast<const float4 *>(r + d)[c4];
    *reinterpret_cast<float4 *>(vc + o) = reinterpret_cast<const float4 *>(r + 2 * d)[c4];
}

int launch_kv_scatter_prefill(hipStream_t st, const float *qkv, const KvCache &c, int layer, int ncap, int P,
                              int beam) {
    const int d = c.heads * c.hd;
    const int tot = ncap * P * (d / 4);
    if (tot <= 0) return 0;
    hipLaunchKernelGGL(kv_scatter_prefill_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, qkv,
                       c.k + layer * c.layer_stride(), c.v + layer * c.layer_stride(), ncap, P, beam, c.heads, c.ctx,
                       d);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

int launch_attn_prefill(hipStream_t st, const float *qkv, const KvCache &c, int layer, int ncap, int P, int beam,
                        float *out, bool causal, void *packed_out, int fmt) {
    CAPDEC_CHECK(c.hd == 64, "attention: head_dim must be 64");
    CAPDEC_CHECK(P <= ATT_CTX_MAX && P <= c.ctx, "attention: prefix longer than the supported context");
    (void)layer; (void)beam;                      // K / V come straight from qkv (the cache is filled by kv_scatter_prefill)
    constexpr int R = 8;
    const int total = ncap * c.heads * ((P + R - 1) / R);
    if (total <= 0) return 0;
    const size_t lds = (size_t)4 * R * P * sizeof(float);
    hipLaunchKernelGGL(attn_prefill_rows_kernel<R>, dim3((total + 3) / 4), dim3(256), lds, st, qkv, total, c.heads, P,
                       c.heads * c.hd, causal ? 1 : 0, out, (char *)packed_out, fmt);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

int launch_attn_decode(hipStream_t st, const float *qkv, const KvCache &c, int layer, int rows, int beam, int L,
                       const uint8_t *anc, int anc_stride, float *out, void *packed_out, const int *cmap, int fmt) {
    CAPDEC_CHECK(c.hd == 64, "attention: head_dim must be 64");
    CAPDEC_CHECK(L >= 1 && L <= ATT_CTX_MAX && L <= c.ctx, "attention: context length out of range");
    if (anc != nullptr && beam > 1) {
        const int ncap = rows / beam, total = ncap * c.heads;
        if (total <= 0) return 0;
        const size_t lds = (size_t)2 * 4 * beam * L * sizeof(float);   // scores + ancestor slots
        dim3 grid((total + 3) / 4), block(256);
        float *kl = c.k + layer * c.layer_stride(), *vl = c.v + layer * c.layer_stride();
#define LAUNCH_BEAMS(B)                                                                                         \
    hipLaunchKernelGGL(attn_decode_beams_kernel<B>, grid, block, lds, st, qkv, kl, vl, total, c.heads, c.ctx,      \
                       c.heads * c.hd, L, anc, anc_stride, out, (char *)packed_out, cmap, fmt)
        switch (beam) {
            case 2: LAUNCH_BEAMS(2); break;
            case 3: LAUNCH_BEAMS(3); break;
            case 4: LAUNCH_BEAMS(4); break;
            case 5: LAUNCH_BEAMS(5); break;
            case 6: LAUNCH_BEAMS(6); break;
            case 7: LAUNCH_BEAMS(7); break;
            case 8: LAUNCH_BEAMS(8); break;
            default: CAPDEC_CHECK(false, "attention: beam must be in 1..8");
        }
#undef LAUNCH_BEAMS
        CAPDEC_HIP(hipGetLastError());
        return 0;
    }
    const int total = rows * c.heads;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(attn_gpt2_kernel<true>, dim3((total + 3) / 4), dim3(256), 0, st, qkv,
                       c.k + layer * c.layer_stride(), c.v + layer * c.layer_stride(), total, c.heads, c.ctx,
                       c.heads * c.hd, beam, L, 0, 1, anc, anc_stride, out, (char *)packed_out, cmap, fmt);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// TransformerMapper self-attention (reference transformer_mapper.py:22-51): bidirectional, seq =
// clip_len + P (20 at the headline config), 8 heads x 96.  One block per (caption, head): K and V
// of the head are staged in LDS ([seq][hd+1], conflict-free for both access directions), each
// wavefront then serves query rows: lanes over keys for q.k, lanes over channels for p.v.
__global__ __launch_bounds__(256) void attn_mapper_kernel(const float *__restrict__ q, int ldq,
                                                          const float *__restrict__ k, const float *__restrict__ v,
                                                          int ldkv, float *__restrict__ out, int seq, int heads,
                                                          int hd, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ld = hd + 1;
    float *Ks = sm;                       // [seq][ld]
    float *Vs = Ks + seq * ld;            // [seq][ld]
    float *qb = Vs + seq * ld;            // [4][hd]
    float *pb = qb + 4 * hd;              // [4][seq]
    const int cap = blockIdx.x / heads, head = blockIdx.x - cap * heads;
    const int d = heads * hd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < seq * hd; i += 256) {
        const int j = i / hd, c = i - j * hd;
        const size_t o = ((size_t)cap * seq + j) * ldkv + head * hd + c;
        Ks[j * ld + c] = k[o];
        Vs[j * ld + c] = v[o];
    }
    __syncthreads();
    for (int i = wave; i < seq; i += 4) {
        const float *qr = q + ((size_t)cap * seq + i) * ldq + head * hd;
        for (int c = lane; c < hd; c += 64) qb[wave * hd + c] = qr[c];
        __builtin_amdgcn_wave_barrier();
        float mx = -INFINITY;
        for (int j = lane; j < seq; j += 64) {
            float s = 0.f;
            for (int c = 0; c < hd; ++c) s += qb[wave * hd + c] * Ks[j * ld + c];
            s *= scale;
            pb[wave * seq + j] = s;
            mx = fmaxf(mx, s);
        }
        mx = wave_max(mx);
        float sum = 0.f;
        for (int j = lane; j < seq; j += 64) {
            const float e = expf(pb[wave * seq + j] - mx);
            pb[wave * seq + j] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        __builtin_amdgcn_wave_barrier();
        const float inv = 1.0f / sum;
        for (int c = lane; c < hd; c += 64) {
            float a = 0.f;
            for (int j = 0; j < seq; ++j) a += pb[wave * seq + j] * Vs[j * ld + c];
            out[((size_t)cap * seq + i) * d + head * hd + c] = a * inv;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

int launch_attn_mapper(hipStream_t st, const float *q, int ldq, const float *k, const float *v, int ldkv, float *out,
                       int n, int seq, int heads, int hd) {
    if (n <= 0) return 0;
    const size_t lds = ((size_t)2 * seq * (hd + 1) + 4 * hd + 4 * seq) * sizeof(float);
    CAPDEC_CHECK(lds <= 160 * 1024, "mapper attention: sequence too long for LDS");
    if (lds > 64 * 1024)
        CAPDEC_HIP(hipFuncSetAttribute((const void *)attn_mapper_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds));
    hipLaunchKernelGGL(attn_mapper_kernel, dim3(n * heads), dim3(256), lds, st, q, ldq, k, v, ldkv, out, seq, heads, hd,
                       (float)pow((double)hd, -0.5));   // python: head_dim ** -0.5
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

}  // namespace capdec
