// Attention kernels (HBM-bound: one query row against a short cached context).
//
// GPT-2 decode / prefill: head_dim 64.  One wavefront per (row, head).  The wavefront is four
// 16-lane groups; a group owns one key position per iteration and its 16 lanes each hold a
// float4 of the 64-wide head (so a key is one 256-byte coalesced read), dot products are
// reduced with 4 xor-shuffles, the softmax over <= 1024 positions with 64-lane shuffles.
// The KV cache is [layer][phys_row][head][ctx][64]: consecutive positions of a head are
// contiguous.  Beam search never copies K/V: row r reads position p from physical row
// caption*beam + anc[r][p] (ancestor table maintained by the beam-step kernel).
#include <cstdlib>

#include "bf16x3.h"
#include "common.h"

namespace capdec {

constexpr int ATT_CTX_MAX = 1024;    // GPT-2's n_positions (contexts beyond ~400 positions need more than 64 KB of LDS per block for
                                     // the slot table / the score rows: the launchers raise the kernel's limit per launch, fewer blocks share a CU)
#define ATT_BIG_LDS(K, bytes)                                                                                   \
    if ((bytes) > 64 * 1024)                                                                                    \
    CAPDEC_HIP(hipFuncSetAttribute((const void *)(K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)))
constexpr int ATT_MFMA_MIN_P = 24;   // prefill sequences of 24 .. 128 positions take the matrix-core kernel (shorter ones -- the
constexpr int ATT_MFMA_MAX_P = 128;  // 10-token caption prefix -- would leave most of a 32 x 32 tile empty; longer ones keep the
                                     // per-row kernel: the score registers of a lane are sized for four key tiles)

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

// ---- KV cache element type: fp32 (parity modes) or bf16 (CAPDEC_GEMM_MODE=bf16, BASELINE configs[1]: K / V are rounded
// to bf16 (RNE) when they are produced -- the value that is cached is also the value the producing step attends to)
template <typename KV> struct KvIo;
template <> struct KvIo<float> {
    static __device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
    static __device__ __forceinline__ void st4(float *p, const float4 v) { *reinterpret_cast<float4 *>(p) = v; }
    static __device__ __forceinline__ float4 round4(const float4 v) { return v; }
};
template <> struct KvIo<__bf16> {
    static __device__ __forceinline__ float4 ld4(const __bf16 *p) {
        const bf16x4 t = *reinterpret_cast<const bf16x4 *>(p);
        return make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
    }
    static __device__ __forceinline__ void st4(__bf16 *p, const float4 v) {
        bf16x4 t;
        t[0] = (__bf16)v.x; t[1] = (__bf16)v.y; t[2] = (__bf16)v.z; t[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4 *>(p) = t;
    }
    static __device__ __forceinline__ float4 round4(const float4 v) {
        return make_float4((float)(__bf16)v.x, (float)(__bf16)v.y, (float)(__bf16)v.z, (float)(__bf16)v.w);
    }
};

constexpr float ATT_QSCALE = 0.125f * 1.4426950408889634f;          // 1/sqrt(head_dim) * log2(e)
constexpr float ATT_NEG = -1.0e30f;                                 // "no score yet": finite, so max - max never is inf - inf
__device__ __forceinline__ float att_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32; 2^-inf = 0

// Beam-shared decode attention: one wavefront per (caption, head) serves all BEAM rows of the caption.
// The beams of a caption mostly share their ancestors (the whole prefix, and usually all but the last
// few generated positions), so K/V of position p are loaded once per DISTINCT physical slot among
// consecutive beams instead of once per row: the BEAM q vectors / accumulators live in registers.
//
// Single pass, online softmax (round 2): the wavefront is four 16-lane groups; group g owns the positions
// p = g (mod 4) and keeps its OWN running (max, sum, weighted-V accumulator) per beam, so the K and the V of a
// chunk of positions are loaded TOGETHER (twice the bytes in flight per wavefront, half the dependent memory round
// trips of the former scores -> LDS -> softmax -> P.V structure), no score ever goes through LDS and there is no
// block barrier in the loop; the four partial softmaxes are merged once at the end (max / rescale / sum across the
// groups).  LDS only holds the caption's ancestor-slot table.
// CUR (round 3): the rows' own K / V are ALREADY in the cache at position L-1 (the qkv GEMM's epilogue wrote them there,
// QkvScatter): the current token is then just the last cached position (slot = the beam itself) -- no own-token loads
// from the qkv activations, no append stores, no special first term of the running softmax.
// DMA (round 3, fp32 cache): the converged prefix is streamed through a per-wavefront LDS double buffer by LDS-DMA
// (global_load_lds: no destination registers), so the K / V of the NEXT 4 NA positions are in flight while the current
// ones are reduced -- the register-landed loop has one iteration's loads in flight only while it waits for them, and
// the converged case is latency-bound (5.4 TB/s; the diverged tail with its 2 BEAM loads per lane reaches 5.8-6.2).
// Every lane reads back exactly the 16 bytes it asked for (LDS image = lane order): the LDS is a landing buffer, not a
// sharing stage -- each key / value is still used by one wavefront only.
template <int BEAM, typename KV, int OCC, int NA = 2, bool CUR = false, bool DMA = false>
__global__ __launch_bounds__(256, OCC) void attn_decode_beams_kernel(const float *__restrict__ qkv, KV *__restrict__ kc,
                                                                KV *__restrict__ vc, int total, int heads,
                                                                int ctx, int d, int L,
                                                                const uint8_t *__restrict__ anc, int anc_stride,
                                                                float *__restrict__ out,
                                                                char *__restrict__ packed_out,
                                                                const int *__restrict__ cmap, int fmt, int npre,
                                                                int ring_off) {
    extern __shared__ __attribute__((aligned(16))) int sl_all[];      // [4 waves][BEAM][L] ancestor slots (+ DMA ring at ring_off)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, sub = lane & 15;
    const int gw = blockIdx.x * 4 + wave;
    const bool active = gw < total;
    const int cap = active ? gw / heads : 0;
    const int head = active ? gw - cap * heads : 0;
    int *sl = sl_all + (size_t)wave * BEAM * L;
    const size_t hstride = (size_t)ctx * 64;
    const int row0 = cap * BEAM;                                   // activation rows (compact)
    const int srow0 = (cmap ? cmap[cap] : cap) * BEAM;             // state rows: KV cache / ancestor table (original)
    const int Lpast = CUR ? L : L - 1;                              // cached positions this step attends to

    // The first `npre` positions (the CLIP prefix) live in slot 0 for every beam whatever the ancestor table says: their
    // K / V loads are issued FIRST, so they fly together with the table's and the q rows' loads instead of one memory
    // round trip behind them (a wavefront lives for ~25 us; the round trip is ~1-2 of them)
    const int npe = min(4 * NA, (min(npre, Lpast) >> 2) << 2);     // positions [0, npe) are peeled off phase A
    float4 kk0[NA], vv0[NA];
    {
        const KV *kb0 = kc + ((size_t)srow0 * heads + head) * hstride + sub * 4;
        const KV *vb0 = vc + ((size_t)srow0 * heads + head) * hstride + sub * 4;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int pj = 4 * j + grp;
            const bool v = active && pj < npe;
            kk0[j] = KvIo<KV>::ld4(v ? kb0 + pj * 64 : kb0);
            vv0[j] = KvIo<KV>::ld4(v ? vb0 + pj * 64 : kb0);
        }
    }
    // ancestor slots of this caption -> LDS (removes the dependent byte load in front of every K/V load)
    // (anc == nullptr: greedy decode -- one row per caption, everything in its own slot 0)
    for (int i = lane; i < BEAM * Lpast; i += 64) {
        const int b = i / Lpast, p = i - b * Lpast;
        sl[b * L + p] = (CUR && p == L - 1) ? b : (anc ? anc[(size_t)(srow0 + b) * anc_stride + p] : 0);
    }

    float4 q[BEAM], acc[BEAM];
    float mrun[BEAM], lrun[BEAM];
#pragma unroll
    for (int b = 0; b < BEAM; ++b) {
        const float *qrow = qkv + (size_t)(row0 + b) * 3 * d;
        q[b] = reinterpret_cast<const float4 *>(qrow + head * 64)[sub];
        // scores are kept in the log2 domain (q pre-scaled by 1/sqrt(64) * log2 e): the softmax weights are then ONE
        // v_exp_f32 each (2^x, ~1 ulp) instead of an expf call -- every lane of a 16-lane group evaluates its group's
        // weights, so the exponential is the dominant VALU cost of this kernel
        q[b].x *= ATT_QSCALE; q[b].y *= ATT_QSCALE; q[b].z *= ATT_QSCALE; q[b].w *= ATT_QSCALE;
        if constexpr (CUR) {
            mrun[b] = ATT_NEG;
            lrun[b] = 0.f;
            acc[b] = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        // the row's own key / value: appended to the cache at its own slot; group 0 starts its running softmax with it
        const float4 kcur = KvIo<KV>::round4(reinterpret_cast<const float4 *>(qrow + d + head * 64)[sub]);
        const float4 vcur = KvIo<KV>::round4(reinterpret_cast<const float4 *>(qrow + 2 * d + head * 64)[sub]);
        if (active && grp == 0) {
            const size_t o = ((size_t)(srow0 + b) * heads + head) * hstride + (size_t)Lpast * 64 + sub * 4;
            KvIo<KV>::st4(kc + o, kcur);
            KvIo<KV>::st4(vc + o, vcur);
        }
        const float s = group16_sum(dot4(q[b], kcur));
        mrun[b] = grp == 0 ? s : ATT_NEG;
        lrun[b] = grp == 0 ? 1.f : 0.f;
        acc[b] = grp == 0 ? vcur : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // slot table visible: the table (and the DMA ring) are PRIVATE to the wavefront and a wavefront's LDS operations
    // complete in order, so a wavefront-scope fence is all the ordering needed -- a block barrier would make every
    // wavefront wait for the slowest of four unrelated (caption, head) pairs' first loads
    if (ring_off < 0) {                                            // (CAPDEC_ATT_WSYNC=0: the round-2 block barrier, for A/B)
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const KV *kbase = kc + ((size_t)srow0 * heads + head) * hstride + sub * 4;
    const KV *vbase = vc + ((size_t)srow0 * heads + head) * hstride + sub * 4;
    const int slot_stride = heads * ctx * 64;      // offsets inside one caption's K / V region fit 32 bits (<= 8 x 16 x 256 x 64)
    // The beams of a caption are paths of one tree: if they all sit on the same node at position p they share every
    // earlier position too, so "all beams read the same slot" holds exactly on a PREFIX [0, nconv) of the history
    // (the CLIP prefix and the converged part of the generated text).  Phase A streams that prefix without any
    // per-beam slot logic; phase B handles the diverged tail.
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
    const KV *dummy = kbase;                                       // a valid address for masked-off loads (value unused)
    // online-softmax update of beam b with NP (score, value) pairs of this group; sv[j] = -inf marks "no position"
#define ATT_UPDATE(b, NP, sv, xv)                                                                        \
    {                                                                                                    \
        float mx_ = fmaxf(mrun[b], ATT_NEG);          /* branch-free: a group that has seen nothing keeps */ \
        _Pragma("unroll") for (int j_ = 0; j_ < NP; ++j_) mx_ = fmaxf(mx_, sv[j_]);   /* max = ATT_NEG, weights 2^-inf = 0 */ \
        const float sc_ = att_exp2(mrun[b] - mx_);                                                       \
        float ls_ = lrun[b] * sc_;                                                                       \
        float4 a_ = make_float4(acc[b].x * sc_, acc[b].y * sc_, acc[b].z * sc_, acc[b].w * sc_);         \
        _Pragma("unroll") for (int j_ = 0; j_ < NP; ++j_) {                                              \
            const float w_ = att_exp2(sv[j_] - mx_);      /* 2^-inf = 0 for masked positions */          \
            ls_ += w_;                                                                                   \
            a_.x += w_ * xv[j_].x; a_.y += w_ * xv[j_].y; a_.z += w_ * xv[j_].z; a_.w += w_ * xv[j_].w;  \
        }                                                                                                \
        mrun[b] = mx_; lrun[b] = ls_; acc[b] = a_;                                                       \
    }
    // ---- phase A: converged prefix, NA positions per group per iteration: NA K + NA V loads in flight per lane
    if (npe > 0) {                                                 // the peeled positions: loaded before the table arrived
#pragma unroll
        for (int b = 0; b < BEAM; ++b) {
            float sv[NA];
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const float t = group16_sum(dot4(q[b], kk0[j]));
                sv[j] = 4 * j + grp < npe ? t : -INFINITY;
            }
            ATT_UPDATE(b, NA, sv, vv0)
        }
    }
    if constexpr (DMA) {
        typedef __attribute__((address_space(3))) void lds_void_a;
        typedef const __attribute__((address_space(1))) void glb_void_a;
        constexpr int BUF = 2 * NA * 1024;                        // one iteration: NA K pieces + NA V pieces of 1 KB
        char *ring = reinterpret_cast<char *>(sl_all) + ring_off + __builtin_amdgcn_readfirstlane(wave) * (2 * BUF);
        const int n_it = (nconv - npe + 4 * NA - 1) / (4 * NA);
#define ATT_ISSUE(it_)                                                                                          \
        {                                                                                                       \
            char *dst = ring + ((it_) & 1) * BUF;                                                               \
            _Pragma("unroll") for (int j = 0; j < NA; ++j) {                                                    \
                const int pj = npe + (it_) * 4 * NA + 4 * j + grp;                                              \
                const bool v = pj < nconv;                                                                      \
                const int o = v ? sl[pj] * slot_stride + pj * 64 : 0;                                           \
                __builtin_amdgcn_global_load_lds((glb_void_a *)(v ? kbase + o : dummy), (lds_void_a *)(dst + j * 1024), 16, 0, 0);             \
                __builtin_amdgcn_global_load_lds((glb_void_a *)(v ? vbase + o : dummy), (lds_void_a *)(dst + (NA + j) * 1024), 16, 0, 0);      \
            }                                                                                                   \
        }
        if (n_it > 0) ATT_ISSUE(0)
        for (int it = 0; it < n_it; ++it) {
            if (it + 1 < n_it) {
                ATT_ISSUE(it + 1)
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * NA));    // vmcnt(2 NA): this iteration's pieces have landed
            } else {
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0)
            }
            asm volatile("" ::: "memory");
            // the landed pieces come back through inline-asm ds_reads: for a compiler-visible LDS load hipcc's waitcnt
            // pass cannot tell which LDS-DMA it may alias and waits vmcnt(0) -- the very drain the double buffer avoids
            static_assert(NA == 2, "the LDS-DMA variant reads two K and two V pieces per iteration");
            typedef float f4v __attribute__((ext_vector_type(4)));
            const unsigned laddr = (unsigned)(uintptr_t)(ring + (it & 1) * BUF + lane * 16);
            f4v k0, k1, v0, v1;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                         "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(k0), "=&v"(k1), "=&v"(v0), "=&v"(v1) : "v"(laddr) : "memory");
            int pp[NA];
            float4 kk[NA], vv[NA];
            kk[0] = make_float4(k0[0], k0[1], k0[2], k0[3]); kk[1] = make_float4(k1[0], k1[1], k1[2], k1[3]);
            vv[0] = make_float4(v0[0], v0[1], v0[2], v0[3]); vv[1] = make_float4(v1[0], v1[1], v1[2], v1[3]);
#pragma unroll
            for (int j = 0; j < NA; ++j) pp[j] = npe + it * 4 * NA + 4 * j + grp;
#pragma unroll
            for (int b = 0; b < BEAM; ++b) {
                float sv[NA];
#pragma unroll
                for (int j = 0; j < NA; ++j) {
                    const float t = group16_sum(dot4(q[b], kk[j]));
                    sv[j] = pp[j] < nconv ? t : -INFINITY;
                }
                ATT_UPDATE(b, NA, sv, vv)
            }
            asm volatile("" ::: "memory");                        // (the buffer is re-filled by the next ATT_ISSUE)
        }
#undef ATT_ISSUE
    } else
    for (int p0 = npe; p0 < nconv; p0 += 4 * NA) {
        int pp[NA];
        float4 kk[NA], vv[NA];
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            pp[j] = p0 + 4 * j + grp;
            const bool v = pp[j] < nconv;
            const int o = v ? sl[pp[j]] * slot_stride + pp[j] * 64 : 0;
            kk[j] = KvIo<KV>::ld4(v ? kbase + o : dummy);
            vv[j] = KvIo<KV>::ld4(v ? vbase + o : dummy);
        }
#pragma unroll
        for (int b = 0; b < BEAM; ++b) {
            float sv[NA];
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const float t = group16_sum(dot4(q[b], kk[j]));
                sv[j] = pp[j] < nconv ? t : -INFINITY;
            }
            ATT_UPDATE(b, NA, sv, vv)
        }
    }
    // ---- phase B: diverged tail, one position per group per iteration.  All 2 x BEAM loads are issued back to back
    // (no branch, no wait between them): a beam whose slot equals the previous beam's re-reads the hot dummy line and
    // takes the previous beam's registers, so HBM traffic stays one load per DISTINCT consecutive slot.
    for (int p0 = nconv; p0 < Lpast; p0 += 4) {
        const int pa = p0 + grp;
        const bool va = pa < Lpast;
        int sa[BEAM];
        float4 ka[BEAM], xa[BEAM];
#pragma unroll
        for (int b = 0; b < BEAM; ++b) sa[b] = va ? sl[b * L + pa] : 0;
#pragma unroll
        for (int b = 0; b < BEAM; ++b) {
            const bool na = va && (b == 0 || sa[b] != sa[b - 1]);
            const int o = sa[b] * slot_stride + pa * 64;
            ka[b] = KvIo<KV>::ld4(na ? kbase + o : dummy);
            xa[b] = KvIo<KV>::ld4(na ? vbase + o : dummy);
        }
#pragma unroll
        for (int b = 0; b < BEAM; ++b) {
            if (b > 0 && sa[b] == sa[b - 1]) { ka[b] = ka[b - 1]; xa[b] = xa[b - 1]; }
            float sv[1];
            const float t = group16_sum(dot4(q[b], ka[b]));
            sv[0] = va ? t : -INFINITY;
            const float4 *xv = &xa[b];
            ATT_UPDATE(b, 1, sv, xv)
        }
    }
#undef ATT_UPDATE
    // ---- merge the four groups' partial softmaxes and write the rows
#pragma unroll
    for (int b = 0; b < BEAM; ++b) {
        float M = mrun[b];
        M = fmaxf(M, __shfl_xor(M, 16, 64));
        M = fmaxf(M, __shfl_xor(M, 32, 64));                       // finite: group 0 holds the row's own token
        const float sc = att_exp2(mrun[b] - M);
        const float lt = groups4_sum(lrun[b] * sc);
        float4 a = make_float4(acc[b].x * sc, acc[b].y * sc, acc[b].z * sc, acc[b].w * sc);
        a.x = groups4_sum(a.x); a.y = groups4_sum(a.y); a.z = groups4_sum(a.z); a.w = groups4_sum(a.w);
        if (active && grp == 0) {
            const float inv = 1.0f / lt;
            const float4 o = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
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
template <int R, typename KV>
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
            kk[j] = KvIo<KV>::round4(*reinterpret_cast<const float4 *>(base + (size_t)min(p0 + 4 * j + grp, P - 1) * 3 * d + d));
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
            xx[j] = KvIo<KV>::round4(*reinterpret_cast<const float4 *>(base + (size_t)min(p0 + 4 * j + grp, P - 1) * 3 * d + 2 * d));
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

// Prefill / CLIP-tower attention on the matrix cores (round 6) for sequences of 24 .. 128 positions (the 77-token text
// tower, the 50-token ViT, long caption prefixes): the scores and the weighted sum of a head ARE two dense GEMMs
// (S x 64 x S and S x S x 64), and the per-row wavefront kernel above spends four xor-shuffle reductions per (query, key)
// pair on them (2.4 ms per launch of 4000 captions x 8 heads x 77 tokens: 38 % of the text tower).
//   block = one (caption, head); wavefront w = the 32 queries [32 w, 32 w + 32).
//   K and V of the head are staged in LDS ONCE per block as two fp16 planes each (a = hi + 2^-11 lo, bf16x3.h: three MFMAs
//   per product, fp32-accurate like the default GEMM mode -- whatever the tower's GEMM precision, the attention keeps it):
//   K row-major [key][64] (144-byte rows: conflict-free 16-byte fragment reads), V TRANSPOSED [dim][key] with the keys of
//   every 16-key group stored in the order the score accumulators hold them (below), so a V fragment is one 16-byte read.
//   scores^T tile = K_tile (A operand: rows = keys) x Q_tile^T (B operand: columns = queries; fragments straight from the
//   qkv activations, pre-scaled by 1/8 log2 e): in the 32 x 32 accumulator layout lane l then holds, for ITS query
//   i = l & 31, the keys (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the tile in registers r = 0 .. 15 -- a softmax row is spread
//   over one lane's registers and its partner lane l ^ 32 only: max and sum are register loops plus ONE shuffle.
//   The un-normalised weights are already the A operand of P x V (row = query l & 31, eight k values per lane half): registers
//   8 s .. 8 s + 7 cover keys 16 s .. 16 s + 15 of the tile in the order (e & 3) + 8 (e >> 2) + 4 half -- the order V^T is
//   stored in.  out tile = P x V lands row = query, column = dimension; it goes through a per-wavefront LDS slab (the K / V
//   area, after a block barrier) so that rows leave as whole 16-byte quads: fp32 or the packed A operand of c_proj.
//   Causal rows never touch key tiles above their own (wavefront w computes w + 1 of them).
// one fp16 plane (RNE), clamped to fp16's range and counted like every 16-bit GEMM operand of this library
__device__ __forceinline__ f16x4 round1h(const float4 v) {
    (void)h2_clamp_count(v);
    f16x4 h;
    h[0] = (_Float16)h2_clamp(v.x); h[1] = (_Float16)h2_clamp(v.y); h[2] = (_Float16)h2_clamp(v.z); h[3] = (_Float16)h2_clamp(v.w);
    return h;
}
constexpr int AM_KST = 72;      // halfs per staged K row (64 + 8: 144 bytes)
constexpr int AM_OST = 68;      // floats per row of the output slab (64 + 4)
inline size_t attn_mfma_lds_bytes(int S, int planes) {
    const int nqt = (S + 31) / 32, Spad = nqt * 32;
    const size_t kv = (size_t)planes * Spad * AM_KST * 2 + (size_t)planes * 64 * (Spad + 8) * 2;
    const size_t slab = (size_t)nqt * 32 * AM_OST * 4;
    return kv > slab ? kv : slab;
}
// (__launch_bounds__(256, 3): <= 168 registers, so that three of the 77-token tower's 192-thread blocks -- 3 x 54 KB of LDS --
//  share a CU; left alone the compiler takes 194-258 registers: two blocks per CU, or ONE wavefront per SIMD for the
//  non-causal forms)
// SPLIT = false (the towers' fp16 / bf16 precision modes, where every GEMM operand of the block stack is 16 bits already):
// ONE fp16 plane per operand of the two attention products -- q, k, v and the un-normalised softmax weights rounded to fp16
// (RNE, clamped to +-65504), one MFMA per product: the arithmetic class of the reference's CLIP on a GPU; half the LDS, a
// third of the MFMAs and -- what matters: the kernel is bound by its VALU conversions -- less than half the conversion work.
template <bool CAUSAL, int NT, bool SPLIT>      // NT: key tiles the score registers are sized for (S <= 32 NT)
__global__ __launch_bounds__(256, 3) void attn_prefill_mfma_kernel(const float *__restrict__ qkv, int heads, int S, int d,
                                                                float *__restrict__ out, char *__restrict__ packed_out, int fmt) {
    extern __shared__ __attribute__((aligned(16))) char am_smem[];
    typedef float f32x16a __attribute__((ext_vector_type(16)));
    const int nqt = (S + 31) >> 5, Spad = nqt * 32, VST = Spad + 8;       // VST: halfs per V^T row
    constexpr int NPL = SPLIT ? 2 : 1;
    _Float16 *Kp = reinterpret_cast<_Float16 *>(am_smem);                 // [NPL planes][Spad][AM_KST]
    _Float16 *Vt = Kp + (size_t)NPL * Spad * AM_KST;                      // [NPL planes][64][VST]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, half = lane >> 5;
    const int head = blockIdx.x % heads, cap = blockIdx.x / heads;
    const float *__restrict__ base = qkv + (size_t)cap * S * 3 * d + head * 64;
    constexpr float LO = 1.0f / H2_LO_SCALE;
    // ---- every global load of the block is issued BEFORE the first one is used (the block is 64 nqt threads and the head
    // Spad = 32 nqt keys of 16 quads: exactly eight (key, quad) items per thread, whatever S) -- the first version waited
    // for each item's K / V pair in turn: eight dependent memory round trips per block, 19 us of a block's 20
    const int i = wave * 32 + l32;                                        // this lane's query
    float4 qa[4], qb[4], kx[8], vx[8];
    {
        const float *qr = base + (size_t)min(i, S - 1) * 3 * d + 8 * half;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qa[s] = *reinterpret_cast<const float4 *>(qr + 16 * s);
            qb[s] = *reinterpret_cast<const float4 *>(qr + 16 * s + 4);
        }
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = tid + it * (int)blockDim.x, key = idx >> 4, q4 = idx & 15;
        const float *r = base + (size_t)min(key, S - 1) * 3 * d + q4 * 4;
        kx[it] = *reinterpret_cast<const float4 *>(r + d);
        vx[it] = *reinterpret_cast<const float4 *>(r + 2 * d);
    }
    // ---- stage K and V^T (rows past S: zeros -- a zero weight times a NaN would still poison the P x V accumulators)
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = tid + it * (int)blockDim.x, key = idx >> 4, q4 = idx & 15;
        float4 kv = kx[it], vv = vx[it];
        if (key >= S) { kv = make_float4(0.f, 0.f, 0.f, 0.f); vv = kv; }
        f16x4 kh, kl, vh, vl;
        if constexpr (SPLIT) { split2h(kv, kh, kl); split2h(vv, vh, vl); }
        else { kh = round1h(kv); vh = round1h(vv); }
        *reinterpret_cast<f16x4 *>(Kp + (size_t)key * AM_KST + q4 * 4) = kh;
        if constexpr (SPLIT) *reinterpret_cast<f16x4 *>(Kp + (size_t)(Spad + key) * AM_KST + q4 * 4) = kl;
        const int w = key & 15;
        const int pos = (key & ~15) + 8 * ((w >> 2) & 1) + (w & 3) + 4 * (w >> 3);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            Vt[(size_t)(q4 * 4 + u) * VST + pos] = vh[u];
            if constexpr (SPLIT) Vt[(size_t)(64 + q4 * 4 + u) * VST + pos] = vl[u];
        }
    }
    // ---- this wavefront's query fragments (B operand of the score product): row i, k = 16 s + 8 half .. + 7
    f16x8 qh[4], ql[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float4 a = qa[s], b = qb[s];
        a.x *= ATT_QSCALE; a.y *= ATT_QSCALE; a.z *= ATT_QSCALE; a.w *= ATT_QSCALE;
        b.x *= ATT_QSCALE; b.y *= ATT_QSCALE; b.z *= ATT_QSCALE; b.w *= ATT_QSCALE;
        f16x4 h0, l0, h1, l1;
        if constexpr (SPLIT) { split2h(a, h0, l0); split2h(b, h1, l1); }
        else { h0 = round1h(a); h1 = round1h(b); l0 = h0; l1 = h1; }
#pragma unroll
        for (int e = 0; e < 4; ++e) { qh[s][e] = h0[e]; qh[s][4 + e] = h1[e]; ql[s][e] = l0[e]; ql[s][4 + e] = l1[e]; }
    }
    __syncthreads();
    // ---- scores^T = K Q^T, tile by tile; sc[t][r]: query i, key 32 t + (r & 3) + 8 (r >> 2) + 4 half
    const int nt = CAUSAL ? wave + 1 : nqt;
    float sc[NT][16];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t < nt) {
            f32x16a am, ac;
#pragma unroll
            for (int r = 0; r < 16; ++r) { am[r] = 0.f; ac[r] = 0.f; }
            const _Float16 *kr = Kp + (size_t)(32 * t + l32) * AM_KST + 8 * half;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f16x8 kh = *reinterpret_cast<const f16x8 *>(kr + 16 * s);
                if constexpr (SPLIT) {
                    const f16x8 kl = *reinterpret_cast<const f16x8 *>(kr + (size_t)Spad * AM_KST + 16 * s);
                    ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s], ac, 0, 0, 0);
                    am = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s], am, 0, 0, 0);
                    ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s], ac, 0, 0, 0);
                } else
                    am = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s], am, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
                const bool vis = key < S && (!CAUSAL || key <= i);
                sc[t][r] = vis ? (SPLIT ? am[r] + ac[r] * LO : am[r]) : -INFINITY;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[t][r] = -INFINITY;
        }
    }
    // ---- softmax over the row: this lane's registers and the partner lane's (key 0 is visible to every query: finite max)
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, sc[t][r]);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float lsum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = att_exp2(sc[t][r] - m);
            sc[t][r] = p;
            lsum += p;
        }
    lsum += __shfl_xor(lsum, 32, 64);
    const float linv = 1.0f / lsum;
    // ---- out = P V: the weights of registers 8 s2 .. 8 s2 + 7 are the A fragment of k-step (t, s2)
    f32x16a om[2], oc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) { om[n][r] = 0.f; oc[n][r] = 0.f; }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t < nt) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                f16x8 ph, pl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float p = sc[t][8 * s2 + e];                     // 0 <= p <= 1
                    if constexpr (SPLIT) {
                        const _Float16 hh = p < 0x1p-14f ? (_Float16)0.f : (_Float16)p;
                        ph[e] = hh;
                        pl[e] = (_Float16)((p - (float)hh) * H2_LO_SCALE);
                    } else
                        ph[e] = (_Float16)p;
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const _Float16 *vr = Vt + (size_t)(32 * n + l32) * VST + 32 * t + 16 * s2 + 8 * half;
                    const f16x8 vh = *reinterpret_cast<const f16x8 *>(vr);
                    if constexpr (SPLIT) {
                        const f16x8 vl = *reinterpret_cast<const f16x8 *>(vr + (size_t)64 * VST);
                        oc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, vh, oc[n], 0, 0, 0);
                        om[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vh, om[n], 0, 0, 0);
                        oc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vl, oc[n], 0, 0, 0);
                    } else
                        om[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vh, om[n], 0, 0, 0);
                }
            }
        }
    }
    // ---- rows out through this wavefront's slab: accumulator (row = query (r & 3) + 8 (r >> 2) + 4 half, column = l32 + 32 n)
    __syncthreads();                                   // every wavefront is done with K / V
    float *ost = reinterpret_cast<float *>(am_smem) + (size_t)wave * 32 * AM_OST;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const float li = __shfl(linv, row, 64);        // lane `row` holds the sum of query 32 w + row
#pragma unroll
        for (int n = 0; n < 2; ++n) ost[row * AM_OST + 32 * n + l32] = (SPLIT ? om[n][r] + oc[n][r] * LO : om[n][r]) * li;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int q = lane; q < 32 * 16; q += 64) {
        const int row = q >> 4, c4 = q & 15, grow = wave * 32 + row;
        if (grow >= S) continue;
        const float4 o = *reinterpret_cast<const float4 *>(ost + row * AM_OST + c4 * 4);
        const int orow = cap * S + grow;
        if (packed_out) x3_store_quad(packed_out, d >> 4, orow, head * 4 + (c4 >> 2), c4 & 3, o, fmt);
        else reinterpret_cast<float4 *>(out + (size_t)orow * d + head * 64)[c4] = o;
    }
}

// K/V of prefill row (cap, i) -> cache[phys = cap*beam][head][i][:]
template <typename KV>
__global__ void kv_scatter_prefill_kernel(const float *__restrict__ qkv, KV *__restrict__ kc,
                                          KV *__restrict__ vc, int ncap, int P, int beam, int heads, int ctx,
                                          int d) {
    const int nv = d / 4;                       // float4 per row per tensor
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncap * P * nv) return;
    const int c4 = i % nv, row = i / nv;
    const int cap = row / P, pos = row - cap * P;
    const int head = (c4 * 4) / 64, within = (c4 * 4) % 64;
    const size_t o = (((size_t)cap * beam * heads + head) * ctx + pos) * 64 + within;
    const float *r = qkv + (size_t)row * 3 * d;
    KvIo<KV>::st4(kc + o, reinterpret_cast<const float4 *>(r + d)[c4]);
    KvIo<KV>::st4(vc + o, reinterpret_cast<const float4 *>(r + 2 * d)[c4]);
}

int launch_kv_scatter_prefill(hipStream_t st, const float *qkv, const KvCache &c, int layer, int ncap, int P,
                              int beam) {
    const int d = c.heads * c.hd;
    const int tot = ncap * P * (d / 4);
    if (tot <= 0) return 0;
    if (c.bf16)
        hipLaunchKernelGGL(kv_scatter_prefill_kernel<__bf16>, dim3((tot + 255) / 256), dim3(256), 0, st, qkv,
                           c.kp<__bf16>(layer), c.vp<__bf16>(layer), ncap, P, beam, c.heads, c.ctx, d);
    else
        hipLaunchKernelGGL(kv_scatter_prefill_kernel<float>, dim3((tot + 255) / 256), dim3(256), 0, st, qkv,
                           c.kp<float>(layer), c.vp<float>(layer), ncap, P, beam, c.heads, c.ctx, d);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

int launch_attn_prefill(hipStream_t st, const float *qkv, const KvCache &c, int layer, int ncap, int P, int beam,
                        float *out, bool causal, void *packed_out, int fmt) {
    CAPDEC_CHECK(c.hd == 64, "attention: head_dim must be 64");
    CAPDEC_CHECK(P <= ATT_CTX_MAX && P <= c.ctx, "attention: prefix longer than the supported context");
    (void)layer; (void)beam;                      // K / V come straight from qkv (the cache is filled by kv_scatter_prefill)
    // one fp16 plane per operand in the TOWERS' 16-bit precision modes (a tower keeps no KV cache: c.k == nullptr), two
    // (fp32-accurate) otherwise.  A tower takes this kernel at every length up to 128: the text tower computes as many
    // positions as its chunk's longest caption has (typically ~20), and a row's arithmetic class must not depend on that
    const bool tower = c.k == nullptr;
    const bool split = !((fmt == PK_F16X1 || fmt == PK_BF16X1) && tower);
    if (!c.bf16 && (P >= ATT_MFMA_MIN_P || tower) && P <= ATT_MFMA_MAX_P && ncap > 0) {
        const int nqt = (P + 31) / 32;
        const size_t lds = attn_mfma_lds_bytes(P, split ? 2 : 1);
#define LAUNCH_AM(CZ, NTV)                                                                                          \
    {                                                                                                               \
        if (lds > 64 * 1024)      /* (per launch, not latched: the limit belongs to the function on the CURRENT device) */ \
            CAPDEC_HIP(hipFuncSetAttribute((const void *)attn_prefill_mfma_kernel<CZ, NTV, true>,                   \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                  \
        if (split)                                                                                                  \
            hipLaunchKernelGGL((attn_prefill_mfma_kernel<CZ, NTV, true>), dim3(ncap * c.heads), dim3(64 * nqt), lds, st, qkv, \
                               c.heads, P, c.heads * c.hd, out, (char *)packed_out, fmt);                            \
        else                                                                                                        \
            hipLaunchKernelGGL((attn_prefill_mfma_kernel<CZ, NTV, false>), dim3(ncap * c.heads), dim3(64 * nqt), lds, st, qkv, \
                               c.heads, P, c.heads * c.hd, out, (char *)packed_out, fmt);                            \
    }
#define LAUNCH_AM_NT(CZ)                                                                                            \
    switch (nqt) {                                                                                                  \
        case 1: LAUNCH_AM(CZ, 1) break;                                                                             \
        case 2: LAUNCH_AM(CZ, 2) break;                                                                             \
        case 3: LAUNCH_AM(CZ, 3) break;                                                                             \
        default: LAUNCH_AM(CZ, 4) break;                                                                            \
    }
        if (causal) { LAUNCH_AM_NT(true) } else { LAUNCH_AM_NT(false) }
#undef LAUNCH_AM_NT
#undef LAUNCH_AM
        CAPDEC_HIP(hipGetLastError());
        return 0;
    }
    constexpr int R = 8;
    const int total = ncap * c.heads * ((P + R - 1) / R);
    if (total <= 0) return 0;
    const size_t lds = (size_t)4 * R * P * sizeof(float);
    CAPDEC_CHECK(lds <= 160 * 1024, "attention: prefix too long for the score rows in LDS");
    ATT_BIG_LDS((attn_prefill_rows_kernel<R, __bf16>), lds);
    ATT_BIG_LDS((attn_prefill_rows_kernel<R, float>), lds);
    if (c.bf16)     // the keys / values this pass attends to are the bf16-rounded ones the cache will hold
        hipLaunchKernelGGL((attn_prefill_rows_kernel<R, __bf16>), dim3((total + 3) / 4), dim3(256), lds, st, qkv, total,
                           c.heads, P, c.heads * c.hd, causal ? 1 : 0, out, (char *)packed_out, fmt);
    else
        hipLaunchKernelGGL((attn_prefill_rows_kernel<R, float>), dim3((total + 3) / 4), dim3(256), lds, st, qkv, total,
                           c.heads, P, c.heads * c.hd, causal ? 1 : 0, out, (char *)packed_out, fmt);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

template <typename KV>
static int attn_decode_typed(hipStream_t st, const float *qkv, const KvCache &c, int layer, int rows, int beam, int L,
                             const uint8_t *anc, int anc_stride, float *out, void *packed_out, const int *cmap,
                             int fmt, bool cur_cached) {
    KV *kl = c.kp<KV>(layer), *vl = c.vp<KV>(layer);
    // positions known to sit in slot 0 whatever the table says: the whole history for greedy rows, the CLIP prefix for
    // beams (c.prefix_len; 0 = unknown)
    const Tuning &tn = c.tune ? *c.tune : default_tuning();     // (the overrides below exist in measurement builds only)
    const int pre_on = tn.att_preload;
    const int npre = !pre_on ? 0 : (anc == nullptr ? L : c.prefix_len);
    // (Round 5 tried a kernel of its own for greedy rows on the bf16 cache -- eight lanes per key, 16-byte loads, the K / V
    //  of the whole history requested before the first use -- and CAPDEC_ATT_NA=4 here: 157 / 143 us against 150 / 141 us per
    //  launch at 5000 captions.  That launch is not latency-bound: 5.5 KB runs at a 9.7 KB stride read at 4.4 TB/s whatever
    //  the structure (profiles/r5_att_greedy_b16_ab.txt).)
    {
        const int ncap = rows / beam, total = ncap * c.heads;
        if (total <= 0) return 0;
        size_t lds = ((size_t)4 * beam * L * sizeof(int) + 1023) & ~(size_t)1023;   // ancestor slots (the DMA ring, if any, follows)
        CAPDEC_CHECK(lds + 32 * 1024 <= 160 * 1024, "attention: context too long for the slot table in LDS");
        const int wsync = tn.att_wsync;
        const int dma_on = tn.att_dma && wsync;
        dim3 grid((total + 3) / 4), block(256);
        // waves per SIMD the register allocation is sized for: beam <= 4 fits 4 without spilling; beam 5 needs 124
        // registers at 4 waves; CAPDEC_ATT_OCC=3 / CAPDEC_ATT_NA=4 are measurement knobs (default = measured best)
        const int occ5 = tn.att_occ == 3 ? 3 : 4;
        // positions per group in flight (NA): 2 when the launch is HBM-bound (5000 captions: 0.429 vs 0.435 ms), 4 when
        // fewer than two rounds of wavefronts make it latency-bound (625 captions: 66.6 vs 68.7 us); CAPDEC_ATT_NA forces
        const int na_env = tn.att_na;
        const int na4 = na_env ? (na_env == 4) : (!c.fixed_variant && total <= 16384);
#define LAUNCH_BEAMS_V(B, OCC, NAV, CURV)                                                                       \
    ATT_BIG_LDS((attn_decode_beams_kernel<B, KV, OCC, NAV, CURV, false>), lds);                                    \
    hipLaunchKernelGGL((attn_decode_beams_kernel<B, KV, OCC, NAV, CURV, false>), grid, block, lds, st, qkv, kl, vl, total, \
                       c.heads, c.ctx, c.heads * c.hd, L, anc, anc_stride, out, (char *)packed_out, cmap, fmt, npre, wsync ? 0 : -1)
    // (LDS-DMA variant: NA = 2 for every launch size -- with its double buffer 16 positions per group are in flight, what
    //  NA = 4 gives the register-landed loop, and 38 KB of LDS per block still lets four blocks share a CU)
#define LAUNCH_BEAMS_DMA(B, OCC)                                                                                \
    ATT_BIG_LDS((attn_decode_beams_kernel<B, float, OCC, 2, true, true>), lds + 4 * 2 * (2 * 2 * 1024));          \
    hipLaunchKernelGGL((attn_decode_beams_kernel<B, float, OCC, 2, true, true>), grid, block, lds + 4 * 2 * (2 * 2 * 1024), st, \
                       qkv, (float *)kl, (float *)vl, total, c.heads, c.ctx, c.heads * c.hd, L, anc, anc_stride, out,  \
                       (char *)packed_out, cmap, fmt, npre, (int)lds)
#define LAUNCH_BEAMS(B, OCC)                                                                                    \
    if (cur_cached && (B == 1 || B == 5)) {       /* (the widths the decode drivers use most: greedy and beam 5) */ \
        if (dma_on && sizeof(KV) == 4) { LAUNCH_BEAMS_DMA(B, OCC); }    /* (the LDS-DMA ring is laid out for fp32 keys) */  \
        else if (na4) { LAUNCH_BEAMS_V(B, OCC, 4, true); } else { LAUNCH_BEAMS_V(B, OCC, 2, true); }             \
    } else if (na4 && B <= 5) { LAUNCH_BEAMS_V(B, OCC, 4, false); }                                             \
    else { LAUNCH_BEAMS_V(B, OCC, 2, false); }
        switch (beam) {
            case 1: LAUNCH_BEAMS(1, 4); break;      // greedy: the same single-pass kernel with one row per caption
            case 2: LAUNCH_BEAMS(2, 4); break;
            case 3: LAUNCH_BEAMS(3, 4); break;
            case 4: LAUNCH_BEAMS(4, 4); break;
            case 5: if (occ5 == 3) { LAUNCH_BEAMS(5, 3); } else { LAUNCH_BEAMS(5, 4); } break;
            case 6: LAUNCH_BEAMS(6, 2); break;
            case 7: LAUNCH_BEAMS(7, 2); break;
            case 8: LAUNCH_BEAMS(8, 2); break;
            default: CAPDEC_CHECK(false, "attention: beam must be in 1..8");
        }
#undef LAUNCH_BEAMS
#undef LAUNCH_BEAMS_DMA
#undef LAUNCH_BEAMS_V
        CAPDEC_HIP(hipGetLastError());
        return 0;
    }
}

int launch_attn_decode(hipStream_t st, const float *qkv, const KvCache &c, int layer, int rows, int beam, int L,
                       const uint8_t *anc, int anc_stride, float *out, void *packed_out, const int *cmap, int fmt,
                       bool cur_cached) {
    CAPDEC_CHECK(c.hd == 64, "attention: head_dim must be 64");
    CAPDEC_CHECK(L >= 1 && L <= ATT_CTX_MAX && L <= c.ctx, "attention: context length out of range");
    CAPDEC_CHECK(!cur_cached || beam == 1 || beam == 5, "attention: cur_cached needs beam 1 or 5");
    return c.bf16 ? attn_decode_typed<__bf16>(st, qkv, c, layer, rows, beam, L, anc, anc_stride, out, packed_out, cmap, fmt, cur_cached)
                  : attn_decode_typed<float>(st, qkv, c, layer, rows, beam, L, anc, anc_stride, out, packed_out, cmap, fmt, cur_cached);
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

CAPDEC_SAT_ACCESSOR(sat_count_attention)

}  // namespace capdec
