// Token selection: merge of the lm_head per-tile candidates, greedy step, and the beam-search
// bookkeeping of reference gpt2_prefix_eval.py:78-108 (one wavefront per caption; all
// per-caption state lives in a few hundred bytes, so these kernels are launch-latency sized).
#include "common.h"

namespace capdec {

constexpr int SEL_T_MAX = 1024;    // max entry_length
constexpr int SEL_CTX_MAX = 1024;  // max context (prefix + generated): GPT-2's n_positions, the reference's only limit
                                   // (the beam step stages beam x (T ints + ctx bytes) of state in dynamic LDS: <= 40 KB)
constexpr int SEL_BEAM_MAX = 8;

__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

// ---- merge: per row, logsumexp over tiles and the global top-k of the per-tile top-k lists.
template <int KSEL>
__global__ __launch_bounds__(256) void topk_merge_kernel(const float *__restrict__ tile_max,
                                                         const float *__restrict__ tile_sum,
                                                         const float *__restrict__ cand_val,
                                                         const int *__restrict__ cand_idx, int rows, int ntiles,
                                                         float *__restrict__ lse, float *__restrict__ top_val,
                                                         int *__restrict__ top_idx) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const size_t base = (size_t)row * ntiles;
    float m = -INFINITY;
    for (int t = lane; t < ntiles; t += 64) m = fmaxf(m, tile_max[base + t]);
    m = wave_max(m);
    float s = 0.f;
    for (int t = lane; t < ntiles; t += 64) s += tile_sum[base + t] * expf(tile_max[base + t] - m);
    s = wave_sum(s);
    if (lane == 0) lse[row] = m + logf(s);

    float bv[KSEL];
    int bi[KSEL];
#pragma unroll
    for (int j = 0; j < KSEL; ++j) { bv[j] = -INFINITY; bi[j] = 0x7fffffff; }
    for (int t = lane; t < ntiles; t += 64) {
#pragma unroll
        for (int kk = 0; kk < KSEL; ++kk) {
            float v = cand_val[(base + t) * KSEL + kk];
            int i = cand_idx[(base + t) * KSEL + kk];
#pragma unroll
            for (int j = 0; j < KSEL; ++j) {
                if (better(v, i, bv[j], bi[j])) {
                    const float tv = bv[j]; const int ti = bi[j];
                    bv[j] = v; bi[j] = i; v = tv; i = ti;
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < KSEL; ++r) {
        float gv = bv[0];
        int gi = bi[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(gv, o, 64);
            const int oi = __shfl_xor(gi, o, 64);
            if (better(ov, oi, gv, gi)) { gv = ov; gi = oi; }
        }
        if (gi == bi[0] && gv == bv[0]) {   // this lane held the winner: pop it
#pragma unroll
            for (int j = 0; j + 1 < KSEL; ++j) { bv[j] = bv[j + 1]; bi[j] = bi[j + 1]; }
            bv[KSEL - 1] = -INFINITY; bi[KSEL - 1] = 0x7fffffff;
        }
        if (lane == 0) { top_val[(size_t)row * KSEL + r] = gv; top_idx[(size_t)row * KSEL + r] = gi; }
    }
}

// ---- k = 3 per tile in, top 5 out, with the "may hide a fourth" flag (common.h)
__global__ __launch_bounds__(256) void topk_merge_k3_kernel(const float *__restrict__ tile_max, const float *__restrict__ tile_sum,
                                                            const float *__restrict__ cand_val, const int *__restrict__ cand_idx,
                                                            int rows, int ntiles, float *__restrict__ lse,
                                                            float *__restrict__ top_val, int *__restrict__ top_idx,
                                                            int *__restrict__ flag_rows, int *__restrict__ flag_count,
                                                            int *__restrict__ flag_total) {
    constexpr int KIN = 3, KOUT = 5;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const size_t base = (size_t)row * ntiles;
    float m = -INFINITY;
    for (int t = lane; t < ntiles; t += 64) m = fmaxf(m, tile_max[base + t]);
    m = wave_max(m);
    float s = 0.f;
    for (int t = lane; t < ntiles; t += 64) s += tile_sum[base + t] * expf(tile_max[base + t] - m);
    s = wave_sum(s);
    if (lane == 0) lse[row] = m + logf(s);

    float bv[KOUT];
    int bi[KOUT];
#pragma unroll
    for (int j = 0; j < KOUT; ++j) { bv[j] = -INFINITY; bi[j] = 0x7fffffff; }
    float l3v = -INFINITY;                 // the best LAST-kept candidate among this lane's tiles: what bounds their hidden ones
    int l3i = 0x7fffffff;
    for (int t = lane; t < ntiles; t += 64) {
#pragma unroll
        for (int kk = 0; kk < KIN; ++kk) {
            float v = cand_val[(base + t) * KIN + kk];
            int i = cand_idx[(base + t) * KIN + kk];
            if (kk == KIN - 1 && better(v, i, l3v, l3i)) { l3v = v; l3i = i; }
#pragma unroll
            for (int j = 0; j < KOUT; ++j) {
                if (better(v, i, bv[j], bi[j])) {
                    const float tv = bv[j]; const int ti = bi[j];
                    bv[j] = v; bi[j] = i; v = tv; i = ti;
                }
            }
        }
    }
    float gv = -INFINITY;
    int gi = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < KOUT; ++r) {
        gv = bv[0];
        gi = bi[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(gv, o, 64);
            const int oi = __shfl_xor(gi, o, 64);
            if (better(ov, oi, gv, gi)) { gv = ov; gi = oi; }
        }
        if (gi == bi[0] && gv == bv[0]) {   // this lane held the winner: pop it
#pragma unroll
            for (int j = 0; j + 1 < KOUT; ++j) { bv[j] = bv[j + 1]; bi[j] = bi[j + 1]; }
            bv[KOUT - 1] = -INFINITY; bi[KOUT - 1] = 0x7fffffff;
        }
        if (lane == 0) { top_val[(size_t)row * KOUT + r] = gv; top_idx[(size_t)row * KOUT + r] = gi; }
    }
    // (gv, gi) = the row's fifth.  A tile's hidden candidates are all worse than its third kept one, so they can only
    // matter if that third one is STRICTLY better than the fifth (if it IS the fifth, or worse, nothing hidden can pass it)
    const bool mine = better(l3v, l3i, gv, gi);
    if (__ballot(mine) != 0ull && lane == 0) {
        flag_rows[atomicAdd(flag_count, 1)] = row;
        atomicAdd(flag_total, 1);
    }
}

int launch_topk_merge_k3(hipStream_t st, const float *tile_max, const float *tile_sum, const float *cand_val,
                         const int *cand_idx, int rows, int ntiles, float *lse, float *top_val, int *top_idx,
                         int *flag_rows, int *flag_count, int *flag_total) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(topk_merge_k3_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, tile_max, tile_sum, cand_val, cand_idx,
                       rows, ntiles, lse, top_val, top_idx, flag_rows, flag_count, flag_total);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// ---- the second pass's merge: compact row i (k = 5 lists) -> top 5 of row out_rows[i]; persistent over *count_dev rows
__global__ __launch_bounds__(256) void topk_merge_rows_kernel(const float *__restrict__ cand_val, const int *__restrict__ cand_idx,
                                                              const int *__restrict__ count_dev, const int *__restrict__ out_rows,
                                                              int ntiles, float *__restrict__ top_val, int *__restrict__ top_idx) {
    constexpr int KSEL = 5;
    const int lane = threadIdx.x & 63;
    const int n = *count_dev;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += gridDim.x * 4) {
        const size_t base = (size_t)row * ntiles;
        float bv[KSEL];
        int bi[KSEL];
#pragma unroll
        for (int j = 0; j < KSEL; ++j) { bv[j] = -INFINITY; bi[j] = 0x7fffffff; }
        for (int t = lane; t < ntiles; t += 64) {
#pragma unroll
            for (int kk = 0; kk < KSEL; ++kk) {
                float v = cand_val[(base + t) * KSEL + kk];
                int i = cand_idx[(base + t) * KSEL + kk];
#pragma unroll
                for (int j = 0; j < KSEL; ++j) {
                    if (better(v, i, bv[j], bi[j])) {
                        const float tv = bv[j]; const int ti = bi[j];
                        bv[j] = v; bi[j] = i; v = tv; i = ti;
                    }
                }
            }
        }
        const size_t orow = (size_t)out_rows[row];
#pragma unroll
        for (int r = 0; r < KSEL; ++r) {
            float gv = bv[0];
            int gi = bi[0];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(gv, o, 64);
                const int oi = __shfl_xor(gi, o, 64);
                if (better(ov, oi, gv, gi)) { gv = ov; gi = oi; }
            }
            if (gi == bi[0] && gv == bv[0]) {
#pragma unroll
                for (int j = 0; j + 1 < KSEL; ++j) { bv[j] = bv[j + 1]; bi[j] = bi[j + 1]; }
                bv[KSEL - 1] = -INFINITY; bi[KSEL - 1] = 0x7fffffff;
            }
            if (lane == 0) { top_val[orow * KSEL + r] = gv; top_idx[orow * KSEL + r] = gi; }
        }
    }
}

int launch_topk_merge_rows(hipStream_t st, const float *cand_val, const int *cand_idx, const int *count_dev,
                           const int *out_rows, int rows_cap, int ntiles, float *top_val, int *top_idx) {
    if (rows_cap <= 0) return 0;
    hipLaunchKernelGGL(topk_merge_rows_kernel, dim3(std::min((rows_cap + 3) / 4, 1024)), dim3(256), 0, st, cand_val, cand_idx,
                       count_dev, out_rows, ntiles, top_val, top_idx);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

int launch_topk_merge(hipStream_t st, const float *tile_max, const float *tile_sum, const float *cand_val,
                      const int *cand_idx, int rows, int ntiles, int k, float *lse, float *top_val, int *top_idx) {
    if (rows <= 0) return 0;
    dim3 grid((rows + 3) / 4), block(256);
#define LAUNCH_MERGE(KS)                                                                                       \
    hipLaunchKernelGGL(topk_merge_kernel<KS>, grid, block, 0, st, tile_max, tile_sum, cand_val, cand_idx, rows, \
                       ntiles, lse, top_val, top_idx)
    switch (k) {
        case 1: LAUNCH_MERGE(1); break;
        case 2: LAUNCH_MERGE(2); break;
        case 3: LAUNCH_MERGE(3); break;
        case 4: LAUNCH_MERGE(4); break;
        case 5: LAUNCH_MERGE(5); break;
        case 6: LAUNCH_MERGE(6); break;
        case 7: LAUNCH_MERGE(7); break;
        case 8: LAUNCH_MERGE(8); break;
        default: CAPDEC_CHECK(false, "topk_merge: k must be in 1..8");
    }
#undef LAUNCH_MERGE
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// ---- greedy: reference gpt2_prefix_eval.py:177-188 (argmax; stop on '.' or id 764)
// (k candidates per row: k = 1 normally; teacher forcing -- forced != nullptr -- feeds forced[row, step] as the next
//  token instead of the arg-max and, with k = 2, records (top-1 logit, top-2 logit, logsumexp) of every step)
__global__ void greedy_step_kernel(const int *__restrict__ top_idx, int rows, int step, int T, int stop_id,
                                   int alt_stop_id, int *__restrict__ ids, int *__restrict__ lens,
                                   uint8_t *__restrict__ done, int *__restrict__ next_tok,
                                   int *__restrict__ alive_count, const int *__restrict__ cmap, int k,
                                   const int *__restrict__ forced, const float *__restrict__ top_val,
                                   const float *__restrict__ lse, float *__restrict__ stats) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;       // activation row (compact)
    if (r >= rows) return;
    const int row = cmap ? cmap[r] : r;                        // caption
    const int tok = top_idx[(size_t)r * k];
    if (stats) {
        float *o = stats + ((size_t)row * T + step) * 3;
        o[0] = top_val[(size_t)r * k];
        o[1] = k > 1 ? top_val[(size_t)r * k + 1] : -INFINITY;
        o[2] = lse[r];
    }
    if (forced) {                                              // teacher forcing: nothing stops, every step is recorded
        next_tok[row] = forced[(size_t)row * T + step];
        ids[(size_t)row * T + step] = tok;
        lens[row] = step + 1;
        atomicAdd(alive_count, 1);
        return;
    }
    next_tok[row] = tok;
    if (done[row]) return;
    ids[(size_t)row * T + step] = tok;
    lens[row] = step + 1;
    if (tok == stop_id || tok == alt_stop_id) done[row] = 1;
    else atomicAdd(alive_count, 1);
}

int launch_greedy_step(hipStream_t st, const int *top_idx, int rows, int step, int T, int stop_id, int alt_stop_id,
                       int *ids, int *lens, uint8_t *done, int *next_tok, int *alive_count, const int *cmap, int k,
                       const int *forced, const float *top_val, const float *lse, float *stats) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(greedy_step_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, top_idx, rows, step, T,
                       stop_id, alt_stop_id, ids, lens, done, next_tok, alive_count, cmap, k, forced, top_val, lse, stats);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// ---- beam step 0: reference gpt2_prefix_eval.py:78-88,105-108
// one wavefront per caption; top_val/top_idx/lse are per CAPTION row here (prefill last position)
__global__ __launch_bounds__(64) void beam_init_kernel(BeamState s, const float *__restrict__ lse,
                                                       const float *__restrict__ top_val,
                                                       const int *__restrict__ top_idx, int ncap, int beam, int k,
                                                       int T, int ctx, int stop_id) {
    const int cap = blockIdx.x, b = threadIdx.x;
    bool stop = true;
    if (b < beam) {
        const int tok = top_idx[(size_t)cap * k + b];
        const float logp = top_val[(size_t)cap * k + b] - lse[cap];   // log softmax
        const size_t cb = (size_t)cap * beam + b;
        s.tokens[cb * T] = tok;
        s.scores[cb] = logp;
        s.seq[cb] = 1.0f;
        stop = tok == stop_id;
        s.stopped[cb] = stop;
        s.next_tok[cb] = tok;
    }
    const bool all = __all(stop);
    if (b == 0) {
        s.done[cap] = all;
        if (!all) atomicAdd(s.alive_count, 1);
    }
}

int launch_beam_init(hipStream_t st, const BeamState &s, const float *lse, const float *top_val, const int *top_idx,
                     int ncap, int beam, int k, int T, int ctx, int P, int stop_id) {
    CAPDEC_CHECK(beam >= 1 && beam <= SEL_BEAM_MAX && k >= beam, "beam: beam size must be in 1..8");
    if (ncap <= 0) return 0;
    hipLaunchKernelGGL(beam_init_kernel, dim3(ncap), dim3(64), 0, st, s, lse, top_val, top_idx, ncap, beam, k, T, ctx,
                       stop_id);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// ---- beam step i >= 1: reference gpt2_prefix_eval.py:89-108.
// lane c < beam*k is candidate (b = c / k, j = c % k).  fp32 arithmetic in the reference's order:
//   scores_sum = scores + logp; seq += ~stopped; avg = scores_sum / seq; top-beam of avg over the
//   flattened [beam * V]; scores = avg * seq[src].
__global__ __launch_bounds__(64) void beam_step_kernel(BeamState s, const float *__restrict__ lse,
                                                       const float *__restrict__ top_val,
                                                       const int *__restrict__ top_idx, int ncap, int beam, int k,
                                                       int T, int ctx, int step, int pos_cur, int vocab,
                                                       int stop_id, const int *__restrict__ cmap) {
    extern __shared__ int sel_dyn[];                            // [beam][T] token history, then [beam][ctx] ancestor bytes
    int *tok_old = sel_dyn;
    uint8_t *anc_old = reinterpret_cast<uint8_t *>(sel_dyn + beam * T);
    __shared__ int w_src[SEL_BEAM_MAX], w_tok[SEL_BEAM_MAX];
    __shared__ float w_key[SEL_BEAM_MAX], seq_new[SEL_BEAM_MAX];
    __shared__ uint8_t st_old[SEL_BEAM_MAX];
    const int lane = threadIdx.x;
    const int cap = cmap ? cmap[blockIdx.x] : blockIdx.x;       // state (tokens, scores, anc ...) by original caption
    if (s.done[cap]) return;
    const size_t cb0 = (size_t)cap * beam;
    const size_t ab0 = (size_t)blockIdx.x * beam;               // lm_head outputs by (compact) activation row
    // stage the state that is permuted in place
    for (int i = lane; i < beam * step; i += 64) {
        const int b = i / step, t = i - b * step;
        tok_old[b * T + t] = s.tokens[(cb0 + b) * T + t];
    }
    for (int i = lane; i < beam * pos_cur; i += 64) {
        const int b = i / pos_cur, p = i - b * pos_cur;
        anc_old[b * ctx + p] = s.anc[(cb0 + b) * ctx + p];
    }
    if (lane < beam) {
        const bool stp = s.stopped[cb0 + lane];
        st_old[lane] = stp;
        seq_new[lane] = s.seq[cb0 + lane] + (stp ? 0.0f : 1.0f);
    }
    __syncthreads();
    // candidate of this lane
    float key = -INFINITY;
    int flat = 0x7fffffff, ctok = 0, cb = 0;
    if (lane < beam * k) {
        cb = lane / k;
        const int j = lane - cb * k;
        const size_t row = cb0 + cb, arow = ab0 + cb;
        if (st_old[cb]) {
            if (j == 0) { ctok = 0; key = (s.scores[row] + 0.0f) / seq_new[cb]; }
        } else {
            ctok = top_idx[arow * k + j];
            const float logp = top_val[arow * k + j] - lse[arow];
            key = (s.scores[row] + logp) / seq_new[cb];
        }
        if (key > -INFINITY || (st_old[cb] && j == 0)) flat = cb * vocab + ctok;
        if (s.diverge && j != 0) { key = -INFINITY; flat = 0x7fffffff; }      // measurement only: each beam keeps its own lineage
    }
    for (int r = 0; r < beam; ++r) {
        float gv = key;
        int gi = flat;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(gv, o, 64);
            const int oi = __shfl_xor(gi, o, 64);
            if (better(ov, oi, gv, gi)) { gv = ov; gi = oi; }
        }
        if (gi == flat && gv == key && flat != 0x7fffffff) {   // winner (flat indices are unique)
            w_src[r] = cb; w_tok[r] = ctok; w_key[r] = key;
            key = -INFINITY; flat = 0x7fffffff;
        }
    }
    __syncthreads();
    // write the permuted state
    bool stop = true;
    if (lane < beam) {
        const int src = w_src[lane], tok = w_tok[lane];
        const float sq = seq_new[src];
        s.seq[cb0 + lane] = sq;
        s.scores[cb0 + lane] = w_key[lane] * sq;
        stop = st_old[src] || tok == stop_id;
        s.stopped[cb0 + lane] = stop;
        s.next_tok[cb0 + lane] = tok;
        s.tokens[(cb0 + lane) * T + step] = tok;
    }
    for (int i = lane; i < beam * step; i += 64) {
        const int b = i / step, t = i - b * step;
        s.tokens[(cb0 + b) * T + t] = tok_old[w_src[b] * T + t];
    }
    for (int i = lane; i < beam * (pos_cur + 1); i += 64) {
        const int b = i / (pos_cur + 1), p = i - b * (pos_cur + 1);
        s.anc[(cb0 + b) * ctx + p] = (p < pos_cur) ? anc_old[w_src[b] * ctx + p] : (uint8_t)w_src[b];
    }
    if (s.kv_stat) {     // distinct slots the NEXT step's attention reads at each of its pos_cur + 1 cached positions
        int cnt = 0;
        for (int p = lane; p <= pos_cur; p += 64) {
            unsigned seen = 0;
            for (int b = 0; b < beam; ++b)
                seen |= 1u << ((p < pos_cur) ? anc_old[w_src[b] * ctx + p] : (uint8_t)w_src[b]);
            cnt += __popc(seen);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if (lane == 0) { s.kv_stat[2 * cap] += (unsigned)cnt; s.kv_stat[2 * cap + 1] += (unsigned)(pos_cur + 1); }
    }
    const bool all = __all(stop);
    if (lane == 0) {
        if (all) s.done[cap] = 1;
        else atomicAdd(s.alive_count, 1);
    }
}

int launch_beam_step(hipStream_t st, const BeamState &s, const float *lse, const float *top_val, const int *top_idx,
                     int ncap, int beam, int k, int T, int ctx, int step, int pos_cur, int vocab, int stop_id,
                     const int *cmap) {
    CAPDEC_CHECK(beam * k <= 64, "beam: beam*k must fit one wavefront");
    CAPDEC_CHECK(T <= SEL_T_MAX && ctx <= SEL_CTX_MAX, "beam: entry_length / context too long");
    CAPDEC_CHECK((long long)beam * vocab < 0x7fffffffLL, "beam: beam*vocab overflows int");
    if (ncap <= 0) return 0;
    const size_t lds = (size_t)beam * T * sizeof(int) + (((size_t)beam * ctx + 3) & ~(size_t)3);
    hipLaunchKernelGGL(beam_step_kernel, dim3(ncap), dim3(64), lds, st, s, lse, top_val, top_idx, ncap, beam, k, T, ctx,
                       step, pos_cur, vocab, stop_id, cmap);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// ---- end of generate_beam: scores /= seq_lengths; order = argsort(desc) (reference :110-114)
__global__ __launch_bounds__(64) void beam_finalize_kernel(BeamState s, int ncap, int beam, int T,
                                                           int *__restrict__ ids, int *__restrict__ lens,
                                                           float *__restrict__ scores, int *__restrict__ order) {
    __shared__ float f[SEL_BEAM_MAX];
    __shared__ int rank_of[SEL_BEAM_MAX];
    const int cap = blockIdx.x, lane = threadIdx.x;
    const size_t cb0 = (size_t)cap * beam;
    if (lane < beam) f[lane] = s.scores[cb0 + lane] / s.seq[cb0 + lane];
    __syncthreads();
    if (lane < beam) {
        int rank = 0;
        for (int o = 0; o < beam; ++o)
            if (f[o] > f[lane] || (f[o] == f[lane] && o < lane)) ++rank;
        rank_of[lane] = rank;
        scores[cb0 + rank] = f[lane];
        lens[cb0 + rank] = (int)s.seq[cb0 + lane];
        if (order) order[cb0 + rank] = lane;
    }
    __syncthreads();
    for (int i = lane; i < beam * T; i += 64) {
        const int b = i / T, t = i - b * T;
        ids[(cb0 + rank_of[b]) * T + t] = s.tokens[(cb0 + b) * T + t];
    }
}

int launch_beam_finalize(hipStream_t st, const BeamState &s, int ncap, int beam, int T, int *ids, int *lens,
                         float *scores, int *order) {
    if (ncap <= 0) return 0;
    hipLaunchKernelGGL(beam_finalize_kernel, dim3(ncap), dim3(64), 0, st, s, ncap, beam, T, ids, lens, scores, order);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// ---- finished-caption compaction: the captions that still generate, in ascending order.  One block; every
// 1024-caption slab is ranked with a wavefront ballot + a 16-entry scan of the wavefront totals.
__global__ __launch_bounds__(1024) void compact_alive_kernel(const uint8_t *__restrict__ done, int ncap,
                                                             int *__restrict__ cmap, int *__restrict__ count) {
    __shared__ int wtot[16];
    __shared__ int base;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < ncap; c0 += 1024) {
        const int c = c0 + t;
        const bool alive = c < ncap && !done[c];
        const unsigned long long m = __ballot(alive);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wtot[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wtot[w];
        if (alive) cmap[off + before] = c;
        __syncthreads();
        if (t == 0) {
            int tot = 0;
            for (int w = 0; w < 16; ++w) tot += wtot[w];
            base += tot;
        }
        __syncthreads();
    }
    if (t == 0) *count = base;
}

int launch_compact_alive(hipStream_t st, const uint8_t *done, int ncap, int *cmap, int *count) {
    if (ncap <= 0) return 0;
    hipLaunchKernelGGL(compact_alive_kernel, dim3(1), dim3(1024), 0, st, done, ncap, cmap, count);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

// ---- token cross-entropy of a logits matrix (the loss of the train step's forward: reference train.py:349 /
// GPT2LMHeadModel's `labels=`): nll[row] = logsumexp(logits[row]) - logits[row][label[row]], one wavefront per row
// (max, then sum of exp(x - max): the order torch's log_softmax uses), rows whose label == ignore_index contribute
// nothing; the mean over the counted rows is taken by ONE block in a fixed order (deterministic).
__global__ __launch_bounds__(256) void token_nll_kernel(const float *__restrict__ logits, int ld, const int *__restrict__ labels,
                                                        int rows, int V, int ignore_index, float *__restrict__ nll) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lab = labels[row];
    if (lab == ignore_index) { if (lane == 0) nll[row] = -1.f; return; }     // -1 = not counted
    // a label outside [0, V) that is not ignore_index is an ERROR in torch's cross_entropy (and in the reference's
    // train.py:350 / GPT2LMHeadModel labels path): a wrong tokenizer or vocabulary must not yield a plausible mean over
    // fewer rows -- the row becomes NaN, so the loss is NaN (capdec.h)
    if (lab < 0 || lab >= V) { if (lane == 0) nll[row] = nanf(""); return; }
    const float *x = logits + (size_t)row * ld;
    float mx = -INFINITY;
    for (int c = lane; c < V; c += 64) mx = fmaxf(mx, x[c]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int c = lane; c < V; c += 64) se += expf(x[c] - mx);
    se = wave_sum(se);
    if (lane == 0) nll[row] = (mx + logf(se)) - x[lab];
}
__global__ __launch_bounds__(256) void nll_mean_kernel(const float *__restrict__ nll, int rows, float *__restrict__ out) {
    __shared__ double part[256];
    __shared__ int cnt[256];
    double s = 0.0;
    int n = 0;
    for (int r = threadIdx.x; r < rows; r += 256) {
        const float v = nll[r];
        if (v >= 0.f || v != v) { s += (double)v; ++n; }          // (a NaN loss must stay visible)
    }
    part[threadIdx.x] = s;
    cnt[threadIdx.x] = n;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { part[threadIdx.x] += part[threadIdx.x + o]; cnt[threadIdx.x] += cnt[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = cnt[0] > 0 ? (float)(part[0] / cnt[0]) : nanf("");    // torch: mean over nothing = nan
}
int launch_cross_entropy_mean(hipStream_t st, const float *logits, int ld, const int *labels, int rows, int V,
                              int ignore_index, float *nll_ws, float *out) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(token_nll_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, logits, ld, labels, rows, V, ignore_index, nll_ws);
    hipLaunchKernelGGL(nll_mean_kernel, dim3(1), dim3(256), 0, st, nll_ws, rows, out);
    CAPDEC_HIP(hipGetLastError());
    return 0;
}

}  // namespace capdec
