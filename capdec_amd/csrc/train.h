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
//
// Translation units (round 6: one 1500-line file until then):
//   train_step.hip    the step itself -- GPT-2 forward with saved activations, loss, backward -- with the kernels only it uses
//                     (GELU, the dropouts, embedding / loss-row maps, cross-entropy); capdec_train_step
//   train_mapper.hip  both mapping networks: forward with saved activations, backward
//   train_ops.hip     what the two share: LayerNorm / attention backward, the block-form attention, transposes, column sums,
//                     the dX / dW products
//   train_optim.hip   the trainable-tensor table, gradient / moment arenas, AdamW, the transposed weight copies, and the
//                     rest of the C entry points (capdec_train_get / _reset / _set_scope / _set_dropout* / _loss)
#pragma once
#include "context.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace capdec {

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
struct SlotDev { float *p; unsigned long long off, n; };
constexpr int ADAMW_CHUNK = 16384;

// ---- block-form attention: geometry shared by the launchers (train_ops.hip) and the callers' fit checks
constexpr int ATTN_BLK_NW = 16;                                     // wavefronts per block (1024 threads: the loops are LDS-latency-bound)
template <int HD> struct AttnBlk {
    static constexpr int LD = HD + 1;
    static size_t fwd_bytes(int S) { return ((size_t)2 * S * LD + ATTN_BLK_NW * (HD + S)) * sizeof(float); }
    static size_t bwd_bytes(int S) {
        return ((size_t)2 * S * LD + (size_t)2 * S * (S + 1) + ATTN_BLK_NW * (2 * HD + S)) * sizeof(float);
    }
};
// the block kernels when a head's tiles fit the CU's LDS (S <= 128), the per-query wavefront kernels otherwise
constexpr size_t ATTN_BLK_LDS_MAX = 150 * 1024;

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

inline dim3 grid1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }
inline int pad32(int n) { return (n + 31) / 32 * 32; }
// K of a weight-gradient product = its row count, zero-padded to what the GEMM in use wants (32; 64 for the f16x2 kernels)
inline int pad_rows(const capdec_ctx *c, int n) { return c->tune.train_f16x2 ? (n + 63) / 64 * 64 : pad32(n); }

// ---- train_ops.hip
int transpose_pad(capdec_ctx *c, const float *src, int rows, int cols, float *dst, int ld);
int gemm_fp32(capdec_ctx *c, const float *A, int lda, const float *Bt, int ldb, float *C, int ldc, int M, int N, int K,
              bool static_weight = false);
int ln_bwd(capdec_ctx *c, const float *x, const float *w, const float *dy, const float *add, float *dx, int rows, int d, float eps,
           float *gw = nullptr, float *gb = nullptr);
int linear_dx(capdec_ctx *c, TrainState &t, const float *dy, const float *W, float *dx, int M, int out, int in);
int linear_dw(capdec_ctx *c, TrainState &t, const float *dy, const float *x, int rows, int out, int in, float *gW, float *gb);
// attention of the step's sequences on fused qkv activations [B S, 3 d]; (hd, causal) = (96, false): the TransformerMapper,
// (64, true): GPT-2.  mask: GPT-2's attn_dropout keep bytes [B, H, S, S] (nullptr: none), inv_keep = 1 / (1 - p).
// The block-per-(sample, head) kernels when a head's tiles fit the CU's LDS (and CAPDEC_TRAIN_ATTN_BLK allows), the
// per-query wavefront kernels otherwise (backward: lse / dsum scratch of the TrainState)
int train_attn_fwd(capdec_ctx *c, const float *qkv, float *out, int B, int S, int heads, int hd, bool causal, float scale,
                   const uint8_t *mask, float inv_keep);
int train_attn_bwd(capdec_ctx *c, TrainState &t, const float *qkv, const float *dout, float *dqkv, int B, int S, int heads, int hd,
                   bool causal, float scale, const uint8_t *mask, float inv_keep);

// ---- train_mapper.hip
int mapper_forward_saved(capdec_ctx *c, TrainState &t, const float *x, int B, float *pe);
int mapper_backward(capdec_ctx *c, TrainState &t, const float *x, const float *dy, int B);

// ---- train_optim.hip
int build_slots(capdec_ctx *c, TrainState &t);
int prepare_backward_weights(capdec_ctx *c, TrainState &t);
// AdamW (transformers 4.24 semantics) on arena x gscale for every slot, then what depends on the new values: cached operand
// planes of the trained tensors are dropped, the transposed GPT-2 copies follow in the full scope
int train_apply_update(capdec_ctx *c, TrainState &t, float lr, float b1, float b2, float eps, float weight_decay);

}  // namespace capdec
