/*
 * capdec.h -- C ABI of libcapdec_hip.so: the MI355X (gfx950) caption hot path of CapDec.
 *
 *   CLIP embedding -> (normalise / noise) -> mapping network -> GPT-2 KV-cached decode
 *   (greedy | beam) -> token ids
 *
 * The reference (DavidHuji/CapDec) is pure Python: its "plugin API" for this path is a set
 * of Python names, not an FFI.  Each entry point below cites the reference interface it
 * replaces (file:line under the reference tree); capdec_amd/ (ctypes) binds exactly these
 * symbols and re-exposes the reference names (MappingType, MLP, TransformerMapper,
 * ClipCaptionModel, generate2, generate_beam, noise_injection).  INTEGRATION.md shows the
 * stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; capdec_last_error() returns
 *     a thread-local message for the last failure on the calling thread.
 *   - one capdec_ctx per GPU / rank, driven by one host thread at a time.
 *   - "h_" pointers are host memory (read during the call), "d_" pointers are device memory
 *     on the context's GPU; bulk data (embeddings in, token ids out) stays on the device.
 *   - all work is enqueued on the context's HIP stream (own stream, or one adopted with
 *     capdec_set_stream); decode calls synchronise that stream before returning.
 *   - tensors are dense row-major fp32 / int32 unless stated.
 */
#ifndef CAPDEC_H
#define CAPDEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAPDEC_ABI_VERSION 5   /* 5: capdec_set_compact, capdec_decode_step_rows (the workload in which captions stop);
                                  4: train step -- GPT-2's dropouts (capdec_train_set_dropout / _masks), capdec_train_loss, loss == NULL
                                     enqueues without waiting, the scope survives capdec_train_reset;
                                  3: the diverged-beam debug hook left the shipped library (measurement builds only);
                                  2: capdec_profile_get takes the array capacity in *count; batch-invariant mode; decode counters */

typedef struct capdec_ctx capdec_ctx;

/* activation codes for capdec_gemm_f32 (test hook) */
enum { CAPDEC_ACT_NONE = 0, CAPDEC_ACT_TANH = 1, CAPDEC_ACT_RELU = 2, CAPDEC_ACT_GELU_NEW = 3,
       CAPDEC_ACT_QUICK_GELU = 4,
       CAPDEC_ACT_RESID_RELU = 5 /* relu(acc + bias + resid): the tail of a ResNet bottleneck */ };

/* ---- context ------------------------------------------------------------------------ */
int capdec_abi_version(void);
/* hash of the sources this library was compiled from (capdec_amd/build.py:source_hash); the Python host refuses a
 * library whose id differs from the source tree next to it (a stale .so would silently run old kernels) */
const char *capdec_build_id(void);
const char *capdec_last_error(void);
/* replaces `device = CUDA(0); model = model.to(device)` (reference predictions_runner.py:154-155) */
int capdec_create(int device_id, capdec_ctx **out);
void capdec_destroy(capdec_ctx *ctx);
/* enqueue on an external hipStream_t (e.g. torch's current stream).  NULL is a valid handle:
 * HIP's default (null) stream.  capdec_use_own_stream() goes back to the context's private
 * non-blocking stream (the default after capdec_create). */
int capdec_set_stream(capdec_ctx *ctx, void *hip_stream);
int capdec_use_own_stream(capdec_ctx *ctx);
int capdec_synchronize(capdec_ctx *ctx);
/* how the dense projections (GPT-2 / CLIP block stacks, lm_head) run:
 * 3 = "f16x2" (DEFAULT): fp32-accurate -- every fp32 operand is two fp16 planes (a = hi + 2^-11 lo), THREE
 *     v_mfma_f32_32x32x16_f16 per product, fp32 accumulate; error below the rounding noise of an fp32 GEMM, every
 *     parity test (token ids bit-identical to the fp32 reference) passes; GEMM inputs are clamped to +-65504;
 * 1 = "bf16x3": fp32-accurate, three bf16 planes, six v_mfma_f32_32x32x16_bf16 per product (round-1 scheme, A/B);
 * 0 = native fp32 MFMA (v_mfma_f32_32x32x2_f32, exact fma chain);
 * 2 = "bf16" (BASELINE configs[1]): weights and GEMM-input activations rounded to bf16 (RNE) and a bf16 KV cache, ONE
 *     MFMA per product, fp32 accumulate; residual stream, LayerNorm and softmax stay fp32.  Not bit-comparable with
 *     the fp32 reference (teacher-forced tolerance tests, capdec_decode_greedy_forced);
 * 4 = "f16": fp16 GEMM operands, fp32 everything else (KV cache included): the precision class of the reference's
 *     CLIP towers on a GPU (clip.load converts to fp16).
 * In the two 16-bit modes the CLIP towers' attention (sequences of 24 .. 128 positions: the 77-token text tower, the
 * 50-token ViT) also takes fp16 operands for its two products -- q, k, v and the un-normalised softmax weights, fp32
 * accumulate, fp32 softmax; GPT-2's attention keeps fp32 operands (bf16 mode: the bf16-rounded K / V of its cache).
 * The mapper, patch-embedding and projection GEMMs are fp32-accurate in every mode.
 * The environment variable CAPDEC_GEMM_MODE=f16x2|bf16x3|f32|bf16|f16 overrides the default at capdec_create (any
 * other value makes capdec_create fail: a typo must not silently select another precision). */
enum { CAPDEC_GEMM_F32 = 0, CAPDEC_GEMM_BF16X3 = 1, CAPDEC_GEMM_BF16 = 2, CAPDEC_GEMM_F16X2 = 3, CAPDEC_GEMM_F16 = 4 };
int capdec_set_gemm_mode(capdec_ctx *ctx, int mode);
int capdec_get_gemm_mode(capdec_ctx *ctx);
/* Batch-invariant mode (default off; CAPDEC_BATCH_INVARIANT=1 sets it at capdec_create).  By default a few kernels
 * pick a variant from the size of the launch -- split-K for under-filled GEMM grids (the K slices are summed in another
 * order than the unsplit loop), the block-tile geometry of a GEMM (one or two accumulator sets), the fused lm_head's
 * tile height, the number of positions the decode attention keeps in flight -- so the fp32 ROUND-OFF
 * of a row depends on how many other rows share its launch: a caption decoded alone, in a 625-caption shard or in a
 * 5000-caption batch can differ in the last bit of a logit, and on a numerical near-tie in a token.  With this mode on,
 * every row goes through the same summation order whatever the batch (unsplit 128x128 GEMMs, pinned kernel variants):
 * results are bit-identical across batch sizes, chunkings and multi-GPU shardings, at the price of under-filled grids
 * for small batches (and they may differ in the last bit from the default mode's). */
int capdec_set_batch_invariant(capdec_ctx *ctx, int on);
/* cap on bytes the decode KV cache may take (captions are processed in chunks that fit);
 * 0 = default (192 GiB of the 288 GB HBM3E) */
int capdec_set_kv_budget(capdec_ctx *ctx, size_t bytes);

/* raw device memory for hosts that do not bring their own allocator (torch is optional) */
int capdec_malloc(capdec_ctx *ctx, size_t bytes, void **d_ptr);
int capdec_free(capdec_ctx *ctx, void *d_ptr);
int capdec_memcpy_h2d(capdec_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int capdec_memcpy_d2h(capdec_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);

/* ---- weights -------------------------------------------------------------------------
 * replaces `model.load_state_dict(torch.load(ckpt))` (reference predictions_runner.py:461);
 * layouts are the checkpoint's own: GPT-2 Conv1D matrices [in, out], nn.Linear [out, in]. */
typedef struct capdec_gpt2_layer {
    const float *ln_1_w, *ln_1_b;            /* [d] */
    const float *c_attn_w, *c_attn_b;        /* [d, 3d], [3d] */
    const float *c_proj_w, *c_proj_b;        /* [d, d], [d] */
    const float *ln_2_w, *ln_2_b;            /* [d] */
    const float *c_fc_w, *c_fc_b;            /* [d, 4d], [4d] */
    const float *mlp_c_proj_w, *mlp_c_proj_b;/* [4d, d], [d] */
} capdec_gpt2_layer;

typedef struct capdec_gpt2_weights {
    int n_layer, n_head, n_embd, vocab, n_pos;
    float ln_eps;
    const float *wte;                        /* [vocab, d]  (lm_head is tied to it) */
    const float *wpe;                        /* [n_pos, d] */
    const capdec_gpt2_layer *layers;         /* [n_layer] */
    const float *ln_f_w, *ln_f_b;            /* [d] */
} capdec_gpt2_weights;

/* transformers.GPT2LMHeadModel as owned by ClipCaptionModel.gpt (reference gpt2_prefix.py:162) */
int capdec_load_gpt2(capdec_ctx *ctx, const capdec_gpt2_weights *h_w);

/* MLP mapper: reference gpt2_prefix.py:114-126, sizes (D, d*P/2, d*P) at :167-168 */
int capdec_load_mapper_mlp(capdec_ctx *ctx, int prefix_dim, int prefix_length, int hidden,
                           const float *h_w1, const float *h_b1,   /* [hidden, D], [hidden] */
                           const float *h_w2, const float *h_b2);  /* [d*P, hidden], [d*P] */

typedef struct capdec_tmapper_layer {
    const float *norm1_w, *norm1_b;          /* [d] */
    const float *to_queries_w;               /* [d, d]   (no bias) */
    const float *to_keys_values_w;           /* [2d, d]  (no bias) */
    const float *project_w, *project_b;      /* [d, d], [d] */
    const float *norm2_w, *norm2_b;          /* [d] */
    const float *fc1_w, *fc1_b;              /* [hid, d], [hid] */
    const float *fc2_w, *fc2_b;              /* [d, hid], [d] */
} capdec_tmapper_layer;

typedef struct capdec_tmapper_weights {
    int prefix_dim, prefix_length, clip_length, num_layers, num_heads, d, mlp_hidden;
    const float *linear_w, *linear_b;        /* [clip_length*d, D], [clip_length*d] */
    const float *prefix_const;               /* [P, d] */
    const capdec_tmapper_layer *layers;      /* [num_layers] */
} capdec_tmapper_weights;

/* TransformerMapper: reference transformer_mapper.py:113-127 (8 heads, pre-LN, ReLU MLP) */
int capdec_load_mapper_transformer(capdec_ctx *ctx, const capdec_tmapper_weights *h_w);

/* ---- prefix stage -------------------------------------------------------------------- */
/* `prefix / prefix.norm(2,-1)` then `+ offset` (reference predictions_runner.py:221-224);
 * d_offset [D] may be NULL; normalize=0 mirrors --dont_normalize_prefix.  In place if out==x. */
int capdec_normalize_prefix(capdec_ctx *ctx, const float *d_x, int n, int dim, int normalize,
                            const float *d_offset, float *d_out);

/* noise_injection (reference train.py:27-39): variance==0 -> copy of x (NOT normalised).
 * d_noise [n,dim]: unit-variance draw standing for torch.randn (NULL -> on-device Philox
 * stream seeded by `seed`).  uniform!=0 -> get_uniform_ball_noise (train.py:18-24) using
 * d_noise as the Gaussian direction and d_u [n] as the torch.rand draw (NULL -> Philox). */
int capdec_noise_inject(capdec_ctx *ctx, const float *d_x, int n, int dim, float variance,
                        const float *d_offset, int uniform, int dont_norm, uint64_t seed,
                        const float *d_noise, const float *d_u, float *d_out);

/* `model.clip_project(prefix)` (reference predictions_runner.py:228; gpt2_prefix.py:145-147):
 * d_x [n, D] -> d_out [n, P, d] with whichever mapper is loaded. */
int capdec_mapper_forward(capdec_ctx *ctx, const float *d_x, int n, float *d_out);

/* ---- CLIP ViT-B/32 towers --------------------------------------------------------------
 * replaces `clip.load("ViT-B/32", device, jit=False)` + `encode_text` / `encode_image`
 * (reference embeddings_generator.py:49,86,89; predictions_runner.py:161,218,220).  Pointers are
 * the tensors of an OpenAI CLIP state dict, in its own layouts (nn.Linear / in_proj [out,in];
 * text_projection / visual.proj [width, embed_dim]).  head_dim must be 64 (ViT-B/32: 512/8, 768/12). */
typedef struct capdec_clip_block {
    const float *ln_1_w, *ln_1_b;            /* [w] */
    const float *in_proj_w, *in_proj_b;      /* [3w, w], [3w] */
    const float *out_proj_w, *out_proj_b;    /* [w, w], [w] */
    const float *ln_2_w, *ln_2_b;            /* [w] */
    const float *c_fc_w, *c_fc_b;            /* [4w, w], [4w] */
    const float *c_proj_w, *c_proj_b;        /* [w, 4w], [w] */
} capdec_clip_block;

typedef struct capdec_clip_text_weights {
    int context_length, vocab, width, heads, layers, embed_dim;
    const float *token_embedding;            /* [vocab, w] */
    const float *positional_embedding;       /* [context_length, w] */
    const capdec_clip_block *blocks;         /* [layers] */
    const float *ln_final_w, *ln_final_b;    /* [w] */
    const float *text_projection;            /* [w, embed_dim] */
} capdec_clip_text_weights;

typedef struct capdec_clip_vision_weights {
    int image_size, patch, width, heads, layers, embed_dim;
    const float *conv1_w;                    /* [w, 3, patch, patch] (no bias) */
    const float *class_embedding;            /* [w] */
    const float *positional_embedding;       /* [(image_size/patch)^2 + 1, w] */
    const float *ln_pre_w, *ln_pre_b;        /* [w] */
    const capdec_clip_block *blocks;         /* [layers] */
    const float *ln_post_w, *ln_post_b;      /* [w] */
    const float *proj;                       /* [w, embed_dim] */
} capdec_clip_vision_weights;

int capdec_load_clip_text(capdec_ctx *ctx, const capdec_clip_text_weights *h_w);
int capdec_load_clip_vision(capdec_ctx *ctx, const capdec_clip_vision_weights *h_w);

/* CLIP ModifiedResNet visual tower (`clip.load("RN50x4")`: the reference's DEFAULT image backbone,
 * predictions_runner.py:158,220; embeddings_generator.py:89,113; train.py:445).  One entry per convolution with the
 * BatchNorm that follows it (inference statistics; folded into the convolution at load time).  Host fp32 pointers in
 * the OpenAI state-dict layouts: w [cout, cin, k, k] (k = 1, or 3 with padding 1), bn_* [cout]. */
typedef struct capdec_conv_bn {
    const float *w, *bn_w, *bn_b, *bn_mean, *bn_var;   /* w == NULL: this convolution does not exist (no downsample) */
    int cin, cout, k;
} capdec_conv_bn;
typedef struct capdec_clip_resnet_weights {
    int image_size;        /* 288 for RN50x4 */
    int width;             /* 80 for RN50x4: stage planes width * {1,2,4,8}, attention-pool input 32 * width channels */
    int embed_dim;         /* 640 */
    int layers[4];         /* bottleneck blocks per stage: {4, 6, 10, 6} */
    const capdec_conv_bn *stem;     /* [3]: visual.conv1/bn1 (stride 2), conv2/bn2, conv3/bn3; then AvgPool2d(2) */
    const capdec_conv_bn *blocks;   /* [4 * sum(layers)]: per bottleneck conv1/bn1, conv2/bn2, conv3/bn3, downsample.0/.1;
                                       the first block of stages 2..4 has stride 2 (AvgPool2d(2) after conv2 and in front
                                       of the downsample convolution) */
    const float *positional_embedding;                 /* visual.attnpool.positional_embedding [(S/32)^2 + 1, 32 width] */
    const float *q_w, *q_b, *k_w, *k_b, *v_w, *v_b;    /* attnpool.{q,k,v}_proj: [32w, 32w], [32w] */
    const float *c_w, *c_b;                            /* attnpool.c_proj: [embed_dim, 32w], [embed_dim] */
} capdec_clip_resnet_weights;
/* after this call capdec_clip_encode_image takes [n, 3, image_size, image_size] pixels through the ResNet tower
 * (a ViT tower loaded earlier is replaced, and vice versa).
 * Arithmetic: in the f16x2 / f16 / bf16 modes every activation between the pixels and the attention pool -- the residual
 * stream included -- exists only as a packed GEMM operand of the mode (f16x2: two fp16 planes, 22 significand bits,
 * |x| <= 65504; f16 / bf16: one 16-bit plane, what the reference's fp16 GPU model does), 3x3 convolutions are implicit
 * GEMMs over it; the bf16x3 / f32 modes keep fp32 activations (im2col + GEMM).  Limits: width even with 32 * width a
 * multiple of 64, image_size a multiple of 32 with at most 256 attention-pool tokens (image_size <= 480). */
int capdec_load_clip_resnet(capdec_ctx *ctx, const capdec_clip_resnet_weights *h_weights);
/* `clip_model.encode_text(clip.tokenize(caption))`: d_tokens int32 [n, context_length] (SOT ... EOT,
 * zero padded; the EOT row is found as argmax of the ids) -> d_out [n, embed_dim], NOT normalised */
int capdec_clip_encode_text(capdec_ctx *ctx, const int32_t *d_tokens, int n, float *d_out);
/* `clip_model.encode_image(preprocess(image))`: d_pixels fp32 [n, 3, S, S] already resized /
 * normalised -> d_out [n, embed_dim], NOT normalised */
int capdec_clip_encode_image(capdec_ctx *ctx, const float *d_pixels, int n, float *d_out);

/* ---- GPT-2 --------------------------------------------------------------------------- */
/* `model.gpt(inputs_embeds=x).logits` (reference gpt2_prefix_eval.py:76-77,163-164).
 * d_embeds [n, L, d].  all_positions!=0 -> d_logits [n, L, vocab] (what the reference
 * materialises); 0 -> d_logits [n, vocab], last position only (what it uses). */
int capdec_gpt2_logits(capdec_ctx *ctx, const float *d_embeds, int n, int L, int all_positions,
                       float *d_logits);
/* ---- the train step (reference train.py:344-354) ------------------------------------------------------------------
 * One iteration: d_prefix [batch, D] is the embedding batch AFTER noise_injection (train.py:347; capdec_noise_inject),
 * d_tokens [batch, length] int32 right-padded with 0 as train.ClipCocoDataset pads (:52-63; under the causal mask the
 * padding mask of :348 changes nothing a real position sees).  Computes logits[:, P-1:-1], the loss of :349
 * (cross_entropy with ignore_index = 0: padding AND real tokens with id 0 are skipped; mean over the rest), its gradient
 * with respect to every tensor of the scope (:350), and -- with apply_update != 0 -- one update of transformers-4.24
 * AdamW (:351; betas / eps / weight_decay as passed: the reference's AdamW(params, lr) means 0.9, 0.999, 1e-6, 0.0; bias
 * correction on; `lr` is the scheduler's current value, :352) on the device-resident weights, which the inference entry
 * points then use.  loss != NULL: *loss (host) receives the loss and the call waits for the device; loss == NULL: the
 * call only enqueues (capdec_train_loss reads the loss, and the running sum, later).  An id outside [0, vocab) -- an
 * IndexError in the reference -- makes the loss NaN and leaves every weight and the optimizer untouched.
 * The optimizer state lives in the context; loading a mapper or GPT-2 again, or capdec_train_reset, drops it (the scope
 * and the dropout setting stay).
 *
 * Scope (capdec_train_set_scope): 0 (default) = the mapper, GPT-2 frozen and in eval mode (--only_prefix:
 * ClipCaptionPrefix, train.py:279-287; a deterministic step); 1 = GPT-2 as well -- the reference's DEFAULT run
 * (train.py:326: AdamW(model.parameters()) of a ClipCaptionModel).  Changing the scope starts a fresh optimizer.
 *
 * capdec_train_get copies a tensor of the scope (kind 0) or its gradient from the last step (kind 1, = what
 * loss.backward() leaves in .grad) to d_out (device).  MLP mapper: which = 0 model.0.weight [hidden, D], 1 model.0.bias,
 * 2 model.2.weight [P * d, hidden], 3 model.2.bias.  TransformerMapper: linear.weight, linear.bias, prefix_const, then per
 * layer norm1.weight, norm1.bias, attn.to_queries.weight, attn.to_keys_values.weight, attn.project.weight, .bias,
 * norm2.weight, .bias, mlp.fc1.weight, .bias, mlp.fc2.weight, .bias.  Scope 1 continues with wte (the tied lm_head), wpe,
 * per layer ln_1.weight, ln_1.bias, attn.c_attn.weight, .bias, attn.c_proj.weight, .bias, ln_2.weight, .bias,
 * mlp.c_fc.weight, .bias, mlp.c_proj.weight, .bias, then ln_f.weight, ln_f.bias; Conv1D weights (and their gradients)
 * are returned in the checkpoint's [in, out] layout. */
int capdec_train_step(capdec_ctx *ctx, const float *d_prefix, const int32_t *d_tokens, int batch, int length, float lr,
                      float beta1, float beta2, float eps, float weight_decay, int apply_update, float *loss);
int capdec_train_get(capdec_ctx *ctx, int kind, int which, float *d_out, size_t n);
int capdec_train_reset(capdec_ctx *ctx);
int capdec_train_set_scope(capdec_ctx *ctx, int train_gpt);
/* The loss of the last train step (*last), the sum of the losses and the number of steps since the last reset (the
 * `accumulated_loss` of train.py:356 without a device round trip per step); reset != 0 clears sum and count.  Waits for
 * the device.  Any pointer may be NULL. */
int capdec_train_loss(capdec_ctx *ctx, float *last, double *sum, long long *steps, int reset);
/* GPT-2's dropouts in scope 1 (GPT2Model in train() mode, train.py:321: `drop` after inputs_embeds + position_embeds;
 * per block attn_dropout on the softmax weights, resid_dropout after attn.c_proj, the MLP's dropout after mlp.c_proj;
 * transformers' default embd_pdrop = attn_pdrop = resid_pdrop = 0.1).  p = 0 (default) computes the step of a model whose
 * GPT2Config has them at 0.  Keep-masks come from a Philox4x32-10 stream keyed by `seed` (counter: element group, number
 * of mask streams drawn since the last capdec_train_reset / capdec_train_set_dropout), survivors scaled by 1 / (1 - p)
 * like torch.  Scope 0 never applies dropout (GPT-2 stays in eval mode there).
 * capdec_train_set_dropout_masks injects the keep-masks of the NEXT train step (consumed once; parity tests): one byte per
 * element, 1 = keep, every site of the step concatenated in call order -- embd [B, S, d], then per block attn
 * [B, heads, S, S], resid [B, S, d], mlp [B, S, d], with S = prefix_length + length; n must equal that total.
 * capdec_train_get_dropout_masks copies the mask stream the last step used (same layout) to d_out. */
int capdec_train_set_dropout(capdec_ctx *ctx, float p, uint64_t seed);
int capdec_train_set_dropout_masks(capdec_ctx *ctx, const uint8_t *d_masks, size_t n);
int capdec_train_get_dropout_masks(capdec_ctx *ctx, uint8_t *d_out, size_t n);

/* The loss of the train step's forward (reference train.py:349 `nnf.cross_entropy(logits, tokens, ignore_index=0)`,
 * and GPT2LMHeadModel's shifted `labels=` loss used by gpt2_prefix.py:154): mean over the rows whose label differs
 * from ignore_index of logsumexp(d_logits[row, 0..vocab)) - d_logits[row, label].  d_logits: device fp32 [rows, ld],
 * d_labels: device int32 [rows], d_loss: device fp32 [1] (NaN when no row counts, like torch).  A label outside
 * [0, vocab) other than ignore_index -- an error in torch -- makes the loss NaN instead of being skipped silently. */
int capdec_cross_entropy(capdec_ctx *ctx, const float *d_logits, int ld, const int32_t *d_labels, int rows, int vocab,
                         int ignore_index, float *d_loss);
/* `model.gpt.transformer.wte(ids)` (reference gpt2_prefix_eval.py:105,181): d_out [n, d] */
int capdec_wte_lookup(capdec_ctx *ctx, const int32_t *d_ids, int n, float *d_out);

/* ---- decode -------------------------------------------------------------------------- */
/* generate2, batched (reference gpt2_prefix_eval.py:118-198; top-p filter + argmax == argmax).
 * d_prefix [n, P, d] -> d_ids [n, entry_length] (zero padded), d_lens [n] = number of tokens
 * INCLUDING the stop token.  A row stops at stop_id or alt_stop_id (764 in the reference,
 * :187; pass -1 to disable). */
int capdec_decode_greedy(capdec_ctx *ctx, const float *d_prefix, int n, int P, int stop_id,
                         int alt_stop_id, int entry_length, int32_t *d_ids, int32_t *d_lens);

/* Teacher-forced greedy decode (evaluation / test hook of the reduced-precision modes, where free-running sequences
 * of two bf16 pipelines diverge by construction): the token fed at step i is d_forced[r, i] (int32 [n, entry_length],
 * e.g. the ids an fp32 run produced) instead of the arg-max; d_ids [n, entry_length] receives the arg-max of every
 * step, d_stats (may be NULL) fp32 [n, entry_length, 3] = (top-1 logit, top-2 logit, logsumexp over the vocabulary) of
 * every step.  Same kernels and KV cache as capdec_decode_greedy; nothing stops early. */
int capdec_decode_greedy_forced(capdec_ctx *ctx, const float *d_prefix, int n, int P, int entry_length,
                                const int32_t *d_forced, int32_t *d_ids, float *d_stats);

/* Limits of the decode entry points (the reference has none, gpt2_prefix_eval.py:50-51,118-129; each is checked and
 * reported through capdec_last_error): beam size 1..8; head_dim = 64 (GPT-2 / CLIP ViT-B/32); context prefix_length +
 * entry_length - 1 <= n_positions (<= 1024, GPT-2's own limit; beyond ~400 positions the attention's per-block tables need
 * most of a CU's LDS and run at lower occupancy); n_embd a multiple of 32, <= 1024.  In the default
 * f16x2 GEMM mode GEMM inputs are clamped to +-65504 (fp16's range; LayerNorm / attention / GELU outputs and weights
 * are orders of magnitude below it). */

/* generate_beam, batched (reference gpt2_prefix_eval.py:50-115).  d_prefix [n, P, d] ->
 * d_ids [n, beam, entry_length], d_lens [n, beam] (= int(seq_lengths)), d_scores [n, beam]
 * (= scores / seq_lengths, :110).  Beams are sorted by score descending (:113-114), so
 * d_ids[i][0] is what `generate_beam(...)[0]` decodes (predictions_runner.py:230).
 * d_order [n, beam] (may be NULL) receives the reference's internal beam index of each row. */
int capdec_decode_beam(capdec_ctx *ctx, const float *d_prefix, int n, int P, int beam, int stop_id,
                       int entry_length, float temperature, int32_t *d_ids, int32_t *d_lens,
                       float *d_scores, int32_t *d_order);
/* Image preprocessing in front of capdec_clip_encode_image: the `preprocess` transform clip.load returns
 * (reference predictions_runner.py:212, embeddings_generator.py:72) = Resize(n_px, BICUBIC) -> CenterCrop(n_px) ->
 * ToTensor -> Normalize(mean, std); stretch != 0 = clip_transform_full (predictions_runner.py:116-122): Resize((n_px,
 * n_px)), no crop.  Bit-identical to PIL's 8-bit bicubic resampler + torch fp32 ToTensor / Normalize.
 * d_rgb: device buffer of concatenated uint8 HWC RGB images; offsets / heights / widths: HOST arrays [n] (byte offset
 * of image i in d_rgb and its size); mean / stdv: HOST float[3]; d_out: device fp32 [n, 3, n_px, n_px]. */
int capdec_preprocess_images(capdec_ctx *ctx, const uint8_t *d_rgb, const int64_t *offsets, const int32_t *heights,
                             const int32_t *widths, int n, int n_px, int stretch, const float *mean,
                             const float *stdv, float *d_out);
/* what the last decode call did: decode steps run (<= entry_length: the loop ends when every caption has stopped),
 * how many times finished captions were compacted out of the batch, and the activation rows pushed through the
 * GPT-2 body after the prefill (n * beam * (steps - 1) without early stopping).  CAPDEC_COMPACT=0 disables compaction. */
int capdec_decode_stats(capdec_ctx *ctx, int *steps, int *compactions, long long *row_steps);
/* finished-caption compaction on / off for the following decode calls (default on; CAPDEC_COMPACT=0 sets it off at
 * capdec_create): with it off every caption stays in the batch until ALL have stopped -- the reference's behaviour per
 * caption is the same either way (reference gpt2_prefix_eval.py:107-109,187-188 stop one caption at a time). */
int capdec_set_compact(capdec_ctx *ctx, int on);
/* rows[i] = activation rows the (i + 1)-th decode step of the last decode call pushed through the GPT-2 body (step 0 is the
 * prefill); *n = how many steps there were (<= entry_length - 1).  At most `cap` entries are written.  With compaction the
 * sequence steps down at the poll points (every 8 steps) as captions finish; without it it is constant. */
int capdec_decode_step_rows(capdec_ctx *ctx, int *rows, int cap, int *n);
/* kv_slots_per_position: over the last capdec_decode_beam call, the mean number of DISTINCT K/V cache slots one
 * (caption, position) of the decode attention read (1.0 = all beams of a caption share their whole history, `beam` =
 * none of it): the decode attention loads each distinct slot once, so its HBM traffic -- and its roofline -- scale with
 * this number (0 when no beam decode ran).  saturated_quads: how many 4-element groups of GEMM operands were clamped to
 * the +-65504 range of the fp16-plane formats since the last call (the counter is reset by the call; NaN is not
 * clamped, it propagates) -- nonzero means the fp32-accuracy claim of the default mode does not hold for this
 * checkpoint / input: switch to CAPDEC_GEMM_BF16X3 or CAPDEC_GEMM_F32.  The saturation counter is kept PER DEVICE (one
 * counter per GPU, shared by every context on it): with several contexts on one GPU the call returns -- and resets --
 * the clamps of all of them.  Either pointer may be NULL; the K/V statistic is accumulated on the device and read back
 * (one small copy + a stream synchronisation) only by this call. */
int capdec_decode_counters(capdec_ctx *ctx, double *kv_slots_per_position, long long *saturated_quads);
/* rows: over the last decode call, how many (row, step) pairs of the fused lm_head went through its exact second pass.
 * Beam search needs the 5 best logits of a row; above 2048 rows the fused kernel keeps only 3 per 128-column vocabulary
 * tile, the merge detects every row for which that can have dropped a candidate that matters, and those rows alone are
 * recomputed with 5 per tile (results are identical to keeping 5 everywhere; CAPDEC_LMHEAD_K3=0 does exactly that).  Compare
 * with row_steps of capdec_decode_stats: a checkpoint whose vocabulary clusters its best candidates in one tile pays for
 * the second pass, and this is the number that shows it. */
int capdec_decode_second_pass_rows(capdec_ctx *ctx, long long *rows);
#ifdef CAPDEC_MEASURE
/* MEASUREMENT BUILDS ONLY (libcapdec_hip_measure.so, compiled with -DCAPDEC_MEASURE; the shipped library does not export
 * it): every beam continues ITSELF (candidates of other parents are ignored), so no two beams of a caption share history
 * after the first step -- the worst-case K/V traffic of the decode attention.  Results are NOT the reference's beam search. */
int capdec_set_debug_diverge(capdec_ctx *ctx, int on);
#endif

/* ---- multi-GPU: caption-batch sharding + ONE gather of the generated ids (RCCL over xGMI) ------------------------
 * The reference has no distributed code: it loops over captions one at a time (predictions_runner.py:194,
 * embeddings_generator.py:58); every caption is independent.  Rank r of R decodes the contiguous block
 * [r * ceil(N/R), (r+1) * ceil(N/R)) of the embedding matrix with replicated weights; the only exchange is an
 * all-gather of int32 token ids / lengths (+ fp32 scores, or fp32 embeddings for the embeddings_generator path),
 * in rank order, so the gathered matrix has the single-GPU layout (and, in batch-invariant mode, the single-GPU
 * bits: by default the fp32 round-off of a row may depend on the size of its launch, see capdec_set_batch_invariant).  One communicator per context; RCCL is
 * dlopen'ed on first use (librccl.so.1), so single-GPU hosts do not need it.  A host without torch distributes the
 * 128-byte id itself (file, socket, MPI, environment ...); capdec_amd/distributed.py can use torch.distributed or
 * this communicator. */
#define CAPDEC_COMM_ID_BYTES 128
/* rank 0: create the communicator id (ncclGetUniqueId) and hand it to every rank */
int capdec_comm_unique_id(char *id /* [CAPDEC_COMM_ID_BYTES] */);
/* every rank, after capdec_create on its own GPU: join (ncclCommInitRank).  nranks == 1 is allowed. */
int capdec_comm_init(capdec_ctx *ctx, int rank, int nranks, const char *id /* [CAPDEC_COMM_ID_BYTES] */);
int capdec_comm_destroy(capdec_ctx *ctx);
/* what RCCL reports for the context's communicator (ncclCommUserRank / ncclCommCount): *rank = 0, *nranks = 1 when the
 * context has none.  A launcher can check that N ranks really joined (bench.py prints it as `rccl_ranks`). */
int capdec_comm_info(capdec_ctx *ctx, int *rank, int *nranks);
/* the shard of rank `rank`: [*lo, *hi) of n_total rows (trailing ranks may get an empty block) */
int capdec_shard_bounds(int n_total, int rank, int nranks, int *lo, int *hi);
/* all-gather of row blocks laid out by capdec_shard_bounds: d_local [n_local, row_elems] 4-byte elements (int32 or
 * fp32) of this rank -> d_global [n_total, row_elems] on every rank (blocks padded to ceil(N/R) rows internally, one
 * ncclAllGather, padding cut off).  Without a communicator (single GPU) it is a device copy and n_local must equal
 * n_total. */
int capdec_gather_rows(capdec_ctx *ctx, const void *d_local, int n_local, int row_elems, int n_total, void *d_global);
/* ids [n_local, T] + lens [n_local] (+ scores [n_local], may be NULL) of this rank's captions -> global arrays */
int capdec_gather_ids(capdec_ctx *ctx, const int32_t *d_ids, const int32_t *d_lens, const float *d_scores, int n_local,
                      int T, int n_total, int32_t *d_ids_global, int32_t *d_lens_global, float *d_scores_global);

/* ---- measurement / test hooks -------------------------------------------------------- */
/* C[M,N] = act(A[M,K] . Bt[N,K]^T + bias[N]) + resid[M,N]; bias / resid may be NULL.
 * The MFMA GEMM every projection above runs on (nn.Linear weights are already [N,K]). */
int capdec_gemm_f32(capdec_ctx *ctx, const float *d_a, int lda, const float *d_bt, int ldb,
                    float *d_c, int ldc, int M, int N, int K, const float *d_bias,
                    const float *d_resid, int ldr, int act);
/* hipEvent timers on the context's stream -- the reference's Timer (predictions_runner.py:125-150) */
int capdec_timer_start(capdec_ctx *ctx);
int capdec_timer_stop_ms(capdec_ctx *ctx, float *ms);   /* records, synchronises, returns elapsed */
/* per-kernel-family accumulated device time since the last reset (hipEvents around the timed launches).
 * on = 0: off; 1: every launch is timed; N > 1: every N-th launch of each family is timed (sampling: the
 * events of the other launches are skipped, so a timed region is barely perturbed).
 * get: on entry *count = capacity of the caller's arrays (24 is always enough; a smaller capacity than the library
 * has families is an error, never an overflow), on return the number of entries filled; ms / launches / flops cover the TIMED launches
 * (flops: algorithmic FLOPs issued, 0 for non-GEMM families), calls = all launches of the family. */
int capdec_profile_enable(capdec_ctx *ctx, int on);
int capdec_profile_get(capdec_ctx *ctx, int *count, const char **names, float *ms, int64_t *launches,
                       double *flops, int64_t *calls);
int capdec_profile_reset(capdec_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* CAPDEC_H */
