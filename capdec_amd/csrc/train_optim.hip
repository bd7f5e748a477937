// Train step, optimizer side (train.h): the table of trainable tensors with its gradient / AdamW-moment arenas, the update,
// the transposed weight copies the dX products read, and the C entry points around the step (capdec.h: capdec_train_get,
// _reset, _set_scope, _set_dropout, _set_dropout_masks, _get_dropout_masks, _loss).
#include "train.h"

namespace capdec {

// ---- full-model scope (GPT-2 trained too) only
// x[i] *= *scale   (capdec_train_get: a gradient leaves the arena normalised)
__global__ void scale_by_kernel(float *x, size_t n, const float *__restrict__ scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= *scale;
}
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
int build_slots(capdec_ctx *c, TrainState &t) {
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
// the forward weights are [out, in] on the device (weights.hip: upload_transposed); dX needs [in, out]
int prepare_backward_weights(capdec_ctx *c, TrainState &t) {
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

int train_apply_update(capdec_ctx *c, TrainState &t, float lr, float b1, float b2, float eps, float weight_decay) {
    hipStream_t st = c->stream;
    StepScalars *sc = t.cnt.as<StepScalars>();
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, st, sc, lr, b1, b2);
    hipLaunchKernelGGL(adamw_multi_kernel, dim3(t.n_chunks), dim3(256), 0, st, t.slotdev.as<SlotDev>(), t.chunkdev.as<int2>(),
                       t.G.as<float>(), t.Mo.as<float>(), t.Vo.as<float>(), sc, b1, b2, eps, lr * weight_decay);
    CAPDEC_HIP(hipGetLastError());
    t.step += 1;
    for (const Slot &sl : t.slots) drop_planes_of(c, sl.p);      // inference must never see planes packed from old values
    if (t.train_gpt) CAPDEC_TRY(refresh_backward_weights(c, t));
    return 0;
}

}  // namespace capdec

using namespace capdec;

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
