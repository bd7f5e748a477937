// C ABI, part 2: weights.  capdec_load_* take HOST pointers in the checkpoint's own layouts (Conv1D [in, out], nn.Linear
// [out, in], OpenAI-CLIP state-dict tensors) and upload them once; Conv1D matrices are transposed to k-contiguous
// [out, in] on the device, BatchNorm is folded into the ResNet tower's convolutions.
#include "context.h"

namespace capdec {

// ---------------------------------------------------------------------------- uploads
int upload(std::vector<void *> &owned, const float *h_src, size_t n, float **out) {
    CAPDEC_CHECK(h_src != nullptr, "weights: null host pointer");
    void *p = nullptr;
    CAPDEC_HIP(hipMalloc(&p, n * sizeof(float)));
    owned.push_back(p);
    CAPDEC_HIP(hipMemcpy(p, h_src, n * sizeof(float), hipMemcpyHostToDevice));
    *out = reinterpret_cast<float *>(p);
    return 0;
}
// host [rows, cols] -> device [cols, rows] (Conv1D [in,out] -> k-contiguous [out,in])
int upload_transposed(capdec_ctx *c, std::vector<void *> &owned, const float *h_src, int rows, int cols,
                             float **out) {
    CAPDEC_CHECK(h_src != nullptr, "weights: null host pointer");
    void *tmp = nullptr, *p = nullptr;
    const size_t n = (size_t)rows * cols;
    CAPDEC_HIP(hipMalloc(&tmp, n * sizeof(float)));
    if (hipMalloc(&p, n * sizeof(float)) != hipSuccess) {
        (void)hipFree(tmp);
        set_error("weights: hipMalloc failed");
        return 1;
    }
    owned.push_back(p);
    int rc = 0;
    if (hipMemcpy(tmp, h_src, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = 1;
    if (!rc) rc = launch_transpose(c->stream, (const float *)tmp, (float *)p, rows, cols);
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = 1;
    (void)hipFree(tmp);
    if (rc) {
        set_error("weights: transpose upload failed");
        return 1;
    }
    *out = reinterpret_cast<float *>(p);
    return 0;
}
void free_all(std::vector<void *> &owned) {
    for (void *p : owned) (void)hipFree(p);
    owned.clear();
}


static int upload_blocks(capdec_ctx *c, Tower &t, const capdec_clip_block *blocks) {
    const int d = t.d;
    t.layers.resize(t.n_layer);
    for (int l = 0; l < t.n_layer; ++l) {
        const capdec_clip_block &s = blocks[l];
        Gpt2Layer &w = t.layers[l];
        CAPDEC_TRY(upload(t.owned, s.ln_1_w, d, &w.ln1w));
        CAPDEC_TRY(upload(t.owned, s.ln_1_b, d, &w.ln1b));
        CAPDEC_TRY(upload(t.owned, s.in_proj_w, (size_t)3 * d * d, &w.wqkv));     // already [out, in]
        CAPDEC_TRY(upload(t.owned, s.in_proj_b, 3 * d, &w.bqkv));
        CAPDEC_TRY(upload(t.owned, s.out_proj_w, (size_t)d * d, &w.wproj));
        CAPDEC_TRY(upload(t.owned, s.out_proj_b, d, &w.bproj));
        CAPDEC_TRY(upload(t.owned, s.ln_2_w, d, &w.ln2w));
        CAPDEC_TRY(upload(t.owned, s.ln_2_b, d, &w.ln2b));
        CAPDEC_TRY(upload(t.owned, s.c_fc_w, (size_t)4 * d * d, &w.wfc));
        CAPDEC_TRY(upload(t.owned, s.c_fc_b, 4 * d, &w.bfc));
        CAPDEC_TRY(upload(t.owned, s.c_proj_w, (size_t)4 * d * d, &w.wproj2));
        CAPDEC_TRY(upload(t.owned, s.c_proj_b, d, &w.bproj2));
    }
    return 0;
}

// one chunk of captions through the text tower: tokens [n, ctx] -> out [n, embed]

// fold BatchNorm (inference) into the convolution, reorder to [cout_p][(ky, kx, c_p)], pad, upload
static int upload_conv_bn(ResNet &r, const capdec_conv_bn &s, bool first, ConvW *out) {
    CAPDEC_CHECK(s.w && s.bn_w && s.bn_b && s.bn_mean && s.bn_var, "load_clip_resnet: null convolution tensor");
    CAPDEC_CHECK((s.k == 1 || s.k == 3) && s.cin >= 1 && s.cout >= 1, "load_clip_resnet: 1x1 or 3x3 convolutions only");
    ConvW c;
    c.cin = s.cin; c.cout = s.cout; c.k = s.k;
    c.cin_p = first ? s.cin : pad64(s.cin);
    c.cout_p = pad64(s.cout);
    c.K = first ? 64 : s.k * s.k * c.cin_p;
    CAPDEC_CHECK(!first || (s.cin == 3 && s.k == 3), "load_clip_resnet: the first convolution is 3x3 on 3 channels");
    std::vector<float> w((size_t)c.cout_p * c.K, 0.f), b((size_t)c.cout_p, 0.f);
    for (int o = 0; o < s.cout; ++o) {
        const float scale = s.bn_w[o] / std::sqrt(s.bn_var[o] + 1e-5f);
        b[o] = s.bn_b[o] - s.bn_mean[o] * scale;
        for (int ci = 0; ci < s.cin; ++ci)
            for (int t = 0; t < s.k * s.k; ++t)
                w[(size_t)o * c.K + (size_t)t * c.cin_p + ci] = s.w[((size_t)o * s.cin + ci) * s.k * s.k + t] * scale;
    }
    CAPDEC_TRY(upload(r.owned, w.data(), w.size(), &c.w));
    CAPDEC_TRY(upload(r.owned, b.data(), b.size(), &c.b));
    *out = c;
    return 0;
}

// out[N, Ho, Wo, cout_p] = act(conv(in) folded-BN (+ resid)); k = 3: im2col + GEMM; returns the output spatial size

}  // namespace capdec

using namespace capdec;

extern "C" {

int capdec_load_gpt2(capdec_ctx *c, const capdec_gpt2_weights *w) {
    CAPDEC_CHECK(c && w, "null argument");
    CAPDEC_CHECK(w->n_layer >= 1 && w->n_head >= 1 && w->vocab >= 8 && w->n_pos >= 1, "load_gpt2: bad geometry");
    CAPDEC_CHECK(w->n_embd % w->n_head == 0 && w->n_embd / w->n_head == 64, "load_gpt2: head_dim must be 64");
    CAPDEC_CHECK(w->n_embd % 32 == 0 && w->n_embd <= 1024, "load_gpt2: n_embd must be a multiple of 32, <= 1024");
    CAPDEC_HIP(hipSetDevice(c->device));
    Gpt2 &g = c->gpt;
    free_all(g.owned);
    drop_planes(c);
    train_release(c);
    g = Gpt2();
    g.n_layer = w->n_layer; g.n_head = w->n_head; g.d = w->n_embd; g.vocab = w->vocab; g.n_pos = w->n_pos;
    g.eps = w->ln_eps > 0 ? w->ln_eps : 1e-5f;
    const int d = g.d;
    CAPDEC_TRY(upload(g.owned, w->wte, (size_t)g.vocab * d, &g.wte));
    CAPDEC_TRY(upload(g.owned, w->wpe, (size_t)g.n_pos * d, &g.wpe));
    CAPDEC_TRY(upload(g.owned, w->ln_f_w, d, &g.lnfw));
    CAPDEC_TRY(upload(g.owned, w->ln_f_b, d, &g.lnfb));
    g.layers.resize(g.n_layer);
    for (int l = 0; l < g.n_layer; ++l) {
        const capdec_gpt2_layer &s = w->layers[l];
        Gpt2Layer &t = g.layers[l];
        CAPDEC_TRY(upload(g.owned, s.ln_1_w, d, &t.ln1w));
        CAPDEC_TRY(upload(g.owned, s.ln_1_b, d, &t.ln1b));
        CAPDEC_TRY(upload_transposed(c, g.owned, s.c_attn_w, d, 3 * d, &t.wqkv));
        CAPDEC_TRY(upload(g.owned, s.c_attn_b, 3 * d, &t.bqkv));
        CAPDEC_TRY(upload_transposed(c, g.owned, s.c_proj_w, d, d, &t.wproj));
        CAPDEC_TRY(upload(g.owned, s.c_proj_b, d, &t.bproj));
        CAPDEC_TRY(upload(g.owned, s.ln_2_w, d, &t.ln2w));
        CAPDEC_TRY(upload(g.owned, s.ln_2_b, d, &t.ln2b));
        CAPDEC_TRY(upload_transposed(c, g.owned, s.c_fc_w, d, 4 * d, &t.wfc));
        CAPDEC_TRY(upload(g.owned, s.c_fc_b, 4 * d, &t.bfc));
        CAPDEC_TRY(upload_transposed(c, g.owned, s.mlp_c_proj_w, 4 * d, d, &t.wproj2));
        CAPDEC_TRY(upload(g.owned, s.mlp_c_proj_b, d, &t.bproj2));
    }
    g.loaded = true;
    return 0;
}

int capdec_load_mapper_mlp(capdec_ctx *c, int D, int P, int hidden, const float *w1, const float *b1, const float *w2,
                           const float *b2) {
    CAPDEC_CHECK(c, "null context");
    CAPDEC_CHECK(D % 32 == 0 && hidden % 32 == 0 && P >= 1, "load_mapper_mlp: dims must be multiples of 32");
    CAPDEC_HIP(hipSetDevice(c->device));
    Mapper &m = c->map;
    free_all(m.owned);
    drop_planes(c);
    train_release(c);
    m = Mapper();
    m.D = D; m.P = P; m.hidden = hidden;
    m.d = c->gpt.loaded ? c->gpt.d : 768;
    CAPDEC_TRY(upload(m.owned, w1, (size_t)hidden * D, &m.w1));
    CAPDEC_TRY(upload(m.owned, b1, hidden, &m.b1));
    CAPDEC_TRY(upload(m.owned, w2, (size_t)m.P * m.d * hidden, &m.w2));
    CAPDEC_TRY(upload(m.owned, b2, (size_t)m.P * m.d, &m.b2));
    m.kind = 1;
    return 0;
}

int capdec_load_mapper_transformer(capdec_ctx *c, const capdec_tmapper_weights *w) {
    CAPDEC_CHECK(c && w, "null argument");
    CAPDEC_CHECK(w->prefix_dim % 32 == 0 && w->d % 32 == 0 && w->mlp_hidden % 32 == 0, "load_mapper_transformer: dims must be multiples of 32");
    CAPDEC_CHECK(w->num_heads >= 1 && w->d % w->num_heads == 0, "load_mapper_transformer: bad head count");
    CAPDEC_CHECK(w->clip_length >= 1 && w->prefix_length >= 1 && w->num_layers >= 1, "load_mapper_transformer: bad geometry");
    CAPDEC_HIP(hipSetDevice(c->device));
    Mapper &m = c->map;
    free_all(m.owned);
    drop_planes(c);
    train_release(c);
    m = Mapper();
    m.D = w->prefix_dim; m.P = w->prefix_length; m.clip_len = w->clip_length; m.n_layers = w->num_layers;
    m.heads = w->num_heads; m.d = w->d; m.mlp_hidden = w->mlp_hidden;
    const int d = m.d;
    CAPDEC_TRY(upload(m.owned, w->linear_w, (size_t)m.clip_len * d * m.D, &m.lin_w));
    CAPDEC_TRY(upload(m.owned, w->linear_b, (size_t)m.clip_len * d, &m.lin_b));
    CAPDEC_TRY(upload(m.owned, w->prefix_const, (size_t)m.P * d, &m.prefix_const));
    m.layers.resize(m.n_layers);
    for (int l = 0; l < m.n_layers; ++l) {
        const capdec_tmapper_layer &s = w->layers[l];
        TMapLayer &t = m.layers[l];
        CAPDEC_TRY(upload(m.owned, s.norm1_w, d, &t.n1w));
        CAPDEC_TRY(upload(m.owned, s.norm1_b, d, &t.n1b));
        // fused projection [3d, d] = [to_queries ; to_keys_values] -> rows [q | k | v]
        CAPDEC_CHECK(s.to_queries_w && s.to_keys_values_w, "load_mapper_transformer: null weight");
        void *p = nullptr;
        CAPDEC_HIP(hipMalloc(&p, (size_t)3 * d * d * 4));
        m.owned.push_back(p);
        t.wqkv = (float *)p;
        CAPDEC_HIP(hipMemcpy(t.wqkv, s.to_queries_w, (size_t)d * d * 4, hipMemcpyHostToDevice));
        CAPDEC_HIP(hipMemcpy(t.wqkv + (size_t)d * d, s.to_keys_values_w, (size_t)2 * d * d * 4, hipMemcpyHostToDevice));
        CAPDEC_TRY(upload(m.owned, s.project_w, (size_t)d * d, &t.wproj));
        CAPDEC_TRY(upload(m.owned, s.project_b, d, &t.bproj));
        CAPDEC_TRY(upload(m.owned, s.norm2_w, d, &t.n2w));
        CAPDEC_TRY(upload(m.owned, s.norm2_b, d, &t.n2b));
        CAPDEC_TRY(upload(m.owned, s.fc1_w, (size_t)m.mlp_hidden * d, &t.wfc1));
        CAPDEC_TRY(upload(m.owned, s.fc1_b, m.mlp_hidden, &t.bfc1));
        CAPDEC_TRY(upload(m.owned, s.fc2_w, (size_t)d * m.mlp_hidden, &t.wfc2));
        CAPDEC_TRY(upload(m.owned, s.fc2_b, d, &t.bfc2));
    }
    m.kind = 2;
    return 0;
}

int capdec_load_clip_text(capdec_ctx *c, const capdec_clip_text_weights *w) {
    CAPDEC_CHECK(c && w, "null argument");
    CAPDEC_CHECK(w->width % 32 == 0 && w->heads >= 1 && w->width / w->heads == 64 && w->width % w->heads == 0,
                 "load_clip_text: head_dim must be 64");
    CAPDEC_CHECK(w->context_length >= 1 && w->context_length <= 256 && w->layers >= 1 && w->embed_dim >= 1 && w->vocab >= 2,
                 "load_clip_text: bad geometry");
    CAPDEC_HIP(hipSetDevice(c->device));
    Tower &t = c->clip_text;
    free_all(t.owned);
    drop_planes(c);
    t = Tower();
    t.n_layer = w->layers; t.n_head = w->heads; t.d = w->width; t.embed = w->embed_dim; t.ctx = w->context_length;
    t.vocab = w->vocab;
    CAPDEC_TRY(upload(t.owned, w->token_embedding, (size_t)t.vocab * t.d, &t.tok_emb));
    CAPDEC_TRY(upload(t.owned, w->positional_embedding, (size_t)t.ctx * t.d, &t.pos_emb));
    CAPDEC_TRY(upload(t.owned, w->ln_final_w, t.d, &t.lnf_w));
    CAPDEC_TRY(upload(t.owned, w->ln_final_b, t.d, &t.lnf_b));
    CAPDEC_TRY(upload_transposed(c, t.owned, w->text_projection, t.d, t.embed, &t.proj_t));
    CAPDEC_TRY(upload_blocks(c, t, w->blocks));
    t.loaded = true;
    return 0;
}

int capdec_load_clip_vision(capdec_ctx *c, const capdec_clip_vision_weights *w) {
    CAPDEC_CHECK(c && w, "null argument");
    CAPDEC_CHECK(w->width % 32 == 0 && w->heads >= 1 && w->width % w->heads == 0 && w->width / w->heads == 64,
                 "load_clip_vision: head_dim must be 64");
    CAPDEC_CHECK(w->patch >= 4 && w->patch % 4 == 0 && w->image_size % w->patch == 0 && (3 * w->patch * w->patch) % 32 == 0,
                 "load_clip_vision: bad patch geometry");
    CAPDEC_HIP(hipSetDevice(c->device));
    Tower &t = c->clip_vision;
    free_all(t.owned);
    free_all(c->clip_resnet.owned);                 // one image tower at a time
    c->clip_resnet = ResNet();
    drop_planes(c);
    t = Tower();
    t.n_layer = w->layers; t.n_head = w->heads; t.d = w->width; t.embed = w->embed_dim; t.image = w->image_size;
    t.patch = w->patch;
    const int g = t.image / t.patch;
    t.ntok = g * g + 1;
    CAPDEC_CHECK(t.ntok <= 256, "load_clip_vision: more than 256 tokens per image");
    CAPDEC_TRY(upload(t.owned, w->conv1_w, (size_t)t.d * 3 * t.patch * t.patch, &t.conv_w));
    CAPDEC_TRY(upload(t.owned, w->class_embedding, t.d, &t.cls));
    CAPDEC_TRY(upload(t.owned, w->positional_embedding, (size_t)t.ntok * t.d, &t.pos_emb));
    CAPDEC_TRY(upload(t.owned, w->ln_pre_w, t.d, &t.ln_pre_w));
    CAPDEC_TRY(upload(t.owned, w->ln_pre_b, t.d, &t.ln_pre_b));
    CAPDEC_TRY(upload(t.owned, w->ln_post_w, t.d, &t.lnf_w));
    CAPDEC_TRY(upload(t.owned, w->ln_post_b, t.d, &t.lnf_b));
    CAPDEC_TRY(upload_transposed(c, t.owned, w->proj, t.d, t.embed, &t.proj_t));
    CAPDEC_TRY(upload_blocks(c, t, w->blocks));
    t.loaded = true;
    return 0;
}

int capdec_load_clip_resnet(capdec_ctx *c, const capdec_clip_resnet_weights *w) {
    CAPDEC_CHECK(c && w && w->stem && w->blocks, "load_clip_resnet: null argument");
    CAPDEC_CHECK(w->width >= 2 && w->width % 2 == 0 && (w->width * 32) % 64 == 0 && w->embed_dim >= 1 &&
                     w->image_size >= 64 && w->image_size % 32 == 0,
                 "load_clip_resnet: bad geometry (width even, 32 * width a multiple of 64, image_size a multiple of 32)");
    CAPDEC_CHECK((w->image_size / 32) * (w->image_size / 32) + 1 <= 256, "load_clip_resnet: more than 256 attention-pool tokens");
    CAPDEC_HIP(hipSetDevice(c->device));
    ResNet &r = c->clip_resnet;
    free_all(r.owned);
    drop_planes(c);
    r = ResNet();
    free_all(c->clip_vision.owned);                 // one image tower at a time
    c->clip_vision = Tower();
    r.image = w->image_size; r.width = w->width; r.embed = w->embed_dim; r.feat = w->width * 32; r.heads = r.feat / 64;
    r.sp = w->image_size / 32;
    int nblocks = 0;
    for (int i = 0; i < 4; ++i) {
        CAPDEC_CHECK(w->layers[i] >= 1, "load_clip_resnet: every stage needs at least one block");
        r.layers[i] = w->layers[i];
        nblocks += w->layers[i];
    }
    for (int i = 0; i < 3; ++i) CAPDEC_TRY(upload_conv_bn(r, w->stem[i], i == 0, &r.stem[i]));
    r.blocks.resize((size_t)4 * nblocks);
    for (int i = 0; i < 4 * nblocks; ++i) {
        if (i % 4 == 3 && w->blocks[i].w == nullptr) continue;          // no downsample in this block
        CAPDEC_TRY(upload_conv_bn(r, w->blocks[i], false, &r.blocks[(size_t)i]));
    }
    const size_t C = (size_t)r.feat, T = (size_t)r.sp * r.sp + 1;
    CAPDEC_CHECK(r.feat % 64 == 0 && r.embed % 32 == 0, "load_clip_resnet: feature / embedding widths must be multiples of 64 / 32");
    CAPDEC_TRY(upload(r.owned, w->positional_embedding, T * C, &r.pos));
    CAPDEC_TRY(upload(r.owned, w->q_w, C * C, &r.wq)); CAPDEC_TRY(upload(r.owned, w->q_b, C, &r.bq));
    CAPDEC_TRY(upload(r.owned, w->k_w, C * C, &r.wk)); CAPDEC_TRY(upload(r.owned, w->k_b, C, &r.bk));
    CAPDEC_TRY(upload(r.owned, w->v_w, C * C, &r.wv)); CAPDEC_TRY(upload(r.owned, w->v_b, C, &r.bv));
    CAPDEC_TRY(upload(r.owned, w->c_w, (size_t)r.embed * C, &r.wc)); CAPDEC_TRY(upload(r.owned, w->c_b, (size_t)r.embed, &r.bc));
    r.loaded = true;
    return 0;
}


}  // extern "C"
