"""Seeded synthetic weights and inputs for the caption hot path.

No GPT-2 / CLIP checkpoints or COCO data exist offline, so every fixture, parity
test and bench run uses weights drawn from a fixed recipe on the *CPU* torch
generator (identical stream here and on the GPU box).  Default HF init
(sigma = 0.02) makes GPT-2 emit one constant token whatever the prefix, which
pins nothing; the "hot" law below gives input-dependent, non-repeating
sequences.

The dict returned by :func:`hot_state_dict` uses exactly the key names and
layouts of a reference checkpoint (``torch.save(model.state_dict())``,
reference train.py:359-371): GPT-2 Conv1D matrices stored ``[in, out]``,
``nn.Linear`` matrices stored ``[out, in]``, ``gpt.lm_head.weight`` tied to
``gpt.transformer.wte.weight``.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from dataclasses import dataclass

import numpy as np
import torch


@dataclass(frozen=True)
class GPT2Dims:
    """GPT-2 geometry (defaults = 'gpt2' small, the checkpoint the reference loads,
    reference gpt2_prefix.py:162)."""
    n_layer: int = 12
    n_head: int = 12
    n_embd: int = 768
    vocab: int = 50257
    n_pos: int = 1024
    ln_eps: float = 1e-5


GPT2_SMALL = GPT2Dims()
#: reduced geometry for fast CPU tests (same d / heads so the same kernels run)
GPT2_TINY = GPT2Dims(n_layer=2, vocab=1531, n_pos=128)


def _randn(gen, *shape, std=1.0, mean=0.0):
    return torch.randn(*shape, generator=gen, dtype=torch.float32) * std + mean


def hot_gpt2_state_dict(seed: int = 42, dims: GPT2Dims = GPT2_SMALL, prefix: str = "gpt.") -> "OrderedDict[str, torch.Tensor]":
    """GPT-2 weights under the hot-init law: LN gamma = 1 + 0.1 N, beta = 0.1 N;
    wte sigma 0.15; wpe 0.05; biases 0.02; every other matrix 0.06."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    d = dims.n_embd
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    t = prefix + "transformer."
    sd[t + "wte.weight"] = _randn(g, dims.vocab, d, std=0.15)
    sd[t + "wpe.weight"] = _randn(g, dims.n_pos, d, std=0.05)
    for i in range(dims.n_layer):
        h = f"{t}h.{i}."
        sd[h + "ln_1.weight"] = _randn(g, d, std=0.1, mean=1.0)
        sd[h + "ln_1.bias"] = _randn(g, d, std=0.1)
        sd[h + "attn.c_attn.weight"] = _randn(g, d, 3 * d, std=0.06)
        sd[h + "attn.c_attn.bias"] = _randn(g, 3 * d, std=0.02)
        sd[h + "attn.c_proj.weight"] = _randn(g, d, d, std=0.06)
        sd[h + "attn.c_proj.bias"] = _randn(g, d, std=0.02)
        sd[h + "ln_2.weight"] = _randn(g, d, std=0.1, mean=1.0)
        sd[h + "ln_2.bias"] = _randn(g, d, std=0.1)
        sd[h + "mlp.c_fc.weight"] = _randn(g, d, 4 * d, std=0.06)
        sd[h + "mlp.c_fc.bias"] = _randn(g, 4 * d, std=0.02)
        sd[h + "mlp.c_proj.weight"] = _randn(g, 4 * d, d, std=0.06)
        sd[h + "mlp.c_proj.bias"] = _randn(g, d, std=0.02)
    sd[t + "ln_f.weight"] = _randn(g, d, std=0.1, mean=1.0)
    sd[t + "ln_f.bias"] = _randn(g, d, std=0.1)
    sd[prefix + "lm_head.weight"] = sd[t + "wte.weight"]  # tied
    return sd


def hot_mlp_mapper_state_dict(seed: int, prefix_dim: int, prefix_length: int, d: int = 768,
                              prefix: str = "clip_project.") -> "OrderedDict[str, torch.Tensor]":
    """MLP mapper (reference gpt2_prefix.py:114-126,167-168): sizes (D, d*P//2, d*P)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    hid, out = (d * prefix_length) // 2, d * prefix_length
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    sd[prefix + "model.0.weight"] = _randn(g, hid, prefix_dim, std=math.sqrt(2.0 / prefix_dim))
    sd[prefix + "model.0.bias"] = _randn(g, hid, std=0.02)
    sd[prefix + "model.2.weight"] = _randn(g, out, hid, std=math.sqrt(2.0 / hid))
    sd[prefix + "model.2.bias"] = _randn(g, out, std=0.02)
    return sd


def hot_transformer_mapper_state_dict(seed: int, prefix_dim: int, prefix_length: int, clip_length: int,
                                      num_layers: int = 8, d: int = 768,
                                      prefix: str = "clip_project.") -> "OrderedDict[str, torch.Tensor]":
    """TransformerMapper (reference transformer_mapper.py:113-127): linear D -> clip_len*d,
    prefix_const [P, d], num_layers pre-LN layers (8 heads, mlp ratio 2, q/kv without bias)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for i in range(num_layers):
        l = f"{prefix}transformer.layers.{i}."
        sd[l + "norm1.weight"] = _randn(g, d, std=0.1, mean=1.0)
        sd[l + "norm1.bias"] = _randn(g, d, std=0.1)
        sd[l + "attn.to_queries.weight"] = _randn(g, d, d, std=math.sqrt(2.0 / d))
        sd[l + "attn.to_keys_values.weight"] = _randn(g, 2 * d, d, std=math.sqrt(2.0 / d))
        sd[l + "attn.project.weight"] = _randn(g, d, d, std=math.sqrt(2.0 / d))
        sd[l + "attn.project.bias"] = _randn(g, d, std=0.02)
        sd[l + "norm2.weight"] = _randn(g, d, std=0.1, mean=1.0)
        sd[l + "norm2.bias"] = _randn(g, d, std=0.1)
        sd[l + "mlp.fc1.weight"] = _randn(g, 2 * d, d, std=math.sqrt(2.0 / d))
        sd[l + "mlp.fc1.bias"] = _randn(g, 2 * d, std=0.02)
        sd[l + "mlp.fc2.weight"] = _randn(g, d, 2 * d, std=math.sqrt(2.0 / (2 * d)))
        sd[l + "mlp.fc2.bias"] = _randn(g, d, std=0.02)
    sd[prefix + "linear.weight"] = _randn(g, clip_length * d, prefix_dim, std=math.sqrt(2.0 / prefix_dim))
    sd[prefix + "linear.bias"] = _randn(g, clip_length * d, std=0.02)
    sd[prefix + "prefix_const"] = _randn(g, prefix_length, d, std=1.0)
    return sd


def hot_state_dict(seed: int = 42, mapping_type: str = "mlp", prefix_dim: int = 512, prefix_length: int = 10,
                   clip_length: int = 10, num_layers: int = 8,
                   dims: GPT2Dims = GPT2_SMALL) -> "OrderedDict[str, torch.Tensor]":
    """Full ClipCaptionModel state dict: mapper keys first, then GPT-2 (order is irrelevant
    to loaders; the two parts use independent generator streams seed+1 / seed)."""
    if mapping_type == "mlp":
        sd = hot_mlp_mapper_state_dict(seed + 1, prefix_dim, prefix_length, dims.n_embd)
    elif mapping_type in ("transformer", "transformer_encoder"):
        sd = hot_transformer_mapper_state_dict(seed + 1, prefix_dim, prefix_length, clip_length, num_layers,
                                               dims.n_embd)
    else:
        raise ValueError(f"unsupported mapping_type {mapping_type!r}")
    sd.update(hot_gpt2_state_dict(seed, dims))
    return sd


def with_stop_bias(sd, stop_id: int = 13, alpha: float = 0.0, prefix: str = "gpt."):
    """A shallow copy of a ClipCaptionModel state dict whose wte row ``stop_id`` (and with it the tied lm_head row) gets
    ``alpha * b / (b . b)`` added, b = ln_f.bias: the stop token's logit (ln_f(h) . wte[stop]) rises by ``alpha`` at EVERY
    step, whatever the hidden state (plus a zero-mean term ~0.36 alpha from the normalised part).  Under the hot-init law
    every vocabulary row is an i.i.d. Gaussian, each of the 50 257 tokens wins a step with probability ~1/V and NO single
    id ends captions early (bench.py's stop profile lists the best candidates: mean length 43 of 67); scaling the stop row
    instead makes its logit follow the sign of one projection of the hidden state, which persists over a caption's steps
    (half the captions stop at once, half never).  A constant offset gives every caption the same per-step chance that
    the stop token beats the best of the others -- caption lengths spread around a mean the offset sets, the shape real
    COCO captions have around their ~11 tokens (reference gpt2_prefix_eval.py:107-109,187-188 end a caption at its stop
    token)."""
    out = OrderedDict(sd)
    w = sd[prefix + "transformer.wte.weight"].clone()
    b = sd[prefix + "transformer.ln_f.bias"].float()
    w[stop_id] += float(alpha) * b / float(b @ b)
    out[prefix + "transformer.wte.weight"] = w
    if prefix + "lm_head.weight" in out:
        out[prefix + "lm_head.weight"] = w
    return out


def synthetic_clip_embeddings(n: int, dim: int = 512, seed: int = 0, normalize: bool = True) -> torch.Tensor:
    """[n, dim] fp32 Gaussian rows, L2-normalised like CLIP embeddings after
    reference predictions_runner.py:222."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(n, dim, generator=g, dtype=torch.float32)
    if normalize:
        x = x / x.norm(2, -1, keepdim=True)
    return x


def checksum(t: torch.Tensor) -> int:
    """crc32 of the raw fp32 bytes: detects RNG drift between torch builds so a fixture is
    never silently compared against different weights."""
    return zlib.crc32(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()) & 0xFFFFFFFF


def state_dict_checksum(sd) -> int:
    c = 0
    for k in sorted(sd):
        c = zlib.crc32(np.ascontiguousarray(sd[k].detach().cpu().numpy()).tobytes(), c)
    return c & 0xFFFFFFFF


# ----------------------------------------------------------------------------------------------
# CLIP ViT-B/32 (reference dependency `clip`, openai/CLIP; call sites embeddings_generator.py:49,86,89,
# predictions_runner.py:158-161,212-220).  Key names are those of an OpenAI CLIP state dict.
@dataclass(frozen=True)
class ClipDims:
    embed_dim: int = 512
    # text tower
    context_length: int = 77
    vocab_size: int = 49408
    text_width: int = 512
    text_heads: int = 8
    text_layers: int = 12
    # vision tower (ViT-B/32)
    image_size: int = 224
    patch: int = 32
    vision_width: int = 768
    vision_heads: int = 12
    vision_layers: int = 12


CLIP_VIT_B32 = ClipDims()
#: reduced depth for fast CPU tests (same widths, so the same kernels run)
CLIP_TINY = ClipDims(text_layers=2, vision_layers=2, vocab_size=49408)


def _hot_resblocks(g, sd, prefix: str, width: int, layers: int):
    for i in range(layers):
        b = f"{prefix}transformer.resblocks.{i}."
        sd[b + "ln_1.weight"] = _randn(g, width, std=0.1, mean=1.0)
        sd[b + "ln_1.bias"] = _randn(g, width, std=0.1)
        sd[b + "attn.in_proj_weight"] = _randn(g, 3 * width, width, std=0.06)
        sd[b + "attn.in_proj_bias"] = _randn(g, 3 * width, std=0.02)
        sd[b + "attn.out_proj.weight"] = _randn(g, width, width, std=0.06)
        sd[b + "attn.out_proj.bias"] = _randn(g, width, std=0.02)
        sd[b + "ln_2.weight"] = _randn(g, width, std=0.1, mean=1.0)
        sd[b + "ln_2.bias"] = _randn(g, width, std=0.1)
        sd[b + "mlp.c_fc.weight"] = _randn(g, 4 * width, width, std=0.06)
        sd[b + "mlp.c_fc.bias"] = _randn(g, 4 * width, std=0.02)
        sd[b + "mlp.c_proj.weight"] = _randn(g, width, 4 * width, std=0.06)
        sd[b + "mlp.c_proj.bias"] = _randn(g, width, std=0.02)


def hot_clip_state_dict(seed: int = 43, dims: ClipDims = CLIP_VIT_B32) -> "OrderedDict[str, torch.Tensor]":
    """Seeded 'hot' CLIP ViT-B/32 weights under OpenAI state-dict names (text + visual towers)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    w = dims.text_width
    sd["token_embedding.weight"] = _randn(g, dims.vocab_size, w, std=0.15)
    sd["positional_embedding"] = _randn(g, dims.context_length, w, std=0.05)
    _hot_resblocks(g, sd, "", w, dims.text_layers)
    sd["ln_final.weight"] = _randn(g, w, std=0.1, mean=1.0)
    sd["ln_final.bias"] = _randn(g, w, std=0.1)
    sd["text_projection"] = _randn(g, w, dims.embed_dim, std=w ** -0.5)
    v = dims.vision_width
    n_tok = (dims.image_size // dims.patch) ** 2 + 1
    sd["visual.conv1.weight"] = _randn(g, v, 3, dims.patch, dims.patch, std=0.02)
    sd["visual.class_embedding"] = _randn(g, v, std=0.15)
    sd["visual.positional_embedding"] = _randn(g, n_tok, v, std=0.05)
    sd["visual.ln_pre.weight"] = _randn(g, v, std=0.1, mean=1.0)
    sd["visual.ln_pre.bias"] = _randn(g, v, std=0.1)
    _hot_resblocks(g, sd, "visual.", v, dims.vision_layers)
    sd["visual.ln_post.weight"] = _randn(g, v, std=0.1, mean=1.0)
    sd["visual.ln_post.bias"] = _randn(g, v, std=0.1)
    sd["visual.proj"] = _randn(g, v, dims.embed_dim, std=v ** -0.5)
    sd["logit_scale"] = torch.tensor(2.6592)
    return sd


# ----------------------------------------------------------------------------------------------
# CLIP ModifiedResNet image tower (RN50x4: the reference's DEFAULT backbone, predictions_runner.py:158,
# embeddings_generator.py:113, train.py:445; `clip.load("RN50x4")`).  OpenAI state-dict names under "visual.".
@dataclass(frozen=True)
class ResNetDims:
    layers: tuple = (4, 6, 10, 6)
    width: int = 80
    image_size: int = 288
    embed_dim: int = 640          # output_dim of the attention pool; heads = width * 32 // 64

    @property
    def feat_dim(self) -> int:    # channels entering the attention pool
        return self.width * 32

    @property
    def heads(self) -> int:
        return self.width * 32 // 64


CLIP_RN50X4 = ResNetDims()
#: small geometry for tests: same block structure (stem, 4 stages with strides 1/2/2/2, attention pool, head_dim 64)
CLIP_RN_TINY = ResNetDims(layers=(1, 2, 1, 1), width=16, image_size=64, embed_dim=128)


def _hot_bn(g, sd, p: str, c: int, gain: float = 1.0):
    sd[p + "weight"] = _randn(g, c, std=0.1 * gain, mean=gain)
    sd[p + "bias"] = _randn(g, c, std=0.1)
    sd[p + "running_mean"] = _randn(g, c, std=0.1)
    sd[p + "running_var"] = (torch.rand(c, generator=g, dtype=torch.float32) * 0.5 + 0.75)
    sd[p + "num_batches_tracked"] = torch.tensor(1000)


def hot_clip_resnet_state_dict(seed: int = 44, dims: ResNetDims = CLIP_RN50X4) -> "OrderedDict[str, torch.Tensor]":
    """Seeded weights of CLIP's ModifiedResNet visual tower under OpenAI state-dict names (He-style conv scales so
    activations keep O(1) magnitude through the 4 stages)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    w = dims.width

    def conv(name, cout, cin, k):
        sd[name] = _randn(g, cout, cin, k, k, std=(2.0 / (cin * k * k)) ** 0.5)

    conv("visual.conv1.weight", w // 2, 3, 3); _hot_bn(g, sd, "visual.bn1.", w // 2)
    conv("visual.conv2.weight", w // 2, w // 2, 3); _hot_bn(g, sd, "visual.bn2.", w // 2)
    conv("visual.conv3.weight", w, w // 2, 3); _hot_bn(g, sd, "visual.bn3.", w)
    inplanes = w
    for li, (planes, blocks) in enumerate(zip((w, 2 * w, 4 * w, 8 * w), dims.layers), start=1):
        for b in range(blocks):
            p = f"visual.layer{li}.{b}."
            stride = 2 if (b == 0 and li > 1) else 1
            conv(p + "conv1.weight", planes, inplanes, 1); _hot_bn(g, sd, p + "bn1.", planes)
            conv(p + "conv2.weight", planes, planes, 3); _hot_bn(g, sd, p + "bn2.", planes)
            # (the branch's last BatchNorm is damped so that the residual stream stays O(1) through 26 blocks)
            conv(p + "conv3.weight", planes * 4, planes, 1); _hot_bn(g, sd, p + "bn3.", planes * 4, gain=0.2)
            if stride > 1 or inplanes != planes * 4:
                conv(p + "downsample.0.weight", planes * 4, inplanes, 1); _hot_bn(g, sd, p + "downsample.1.", planes * 4)
            inplanes = planes * 4
    e = dims.feat_dim
    sp = dims.image_size // 32
    sd["visual.attnpool.positional_embedding"] = _randn(g, sp * sp + 1, e, std=e ** -0.5)
    for n in ("q_proj", "k_proj", "v_proj"):
        sd[f"visual.attnpool.{n}.weight"] = _randn(g, e, e, std=e ** -0.5)
        sd[f"visual.attnpool.{n}.bias"] = _randn(g, e, std=0.02)
    sd["visual.attnpool.c_proj.weight"] = _randn(g, dims.embed_dim, e, std=e ** -0.5)
    sd["visual.attnpool.c_proj.bias"] = _randn(g, dims.embed_dim, std=0.02)
    return sd


def synthetic_clip_tokens(n: int, seed: int = 2, context_length: int = 77, vocab: int = 49408,
                          min_len: int = 8, max_len: int = 20) -> torch.Tensor:
    """int64 [n, 77] rows shaped like clip.tokenize output (reference embeddings_generator.py:80-85):
    SOT 49406, 8-20 random ids in [1, 49405], EOT 49407, zero padding."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = torch.zeros(n, context_length, dtype=torch.int64)
    lens = torch.randint(min_len, max_len + 1, (n,), generator=g)
    body = torch.randint(1, vocab - 2, (n, max_len), generator=g)
    out[:, 0] = vocab - 2
    for r in range(n):
        L = int(lens[r])
        out[r, 1:1 + L] = body[r, :L]
        out[r, 1 + L] = vocab - 1
    return out


def synthetic_images(n: int, seed: int = 4, size: int = 224) -> torch.Tensor:
    """fp32 [n, 3, size, size] ~ N(0,1): stands for already-preprocessed (normalised) pixels."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(n, 3, size, size, generator=g, dtype=torch.float32)


def synthetic_photo(h: int, w: int, seed: int) -> "np.ndarray":
    """uint8 RGB test image [h, w, 3]: smooth colour gradients + a few hard edges + noise (exercises both the
    antialiasing taps and the clipping of the bicubic resampler).  numpy RandomState: stable across versions."""
    import numpy as np
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.empty((h, w, 3), np.float64)
    for c in range(3):
        fx, fy, ph = rs.uniform(0.5, 4.0), rs.uniform(0.5, 4.0), rs.uniform(0, 6.28)
        img[..., c] = 127.5 + 110.0 * np.sin(fx * xx / max(w, 1) * 6.28 + fy * yy / max(h, 1) * 6.28 + ph)
    for _ in range(4):                                   # rectangles with saturated colours (hard edges)
        y0, x0 = rs.randint(0, h), rs.randint(0, w)
        y1, x1 = min(h, y0 + rs.randint(1, max(2, h // 3))), min(w, x0 + rs.randint(1, max(2, w // 3)))
        img[y0:y1, x0:x1] = rs.choice([0.0, 255.0], size=3)
    img += rs.normal(0.0, 12.0, size=img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


#: (H, W) of the preprocessing fixtures: landscape / portrait / square / tiny (upscale) / extreme aspect / odd sizes
PREPROCESS_SIZES = [(480, 640), (640, 480), (224, 224), (37, 53), (1000, 130), (333, 500), (500, 333), (17, 17),
                    (225, 223), (768, 1024)]
