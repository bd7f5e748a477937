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
