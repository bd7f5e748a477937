"""Drop-in surface of reference ``transformer_mapper.py``: ``TransformerMapper`` (:113-127).
The layer stack (Mlp :4-19, MultiHeadAttention :22-51, TransformerLayer :54-73, Transformer
:76-110) runs as HIP kernels inside ``capdec_mapper_forward``; ``TransformerEncoderDecoder``
(:130-145, MappingType.TransformerDecoder) is outside the hot path."""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional

import torch

from ._capi import CapdecError
from .engine import Engine
from .gpt2_prefix import _HipModule


class TransformerMapper(_HipModule):
    def __init__(self, dim_clip: int, dim_embedding: int, prefix_length: int, clip_length: int, num_layers: int = 8,
                 _owner: Optional[_HipModule] = None):
        super().__init__()
        self.dim_clip, self.dim_embedding = dim_clip, dim_embedding
        self.prefix_length, self.clip_length, self.num_layers = prefix_length, clip_length, num_layers
        self._owner = _owner

    def _keys(self):
        keys = ["linear.weight", "linear.bias", "prefix_const"]
        for i in range(self.num_layers):
            p = f"transformer.layers.{i}."
            keys += [p + s for s in ("norm1.weight", "norm1.bias", "attn.to_queries.weight",
                                     "attn.to_keys_values.weight", "attn.project.weight", "attn.project.bias",
                                     "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight",
                                     "mlp.fc2.bias")]
        return keys

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        need = self._keys()
        missing = [k for k in need if k not in sd]
        if missing and strict:
            raise RuntimeError(f"Missing key(s) in state_dict: {missing}")
        for k in need:
            if k in sd:
                self._sd[k] = sd[k].detach().float().cpu()
        if tuple(self._sd["prefix_const"].shape) != (self.prefix_length, self.dim_embedding):
            raise RuntimeError("size mismatch for prefix_const")
        if tuple(self._sd["linear.weight"].shape) != (self.clip_length * self.dim_embedding, self.dim_clip):
            raise RuntimeError("size mismatch for linear.weight")
        self._dirty = True
        return SimpleNamespace(missing_keys=missing, unexpected_keys=[k for k in sd if k not in need])

    def _upload(self, eng: Engine):
        eng.load_mapper_transformer({"clip_project." + k: v for k, v in self._sd.items()})

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        eng = self._owner.engine if self._owner is not None else self.engine
        return eng.mapper_forward(x)   # [B, P, 768]


class TransformerEncoderDecoder:
    def __init__(self, *a, **k):
        raise CapdecError("TransformerEncoderDecoder (MappingType.TransformerDecoder) is outside the accelerated path")
