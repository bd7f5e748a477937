"""Batched counterpart of the per-caption loop of reference ``embeddings_generator.main``
(:58-101): token rows -> ``clip_model.encode_text`` -> embeddings, sharded over the ranks of a
node and gathered in caption order; plus the train-time continuation the north_star's config 4
names (``noise_injection`` train.py:347 -> ``clip_project`` train.py:253-254)."""
from __future__ import annotations

from typing import Optional

import torch

from . import distributed as cdist
from .clip import ClipModel
from .gpt2_prefix import ClipCaptionModel
from .train import noise_injection


def encode_captions(clip_model: ClipModel, tokens: torch.Tensor, rank: int = 0, world: int = 1,
                    gather: bool = True) -> torch.Tensor:
    """tokens int [N, 77] (already `clip.tokenize`d; rows longer than 77 tokens are the caller's
    retry-on-`caption[:100]` case, reference :80-85) -> text embeddings [N, 512] fp32, NOT normalised
    (reference :86-87), rank order = caption order."""
    lo, hi = cdist.shard_bounds(tokens.shape[0], rank, world)
    emb = clip_model.encode_text(tokens[lo:hi])
    return cdist.gather_rows(emb, tokens.shape[0]) if gather else emb


def encode_images(clip_model: ClipModel, pixels: torch.Tensor, rank: int = 0, world: int = 1,
                  gather: bool = True) -> torch.Tensor:
    """pixels [N, 3, 224, 224] (preprocessed) -> image embeddings [N, 512] (reference :72,89)."""
    lo, hi = cdist.shard_bounds(pixels.shape[0], rank, world)
    emb = clip_model.encode_image(pixels[lo:hi])
    return cdist.gather_rows(emb, pixels.shape[0]) if gather else emb


def text_to_prefix(clip_model: ClipModel, model: ClipCaptionModel, tokens: torch.Tensor, noise_variance: float = 0.0,
                   modality_offset: Optional[torch.Tensor] = None, uniform_noise: bool = False,
                   dont_norm: bool = False, seed: Optional[int] = None, rank: int = 0, world: int = 1) -> torch.Tensor:
    """config 4 of BASELINE.json on this rank's shard: encode_text -> noise_injection -> clip_project
    -> [n_local, P, 768]"""
    emb = encode_captions(clip_model, tokens, rank, world, gather=False)
    emb = noise_injection(emb, noise_variance, modality_offset, uniform_noise, dont_norm, seed=seed)
    return model.clip_project(emb).reshape(emb.shape[0], model.prefix_length, -1)
