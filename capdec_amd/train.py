"""Drop-in surface of the inference-relevant part of reference ``train.py``: ``noise_injection``
(:27-39), ``get_uniform_ball_noise`` (:18-24), ``MappingType`` (:42-44) and the
``ClipCaptionModel`` ctor spelling with ``prefix_size`` (:262).  The optimiser / backward loop
(:317-392) is out of scope."""
from __future__ import annotations

import math
from typing import Optional

import torch

from .engine import get_engine
from .gpt2_prefix import ClipCaptionModel, ClipCaptionPrefix, MappingType  # noqa: F401  (re-exported names)

device = torch.device('cuda:0')   # reference train.py:15

_seed_counter = [0]


def _next_seed() -> int:
    """a fresh Philox key per call, derived from torch's global seed so that
    ``torch.manual_seed`` makes runs repeatable"""
    _seed_counter[0] += 1
    return (torch.initial_seed() * 1000003 + _seed_counter[0]) & 0xFFFFFFFFFFFFFFFF


def get_uniform_ball_noise(input_shape, radius=0.1, *, noise: Optional[torch.Tensor] = None,
                           u: Optional[torch.Tensor] = None, seed: Optional[int] = None):
    """reference train.py:18-24: direction = normalize(randn), length = rand ** (1/dim) * radius.
    The direction comes from the fused noise kernel run on x = 0 (its trailing normalisation
    leaves exactly normalize(g)); ``noise`` / ``u`` inject the two draws for parity tests."""
    eng = get_engine(device.index or 0)
    n, dim = int(input_shape[0]), int(input_shape[1])
    if u is None:
        u = torch.rand(n, device=eng.device)
    u = u.to(eng.device).float()
    sphere = eng.noise_inject(torch.zeros(n, dim, device=eng.device), 1.0, None, uniform=True, dont_norm=True,
                              seed=_next_seed() if seed is None else seed, noise=noise, u=u)
    return sphere * ((u ** (1.0 / dim)) * radius)[:, None]


def noise_injection(x, variance=0.001, modality_offset=None, uniform_noise=False, dont_norm=False, *,
                    noise: Optional[torch.Tensor] = None, u: Optional[torch.Tensor] = None,
                    seed: Optional[int] = None):
    """reference train.py:27-39 on the GPU: normalise -> + N(0, variance) (or uniform ball of
    radius sqrt(variance)) -> + modality_offset -> normalise, one fused kernel.
    ``variance == 0`` returns x unchanged (not normalised), like the reference.
    Keyword-only extras: ``noise`` / ``u`` inject the random draws (parity tests),
    ``seed`` fixes the on-device Philox stream."""
    if variance == 0.0:
        return x
    eng = get_engine(x.device.index or 0 if x.is_cuda else device.index or 0)
    return eng.noise_inject(x, variance, modality_offset, uniform=uniform_noise, dont_norm=dont_norm,
                            seed=_next_seed() if seed is None else seed, noise=noise, u=u)
