"""Drop-in surface of reference ``gpt2_prefix_eval.py``: ``generate_beam`` (:50-115) and
``generate2`` (:118-198) with the reference signatures, plus batched variants
(``embed`` [N, P, 768]) that the throughput path uses.  The per-token Python loop, the
no-cache re-forward and the per-token host syncs of the reference are replaced by one call
into the KV-cached HIP decode (``capdec_decode_greedy`` / ``capdec_decode_beam``)."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from ._capi import CapdecError
from .gpt2_prefix import ClipCaptionModel

ALT_STOP_ID = 764   # hard-coded second stop id of generate2 (reference gpt2_prefix_eval.py:187)


def _prefix_from(model, tokenizer, tokens, prompt, embed) -> Tuple[torch.Tensor, Optional[List[int]]]:
    """reference :70-74 / :141-151: use ``embed`` if given, else wte(tokens or encode(prompt))."""
    if embed is not None:
        return embed, None
    if tokens is None:
        tokens = torch.tensor(tokenizer.encode(prompt))
        tokens = tokens.unsqueeze(0)
    tokens = tokens.reshape(1, -1)
    return model.gpt.transformer.wte(tokens), [int(t) for t in tokens.reshape(-1).tolist()]


# --------------------------------------------------------------------------- batched id-level API
def decode_greedy_ids(model: ClipCaptionModel, embed: torch.Tensor, stop_token_index: int, entry_length: int = 67,
                      alt_stop_id: int = ALT_STOP_ID) -> Tuple[torch.Tensor, torch.Tensor]:
    """embed [N, P, d] -> ids int32 [N, entry_length] (zero padded), lens int32 [N] (tokens
    emitted INCLUDING the stop token) -- device tensors."""
    return model.engine.decode_greedy(embed, stop_token_index, entry_length, alt_stop_id)


def decode_beam_ids(model: ClipCaptionModel, embed: torch.Tensor, stop_token_index: int, beam_size: int = 5,
                    entry_length: int = 67, temperature: float = 1.0):
    """embed [N, P, d] -> (ids [N, beam, T], lens [N, beam], scores [N, beam], order [N, beam]),
    beams sorted by mean log-prob descending (the order generate_beam returns)."""
    return model.engine.decode_beam(embed, stop_token_index, beam_size, entry_length, temperature)


def generate2_batch(model, tokenizer, embed: torch.Tensor, entry_length: int = 67, stop_token: str = '.') -> List[str]:
    stop = tokenizer.encode(stop_token)[0]
    ids, lens = decode_greedy_ids(model, embed, stop, entry_length)
    ids, lens = ids.cpu().numpy(), lens.cpu().numpy()
    return [tokenizer.decode(list(ids[r, :lens[r]])) for r in range(ids.shape[0])]


def generate_beam_batch(model, tokenizer, embed: torch.Tensor, beam_size: int = 5, entry_length: int = 67,
                        temperature: float = 1., stop_token: str = '.') -> List[List[str]]:
    stop = tokenizer.encode(stop_token)[0]
    ids, lens, _, _ = decode_beam_ids(model, embed, stop, beam_size, entry_length, temperature)
    ids, lens = ids.cpu().numpy(), lens.cpu().numpy()
    return [[tokenizer.decode(ids[r, b, :int(lens[r, b])]) for b in range(ids.shape[1])] for r in range(ids.shape[0])]


# --------------------------------------------------------------------------- reference signatures
def generate_beam(model: ClipCaptionModel, tokenizer, beam_size: int = 5, prompt=None, embed=None,
                  entry_length=67, temperature=1., stop_token: str = '.'):
    """reference gpt2_prefix_eval.py:50-115 -> List[str] of ``beam_size`` texts, best first."""
    model.eval()
    prefix, prompt_ids = _prefix_from(model, tokenizer, None, prompt, embed)
    if prefix.shape[0] != 1:
        raise CapdecError("generate_beam takes one caption ([1, P, d]); use generate_beam_batch for [N, P, d]")
    stop = tokenizer.encode(stop_token)[0]
    ids, lens, _, _ = decode_beam_ids(model, prefix, stop, beam_size, entry_length, temperature)
    ids, lens = ids.cpu().numpy()[0], lens.cpu().numpy()[0]
    out = []
    for b in range(beam_size):
        toks = ids[b, :int(lens[b])]
        if prompt_ids is not None:   # reference :86-87: prompt tokens stay in the output
            # seq_lengths counts generated tokens only (+ the reference slices the concatenated row)
            toks = np.concatenate([np.asarray(prompt_ids, dtype=toks.dtype), ids[b]])[:int(lens[b])]
        out.append(tokenizer.decode(toks))
    return out


def generate2(model, tokenizer, tokens=None, prompt=None, embed=None, entry_count=1, entry_length=67,
              top_p=0.8, temperature=1., stop_token: str = '.'):
    """reference gpt2_prefix_eval.py:118-198 -> str.  ``top_p`` is accepted and has no effect,
    exactly as in the reference: the filter never removes the arg-max (:172), and the next
    token is ``argmax`` (:177); ``temperature`` > 0 does not change an arg-max either."""
    model.eval()
    prefix, prompt_ids = _prefix_from(model, tokenizer, tokens, prompt, embed)
    if prefix.shape[0] != 1:
        raise CapdecError("generate2 takes one caption ([1, P, d]); use generate2_batch for [N, P, d]")
    stop = tokenizer.encode(stop_token)[0]
    ids, lens = decode_greedy_ids(model, prefix, stop, entry_length)
    n = int(lens.cpu()[0])
    out = [int(t) for t in ids.cpu().numpy()[0, :n]]
    if prompt_ids is not None:
        out = prompt_ids + out
    if len(out) == 1:
        # reference :191 does list(tokens.squeeze().cpu().numpy()) -- a 0-d array when the very
        # first token stops -- and raises; keep the error behaviour
        raise TypeError("iteration over a 0-d array")
    return tokenizer.decode(out)
