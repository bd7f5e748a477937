"""Batched counterpart of the per-image loop of reference ``predictions_runner.make_preds``
(:194-234,300-301): embeddings -> normalise (+ modality offset) -> ``clip_project`` ->
``generate_beam`` / ``generate2`` -> ``{"caption": text.lower(), "image_id": ...}``.
``generate_beam`` / ``generate2`` are re-exported here because the reference imports them
into this module (:13) and BASELINE.json's north_star names them under it."""
from __future__ import annotations

import json
from typing import Dict, List, Optional, Sequence

import torch

from . import distributed as cdist
from .gpt2_prefix import ClipCaptionModel, MappingType  # noqa: F401
from .gpt2_prefix_eval import (decode_beam_ids, decode_greedy_ids, generate2, generate2_batch,  # noqa: F401
                               generate_beam, generate_beam_batch)


class Timer:
    """reference predictions_runner.py:125-150 (``with timer: ...`` around the work of one image, ``print(timer)`` every 99
    images, :214,:233,:253): here the events are the C ABI's hipEvent pair on the engine's stream (capdec_timer_start /
    capdec_timer_stop_ms) and one interval covers one BATCH; ``per_item`` divides by the batch size so that the printed
    mean / std stay "ms per image" like the reference's."""

    def __init__(self, model: Optional[ClipCaptionModel] = None):
        self.sum = 0.0
        self.count = 0
        self.timings: List[float] = []
        self.items = 0
        self._model = model

    def bind(self, model: ClipCaptionModel):
        self._model = model
        return self

    def __enter__(self):
        self._model.engine.timer_start()
        return self

    def __exit__(self, *args):
        interval = self._model.engine.timer_stop_ms()
        self.timings.append(interval)
        self.sum += interval
        self.count += 1

    def add_items(self, n: int):
        self.items += int(n)

    def __str__(self):
        import numpy as np
        if self.count == 0:
            return "mean: nan ms, std: nan ms"
        s = f"mean: {self.sum / self.count:.2f} ms, std: {float(np.std(self.timings)):.2f} ms"
        if self.items:
            s += f" per batch; {self.sum / self.items:.4f} ms per image over {self.items} images"
        return s


def prefix_from_embeddings(model: ClipCaptionModel, embeddings: torch.Tensor, dont_normalize_prefix: bool = False,
                           modality_offset: Optional[torch.Tensor] = None) -> torch.Tensor:
    """reference :221-228 for a batch: ``prefix / prefix.norm(2,-1)``; ``+ offset``;
    ``clip_project(prefix).reshape(N, P, -1)``."""
    eng = model.engine
    x = eng.normalize_prefix(embeddings, normalize=not dont_normalize_prefix, offset=modality_offset)
    return model.clip_project(x).reshape(x.shape[0], model.prefix_length, -1)


def caption_ids(model: ClipCaptionModel, embeddings: torch.Tensor, stop_token_index: int, beam: bool = True,
                beam_size: int = 5, entry_length: int = 67, dont_normalize_prefix: bool = False,
                modality_offset: Optional[torch.Tensor] = None, rank: int = 0, world: int = 1):
    """The whole device-side path for this rank's shard of ``embeddings`` [N, D].
    Returns (ids [n_local, T] int32 of the best caption, lens [n_local]) and, for beam, the
    mean-log-prob score of the best beam."""
    lo, hi = cdist.shard_bounds(embeddings.shape[0], rank, world)
    pe = prefix_from_embeddings(model, embeddings[lo:hi], dont_normalize_prefix, modality_offset)
    if beam:
        ids, lens, scores, _ = decode_beam_ids(model, pe, stop_token_index, beam_size, entry_length)
        return ids[:, 0].contiguous(), lens[:, 0].contiguous(), scores[:, 0].contiguous()
    ids, lens = decode_greedy_ids(model, pe, stop_token_index, entry_length)
    return ids, lens, None


def make_preds(data: Sequence[Dict], embeddings: torch.Tensor, model: ClipCaptionModel, tokenizer,
               out_path: Optional[str] = None, beam: bool = True, entry_length: int = 67,
               dont_normalize_prefix: bool = False, modality_offset: Optional[torch.Tensor] = None,
               rank: int = 0, world: int = 1, timer: Optional[Timer] = None) -> List[Dict]:
    """``data[i]`` = {"image_id": ...}; ``embeddings[i]`` its CLIP embedding.  Writes the
    reference's predictions JSON (``[{"caption": lower-cased text, "image_id": id}]``, :260-261,
    :301) -- the whole list, not only every 99th flush."""
    cdist.check_world(rank, world)
    if len(data) != embeddings.shape[0]:
        raise ValueError(f"make_preds: {len(data)} data entries but {embeddings.shape[0]} embeddings")
    stop = tokenizer.encode('.')[0]
    if timer is not None:           # the reference's `with timer:` (:214-233) around this rank's share of the batch
        with timer.bind(model):
            ids, lens, _ = caption_ids(model, embeddings, stop, beam, 5, entry_length, dont_normalize_prefix,
                                       modality_offset, rank, world)
        lo, hi = cdist.shard_bounds(embeddings.shape[0], rank, world)
        timer.add_items(hi - lo)
        print(timer)                # (:253)
    else:
        ids, lens, _ = caption_ids(model, embeddings, stop, beam, 5, entry_length, dont_normalize_prefix,
                                   modality_offset, rank, world)
    ids, lens, _ = cdist.gather_ids(ids, lens, embeddings.shape[0])
    ids, lens = ids.cpu().numpy(), lens.cpu().numpy()
    new_data = [{"caption": tokenizer.decode(list(ids[i, :int(lens[i])])).lower(), "image_id": d["image_id"]}
                for i, d in enumerate(data)]
    if out_path and rank == 0:
        with open(out_path, 'w') as outfile:
            json.dump(new_data, outfile)
    return new_data


def make_preds_from_images(data: Sequence[Dict], images: Sequence, clip_model, preprocess, model: ClipCaptionModel,
                           tokenizer, out_path: Optional[str] = None, beam: bool = True, is_rn: bool = False,
                           entry_length: int = 67, dont_normalize_prefix: bool = False,
                           modality_offset: Optional[torch.Tensor] = None, rank: int = 0, world: int = 1,
                           image_batch: int = 256) -> List[Dict]:
    """The image half of the reference loop in front of ``make_preds`` (:156-161, :207-220): ``images[i]`` is what
    ``Image.open(filename).convert("RGB")`` gave for ``data[i]`` (a PIL image or a uint8 [H, W, 3] array), or ``None``
    for a file the reference would skip (:207-210: the entry is left out of the output).  ``preprocess`` and
    ``clip_model.encode_image`` (``capdec_amd.clip.load``) run on the device in batches of ``image_batch``;
    ``is_rn`` (the RN50x4 backbone) forces ``beam=True`` exactly as the reference does (:157-159)."""
    if len(data) != len(images):
        raise ValueError(f"make_preds_from_images: {len(data)} data entries but {len(images)} images")
    if is_rn:
        beam = True
    keep = [i for i, im in enumerate(images) if im is not None]
    if not keep:                       # nothing to caption: the reference writes an empty list
        if out_path and rank == 0:
            with open(out_path, 'w') as outfile:
                json.dump([], outfile)
        return []
    # the image tower is the expensive half of this path (RN50x4: ~10x a caption's decode), so it is sharded like the
    # decode: rank r preprocesses / encodes / decodes only ITS block of the kept images; the ids are gathered at the end
    cdist.check_world(rank, world)
    lo, hi = cdist.shard_bounds(len(keep), rank, world)
    mine = keep[lo:hi]
    feats = []
    for b0 in range(0, len(mine), max(1, image_batch)):
        batch = [images[i] for i in mine[b0:b0 + max(1, image_batch)]]
        feats.append(clip_model.encode_image(preprocess.batch(batch)).float())
    stop = tokenizer.encode('.')[0]
    T = entry_length
    if feats:
        ids, lens, _ = caption_ids(model, torch.cat(feats), stop, beam, 5, T, dont_normalize_prefix, modality_offset)
    else:                              # an empty shard (more ranks than images) still takes part in the gather
        dev = next(model.parameters()).device
        ids, lens = torch.zeros(0, T, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.int32, device=dev)
    ids, lens, _ = cdist.gather_ids(ids, lens, len(keep))
    ids, lens = ids.cpu().numpy(), lens.cpu().numpy()
    new_data = [{"caption": tokenizer.decode(list(ids[j, :int(lens[j])])).lower(), "image_id": data[i]["image_id"]}
                for j, i in enumerate(keep)]
    if out_path and rank == 0:
        with open(out_path, 'w') as outfile:
            json.dump(new_data, outfile)
    return new_data


def make_preds_from_captions(data: Sequence[Dict], clip_model, model: ClipCaptionModel, tokenizer, tokenize=None,
                             out_path: Optional[str] = None, beam: bool = True, entry_length: int = 67,
                             dont_normalize_prefix: bool = False, modality_offset: Optional[torch.Tensor] = None,
                             rank: int = 0, world: int = 1, text_batch: int = 2048) -> List[Dict]:
    """The TEXT-input branch of the reference loop (:215-218: ``args.text_autoencoder`` or ``dataset_mode == 5`` -- "the
    image is actually text input"): ``caption_tokens = clip.tokenize(d['caption'])``; ``prefix =
    clip_model.encode_text(caption_tokens).float()``; then the common tail (:221-234) -- normalise, modality offset,
    ``clip_project``, ``generate_beam`` / ``generate2`` -- and the predictions JSON of :260-261.  ``tokenize`` defaults to
    ``capdec_amd.clip.tokenize`` (needs the CLIP BPE vocabulary file); any callable ``list[str] -> int [N, 77]`` does.
    This rank tokenises / encodes / decodes only its block of ``data``; the ids are gathered in caption order."""
    if tokenize is None:
        from .clip import tokenize
    cdist.check_world(rank, world)
    n = len(data)
    lo, hi = cdist.shard_bounds(n, rank, world)
    feats = []
    for b0 in range(lo, hi, max(1, text_batch)):
        toks = tokenize([d["caption"] for d in data[b0:min(hi, b0 + max(1, text_batch))]])
        feats.append(clip_model.encode_text(toks).float())
    stop = tokenizer.encode('.')[0]
    if feats:
        ids, lens, _ = caption_ids(model, torch.cat(feats), stop, beam, 5, entry_length, dont_normalize_prefix, modality_offset)
    else:                              # an empty shard (more ranks than captions) still takes part in the gather
        dev = next(model.parameters()).device
        ids = torch.zeros(0, entry_length, dtype=torch.int32, device=dev)
        lens = torch.zeros(0, dtype=torch.int32, device=dev)
    ids, lens, _ = cdist.gather_ids(ids, lens, n)
    ids, lens = ids.cpu().numpy(), lens.cpu().numpy()
    new_data = [{"caption": tokenizer.decode(list(ids[i, :int(lens[i])])).lower(), "image_id": d["image_id"]}
                for i, d in enumerate(data)]
    if out_path and rank == 0:
        with open(out_path, 'w') as outfile:
            json.dump(new_data, outfile)
    return new_data
