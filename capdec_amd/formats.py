"""On-disk formats either side of the hot path, host-side only.

* embeddings pickle written by reference ``embeddings_generator.main`` (:91-101) and read by
  ``train.ClipCocoDataset`` (train.py:63-71): ``{"clip_embedding": Tensor[N_img, D] (fp16 when produced on a
  GPU, possibly empty), "captions": [ {image_id, caption, id, clip_embedding: row_idx, filename?} ],
  "clip_embedding_text_dave": Tensor[N_cap, D]}``
* modality-offset pickle ``others/CLIP_embeddings_centers_info.pkl`` (modality_offset_calculator.py:52-62)
* checkpoints ``torch.save(model.state_dict())`` (train.py:359-371)
* predictions JSON ``[{"caption": lower-cased text, "image_id": id}]`` (predictions_runner.py:260-261,301)
"""
from __future__ import annotations

import json
import pickle
from typing import Dict, List, Optional, Sequence, Tuple

import torch


def _as_f32(t) -> torch.Tensor:
    if t is None or (not torch.is_tensor(t) and not isinstance(t, (list, tuple)) and not hasattr(t, "shape")):
        # the reference stores the scalar 0 for a side it did not compute ('clip_embedding_text_dave': 0 when
        # add_text_embedding is False, embeddings_generator.py:101): an absent side
        return torch.zeros(0, 0)
    if isinstance(t, (list, tuple)):
        t = torch.cat([torch.as_tensor(x).reshape(1, -1) if torch.as_tensor(x).dim() < 2 else torch.as_tensor(x)
                       for x in t], dim=0) if len(t) else torch.zeros(0, 0)
    return torch.as_tensor(t).detach().float().cpu()


def load_embeddings_pickle(path: str) -> Tuple[torch.Tensor, torch.Tensor, List[dict]]:
    """-> (image embeddings fp32 [N_img, D], text embeddings fp32 [N_cap, D], captions).  fp16 tensors are
    upcast like the reference's ``.float()`` (train.py:70); an absent / empty side comes back as [0, D]."""
    with open(path, "rb") as f:
        d = pickle.load(f)
    img = _as_f32(d.get("clip_embedding", torch.zeros(0, 0)))
    txt = _as_f32(d.get("clip_embedding_text_dave", torch.zeros(0, 0)))
    if img.dim() == 0:          # a 0-d tensor is a placeholder too
        img = torch.zeros(0, 0)
    if txt.dim() == 0:
        txt = torch.zeros(0, 0)
    dim = max(img.shape[-1] if img.numel() else 0, txt.shape[-1] if txt.numel() else 0)
    if img.numel() == 0:
        img = torch.zeros(0, dim)
    if txt.numel() == 0:
        txt = torch.zeros(0, dim)
    if dim == 0:                # both sides absent
        return torch.zeros(0, 0), torch.zeros(0, 0), list(d.get("captions", []))
    return img.reshape(-1, dim), txt.reshape(-1, dim), list(d.get("captions", []))


def save_embeddings_pickle(path: str, captions: Sequence[dict], text_embeddings: Optional[torch.Tensor] = None,
                           image_embeddings: Optional[torch.Tensor] = None, half: bool = False) -> None:
    """Writes the reference layout (embeddings_generator.py:96-101).  ``half=True`` stores fp16 like a GPU run of
    the reference; each caption dict gets ``clip_embedding`` = its row index if it has none."""
    caps = []
    for i, c in enumerate(captions):
        c = dict(c)
        c.setdefault("clip_embedding", i)
        caps.append(c)

    def prep(t):
        if t is None:
            return torch.zeros(0)
        t = t.detach().cpu()
        return t.half() if half else t.float()
    with open(path, "wb") as f:
        pickle.dump({"clip_embedding": prep(image_embeddings), "captions": caps,
                     "clip_embedding_text_dave": prep(text_embeddings)}, f)


def load_modality_offset(path: str, which: str = "offset_to_add_in_inference") -> torch.Tensor:
    """``get_precalculated_centers()[which]`` (others/modality_offset_calculator.py:59-62;
    predictions_runner.py:165-166 uses 'offset_to_add_in_inference', train.py:332-334 '..._in_training')."""
    with open(path, "rb") as f:
        d = pickle.load(f)
    if which not in d:
        raise KeyError(f"{which!r} not in {sorted(d)}")
    return _as_f32(d[which]).reshape(1, -1)


def load_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    """``torch.load(ckpt, map_location=cpu)`` (predictions_runner.py:461) -> flat state dict; accepts a bare state
    dict or a dict wrapping one under 'state_dict' / 'model'."""
    obj = torch.load(path, map_location="cpu")
    if isinstance(obj, dict) and not any(torch.is_tensor(v) for v in obj.values()):
        for k in ("state_dict", "model"):
            if k in obj and isinstance(obj[k], dict):
                return obj[k]
    if hasattr(obj, "state_dict"):
        return obj.state_dict()
    return obj


def write_predictions_json(path: str, captions: Sequence[str], image_ids: Sequence) -> List[dict]:
    out = [{"caption": c.lower(), "image_id": i} for c, i in zip(captions, image_ids)]
    with open(path, "w") as f:
        json.dump(out, f)
    return out
