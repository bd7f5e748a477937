"""Drop-in surface of the `clip` package calls the reference makes (embeddings_generator.py:49,
72-89; predictions_runner.py:158-161,212-220): ``load`` -> (model, preprocess) with
``model.encode_text(tokens)`` / ``model.encode_image(images)`` running as HIP kernels.

Differences that follow from the offline environment (no checkpoint / BPE vocabulary files):
``load`` takes an OpenAI-CLIP state dict (or a ``.pt`` path holding one) instead of a model name
to download; ``tokenize`` needs a vocabulary file and otherwise raises.  The visual tower is a ViT
(ViT-B/32: head_dim 64) or a ModifiedResNet (RN50x4, the reference's default backbone: folded
BatchNorm, im2col + GEMM convolutions, attention pool).  Compute is fp32-accurate by
default; ``load(..., precision="fp16")`` runs the towers' block GEMMs with fp16 operands (fp32
accumulate, fp32 residual stream / LayerNorm / softmax) -- the precision class of the reference
on a GPU, where ``clip.load`` converts the model to fp16 and the callers cast the result back
with ``.float()`` (predictions_runner.py:218,220)."""
from __future__ import annotations

from typing import Dict, Optional, Union

import torch

from ._capi import CapdecError
from .engine import Engine


class ClipModel:
    def __init__(self, state_dict: Dict[str, torch.Tensor], device=0, precision: str = "fp32"):
        idx = device if isinstance(device, int) else (torch.device(device).index or 0)
        self._engine = Engine(idx)
        if precision not in ("fp32", "fp16", "bf16"):
            raise CapdecError("clip.load: precision must be 'fp32', 'fp16' or 'bf16'")
        self.precision = precision
        if precision != "fp32":
            self._engine.set_gemm_mode({"fp16": "f16", "bf16": "bf16"}[precision])
        sd = {k: v for k, v in state_dict.items()}
        self.has_vision = "visual.conv1.weight" in sd           # ViT patch embedding or the ResNet stem
        self.has_text = "token_embedding.weight" in sd
        if not (self.has_vision or self.has_text):
            raise CapdecError("clip.load: the state dict holds neither a text nor a visual tower")
        self._engine.load_clip(sd, text=self.has_text, vision=self.has_vision)
        self.context_length = self._engine.clip_text["context_length"] if self.has_text else 77
        self.input_resolution = self._engine.clip_vision["image_size"] if self.has_vision else 224
        self.device = self._engine.device

    def eval(self):
        return self

    def encode_text(self, text: torch.Tensor) -> torch.Tensor:
        """int tokens [N, 77] -> [N, 512] (not normalised, reference embeddings_generator.py:86-87)"""
        if not self.has_text:
            raise CapdecError("this CLIP state dict has no text tower")
        return self._engine.clip_encode_text(text)

    def encode_image(self, image: torch.Tensor) -> torch.Tensor:
        if not self.has_vision:
            raise CapdecError("this CLIP state dict has no visual tower")
        return self._engine.clip_encode_image(image)


def load(name_or_state_dict: Union[str, Dict[str, torch.Tensor]], device=0, jit: bool = False,
         precision: str = "fp32"):
    """``clip.load("ViT-B/32", device=device, jit=False)`` -> (model, preprocess).  Pass the state dict
    (or the path of a ``torch.save``d one / an OpenAI ``.pt`` archive readable by torch.load)."""
    if isinstance(name_or_state_dict, str):
        obj = torch.load(name_or_state_dict, map_location="cpu")
        sd = obj.state_dict() if hasattr(obj, "state_dict") else obj
    else:
        sd = name_or_state_dict
    model = ClipModel(sd, device, precision)
    return model, Preprocess(model._engine, model.input_resolution)


class Preprocess:
    """The ``preprocess`` callable ``clip.load`` returns (reference predictions_runner.py:212
    ``preprocess(image_raw).unsqueeze(0).to(device)``, embeddings_generator.py:72): Resize(n_px, BICUBIC) ->
    CenterCrop(n_px) -> RGB -> ToTensor -> Normalize, bit-identical to the PIL / torchvision pipeline but computed by
    the HIP kernels of ``preprocess.hip``.  Accepts a PIL image (anything with ``.convert``), a uint8 [H, W, 3] numpy
    array or torch tensor; returns fp32 [3, n_px, n_px] on the device.  ``batch(images)`` does a whole list in one launch
    pair -> [N, 3, n_px, n_px]; ``stretch=True`` is the reference's ``clip_transform_full`` (:116-122)."""

    def __init__(self, engine: Engine, n_px: int = 224, stretch: bool = False):
        self._engine, self.n_px, self.stretch = engine, int(n_px), bool(stretch)

    @staticmethod
    def _rgb(image):
        if hasattr(image, "convert"):                      # PIL.Image: `lambda image: image.convert("RGB")`
            import numpy as np
            return np.asarray(image.convert("RGB"))
        return image

    def batch(self, images) -> torch.Tensor:
        return self._engine.preprocess_images([self._rgb(im) for im in images], self.n_px, self.stretch)

    def __call__(self, image) -> torch.Tensor:
        return self.batch([image])[0]


def preprocess_tensor(image: torch.Tensor) -> torch.Tensor:
    """Normalisation step of the reference's transform (mean / std of predictions_runner.py:121) for a
    float image tensor [3, 224, 224] in [0, 1]; resize / crop / decoding stay with the caller (PIL and
    torchvision are not part of this package)."""
    mean = torch.tensor((0.48145466, 0.4578275, 0.40821073), device=image.device).view(3, 1, 1)
    std = torch.tensor((0.26862954, 0.26130258, 0.27577711), device=image.device).view(3, 1, 1)
    return (image - mean) / std


_tokenizer = None


def tokenize(texts, context_length: int = 77, truncate: bool = False):
    """``clip.tokenize`` (reference embeddings_generator.py:81, predictions_runner.py:217).  The BPE vocabulary
    (bpe_simple_vocab_16e6.txt.gz) is not shipped and not downloadable here: point CAPDEC_CLIP_BPE at it."""
    global _tokenizer
    if _tokenizer is None:
        import os
        path = os.environ.get("CAPDEC_CLIP_BPE")
        if not path or not os.path.exists(path):
            raise CapdecError("clip.tokenize needs the CLIP BPE vocabulary: set CAPDEC_CLIP_BPE to "
                              "bpe_simple_vocab_16e6.txt.gz, or pass pre-tokenised int [N, 77] rows to encode_text")
        from .bpe import ClipBPE
        _tokenizer = ClipBPE(path)
    return _tokenizer.tokenize(texts, context_length, truncate)
