"""Drop-in surface of reference ``gpt2_prefix.py`` for the caption hot path: ``MappingType``
(:15-18), ``MLP`` (:114-126), ``ClipCaptionModel`` (:139-171), ``ClipCaptionPrefix`` (:178-186).

Same names, constructor arguments and attribute surface (``.clip_project``, ``.gpt``,
``.gpt.transformer.wte``, ``.prefix_length``, ``.load_state_dict``, ``.eval``, ``.to``,
``.parameters``), but the arithmetic runs in libcapdec_hip.so on an MI355X.  ``forward`` is the
forward pass of the reference's train step (:145-155; train.py:251-260) -- logits and loss, inference
only: backward and the optimiser are outside this path.
"""
from __future__ import annotations

from collections import OrderedDict
from enum import Enum
from types import SimpleNamespace
from typing import Dict, Iterator, Optional, Tuple

import torch

from . import synth
from ._capi import CapdecError
from .engine import Engine


class MappingType(Enum):
    """reference gpt2_prefix.py:15-18; ``Transformer`` is train.py:42-44's name for the encoder."""
    MLP = 'mlp'
    TransformerEncoder = 'transformer_encoder'
    TransformerDecoder = 'transformer_decoder'
    Transformer = 'transformer_encoder'   # alias (same value -> same member)


def _device_index(device) -> int:
    if isinstance(device, int):
        return device
    d = torch.device(device)
    if d.type != "cuda":
        raise CapdecError(f"capdec_amd runs on HIP devices only (got {d}); there is no CPU fallback")
    return d.index or 0


class _HipModule:
    """Minimal nn.Module-like shell: a named fp32 state dict on the host plus a lazily created
    Engine (the HIP context that holds the device copy)."""

    def __init__(self):
        self._sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self._engine: Optional[Engine] = None
        self._device_index = 0
        self._dirty = True
        self._measure = False
        self.training = False

    # --- nn.Module-ish API the reference's callers use
    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def _before_engine_close(self):
        """hook: values that only the device holds (weights updated by train steps) are pulled to the host first"""

    def to(self, device):
        idx = _device_index(device)
        if idx != self._device_index and self._engine is not None:
            self._before_engine_close()
            self._engine.close()
            self._engine = None
            self._dirty = True
        self._device_index = idx
        return self

    def cuda(self, device: int = 0):
        return self.to(torch.device("cuda", device))

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict(self._sd)

    def parameters(self, recurse: bool = True) -> Iterator[torch.Tensor]:
        """Device-resident handles; the reference only uses them to find the device
        (gpt2_prefix_eval.py:64,135)."""
        dev = torch.device("cuda", self._device_index)
        if not self._sd:
            yield torch.empty(0, device=dev)
        for v in self._sd.values():
            yield torch.empty(0, device=dev, dtype=torch.float32)

    def use_measurement_build(self, on: bool = True):
        """measurement tooling only (bench.py's untimed tail, tools/): move this module to a context of the library's
        -DCAPDEC_MEASURE build (ablation / override knobs, the diverged-beam hook).  The current context -- weights, KV
        cache, workspaces -- is released; the weights are uploaded again on the next use."""
        if bool(on) != self._measure:
            if self._engine is not None:
                self._before_engine_close()
                self._engine.close()
                self._engine = None
            self._measure = bool(on)
            self._dirty = True
        return self

    def release(self):
        """drop this module's device context -- weights, KV cache, workspaces (the decode of 5000 captions x beam 5 holds
        140 GB) -- keeping the host copy: the next use creates a context and uploads the weights again"""
        if self._engine is not None:
            self._before_engine_close()
            self._engine.close()
            self._engine = None
        self._dirty = True
        return self

    @property
    def engine(self) -> Engine:
        if self._engine is None:
            self._engine = Engine(self._device_index, measure=self._measure)
            self._dirty = True
        if self._dirty:
            self._upload(self._engine)
            self._dirty = False
        return self._engine

    def _upload(self, eng: Engine):
        raise NotImplementedError

    def __call__(self, *a, **k):
        return self.forward(*a, **k)


class MLP(_HipModule):
    """reference gpt2_prefix.py:114-126: Linear -> Tanh -> Linear for ``sizes`` of length 3
    (the only shape ClipCaptionModel builds, :167-168)."""

    def __init__(self, sizes: Tuple[int, ...], bias=True, act=None, _owner: Optional[_HipModule] = None):
        super().__init__()
        if len(sizes) != 3 or not bias:
            raise CapdecError("MLP mapper: only the 3-size, biased form used by ClipCaptionModel is supported")
        self.sizes = tuple(int(s) for s in sizes)
        self._owner = _owner

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        need = ["model.0.weight", "model.0.bias", "model.2.weight", "model.2.bias"]
        missing = [k for k in need if k not in sd]
        if missing and strict:
            raise RuntimeError(f"Missing key(s) in state_dict: {missing}")
        for k in need:
            self._sd[k] = sd[k].detach().float().cpu()
        self._dirty = True
        return SimpleNamespace(missing_keys=missing, unexpected_keys=[k for k in sd if k not in need])

    def _upload(self, eng: Engine):
        eng.load_mapper_mlp({"clip_project." + k: v for k, v in self._sd.items()})

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        eng = self._owner.engine if self._owner is not None else self.engine
        return eng.mapper_forward(x).reshape(x.shape[0], self.sizes[2])   # [B, P*768] like nn.Sequential


class ClipCaptionModel(_HipModule):
    """reference gpt2_prefix.py:139-171 (ctor kwargs of :157-158; ``prefix_size`` is accepted as
    the train.py:262 spelling of ``prefix_dim``)."""

    def __init__(self, prefix_length: int, clip_length: Optional[int] = None, prefix_dim: int = 640,
                 num_layers: int = 8, mapping_type: MappingType = MappingType.TransformerEncoder,
                 prefix_size: Optional[int] = None, gpt2_dims: synth.GPT2Dims = synth.GPT2_SMALL):
        super().__init__()
        from . import transformer_mapper
        if prefix_size is not None:
            prefix_dim = prefix_size
        clip_length = prefix_length if clip_length is None else clip_length
        self.prefix_length = prefix_length
        self.clip_length = clip_length
        self.prefix_dim = prefix_dim
        self.num_layers = num_layers
        self.mapping_type = mapping_type
        self.gpt_dims = gpt2_dims
        self.gpt_embedding_size = gpt2_dims.n_embd
        self.gpt = _Gpt2Facade(self)
        if mapping_type == MappingType.TransformerEncoder:
            self.clip_project = transformer_mapper.TransformerMapper(prefix_dim, self.gpt_embedding_size, prefix_length,
                                                                     clip_length, num_layers, _owner=self)
        elif mapping_type == MappingType.MLP:
            self.clip_project = MLP((prefix_dim, (self.gpt_embedding_size * prefix_length) // 2,
                                     self.gpt_embedding_size * prefix_length), _owner=self)
        else:
            raise CapdecError("MappingType.TransformerDecoder (TransformerEncoderDecoder) is outside the hot path")

    def get_dummy_token(self, batch_size: int, device) -> torch.Tensor:
        return torch.zeros(batch_size, self.prefix_length, dtype=torch.int64, device=device)

    # ---- train steps update the mapper ON THE DEVICE (capdec_amd.train.train_step): the host copy follows lazily
    _device_ahead = False

    _train_gpt = False          # scope of the train steps run on this model (capdec_amd.train.train_step)
    _drop_seed_epoch = 0        # bump to re-seed the stream (e.g. after torch.manual_seed)

    def _train_shapes(self):
        """ordered {state-dict name: shape} of every tensor the current train scope updates, in the device's slot order"""
        shapes = OrderedDict(("clip_project." + k, v) for k, v in self._mapper_shapes().items())
        if self._train_gpt:
            for n in Engine.train_gpt2_tensor_names(self.gpt_dims.n_layer):
                shapes["gpt." + n] = tuple(self._sd["gpt." + n].shape)
        return shapes

    def _mapper_shapes(self):
        """ordered {name: shape} of the mapper's trainable tensors, in the device's slot order"""
        mlp = self.mapping_type == MappingType.MLP
        names = Engine.train_tensor_names("mlp" if mlp else "transformer", self.num_layers)
        sd = self.clip_project._sd
        missing = [n for n in names if n not in sd]
        if missing or len(names) != len(sd):
            raise CapdecError(f"mapper state dict does not match its trainable tensors: {missing or sorted(set(sd) - set(names))}")
        return OrderedDict((n, tuple(sd[n].shape)) for n in names)

    def _pull_mapper(self):
        if not self._device_ahead or self._engine is None:
            return
        for k, v in self._engine.mapper_parameters(self._train_shapes()).items():
            self._sd[k] = v.cpu()
            if k.startswith("clip_project."):
                self.clip_project._sd[k[len("clip_project."):]] = self._sd[k]
        if self._train_gpt and "gpt.lm_head.weight" in self._sd:
            self._sd["gpt.lm_head.weight"] = self._sd["gpt.transformer.wte.weight"]      # tied
        self._device_ahead = False

    def _before_engine_close(self):
        self._pull_mapper()

    def state_dict(self):
        """reference train.py:359-371 saves ``model.state_dict()``: after train steps the mapper's tensors are read back
        from the device first"""
        self._pull_mapper()
        return OrderedDict(self._sd)

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        """Reference checkpoints (train.py:359-371).  Ignores the transformers-4.24 buffers
        ``gpt.transformer.h.{i}.attn.bias`` / ``.attn.masked_bias``; accepts fp16 tensors."""
        mapper_keys = [k for k in sd if k.startswith("clip_project.")]
        gpt_keys = [k for k in sd if k.startswith("gpt.") and not (k.endswith(".attn.bias") or k.endswith(".attn.masked_bias"))]
        if strict and (not mapper_keys or "gpt.transformer.wte.weight" not in sd):
            raise RuntimeError("Missing key(s) in state_dict: clip_project.* / gpt.transformer.wte.weight")
        self._device_ahead = False
        self._sd = OrderedDict((k, sd[k].detach().float().cpu()) for k in mapper_keys + gpt_keys)
        self.clip_project.load_state_dict({k[len("clip_project."):]: self._sd[k] for k in mapper_keys}, strict=strict)
        self._dirty = True
        unexpected = [k for k in sd if k not in self._sd and not (k.endswith(".attn.bias") or k.endswith(".attn.masked_bias"))]
        if strict and unexpected:
            raise RuntimeError(f"Unexpected key(s) in state_dict: {unexpected}")
        return SimpleNamespace(missing_keys=[], unexpected_keys=unexpected)

    def _upload(self, eng: Engine):
        if "gpt.transformer.wte.weight" not in self._sd:
            raise CapdecError("ClipCaptionModel has no weights: call load_state_dict(checkpoint) first")
        eng.load_gpt2(self._sd, "gpt.", n_head=self.gpt_dims.n_head, ln_eps=self.gpt_dims.ln_eps)
        self.clip_project._upload(eng)

    def forward(self, tokens, prefix, mask=None, labels=None):
        """The forward pass of the reference's TRAIN step (train.py:251-260,348; gpt2_prefix.py:145-155), inference only
        (no autograd: backward / optimiser are outside this path): logits [B, P + L, V] of
        ``cat(clip_project(prefix).view(-1, P, d), wte(tokens))`` under the causal mask; ``out.loss`` (when ``labels`` is
        given) is GPT2LMHeadModel's shifted cross-entropy over ``cat(dummy_token, tokens)``.  ``mask`` must be the
        reference dataset's right-padding mask (ones for the prefix and the real tokens, zeros for the padding at the END,
        train.py:52-63): under the causal mask the real positions never see a padded key, so their logits equal the
        reference's; logits AT padded positions are unspecified (the train loss ignores them, train.py:349)."""
        tokens = tokens.to(torch.device("cuda", self._device_index))
        if mask is not None:
            m = mask.to(tokens.device) > 0
            if m.shape != (tokens.shape[0], self.prefix_length + tokens.shape[1]) or \
                    bool((m[:, 1:] & ~m[:, :-1]).any()) or not bool(m[:, :self.prefix_length].all()):
                raise CapdecError("forward: only the reference dataset's right-padding mask is supported "
                                  "([B, prefix_length + L], ones then zeros)")
        embedding_text = self.gpt.transformer.wte(tokens)
        prefix_projections = self.clip_project(prefix).view(-1, self.prefix_length, self.gpt_embedding_size)
        embedding_cat = torch.cat((prefix_projections, embedding_text), dim=1)
        logits = self.engine.gpt2_logits(embedding_cat, all_positions=True)
        loss = None
        if labels is not None:
            dummy_token = self.get_dummy_token(tokens.shape[0], tokens.device)
            lab = torch.cat((dummy_token, tokens.long()), dim=1)
            loss = self.engine.cross_entropy(logits[:, :-1], lab[:, 1:])          # device kernel (capdec_cross_entropy)
        return SimpleNamespace(logits=logits, loss=loss)

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)


class ClipCaptionPrefix(ClipCaptionModel):
    """reference gpt2_prefix.py:178-186: exposes only the mapper's parameters."""

    def parameters(self, recurse: bool = True):
        dev = torch.device("cuda", self._device_index)
        for k in self._sd:
            if k.startswith("clip_project."):
                yield torch.empty(0, device=dev)

    def train(self, mode: bool = True):
        super().train(mode)
        return self


class _Gpt2Facade:
    """What the decode code touches on ``model.gpt`` (reference gpt2_prefix_eval.py:74-76,105,
    151,163,181; predictions_runner.py:186)."""

    def __init__(self, owner: ClipCaptionModel):
        self._owner = owner
        self.transformer = SimpleNamespace(wte=self._wte)
        # what GPT2LMHeadModel.from_pretrained('gpt2').config says (reference gpt2_prefix.py:160): the train step of a
        # plain ClipCaptionModel in train() mode applies these (capdec_amd.train.train_step)
        self.config = SimpleNamespace(resid_pdrop=0.1, embd_pdrop=0.1, attn_pdrop=0.1)

    def _wte(self, ids: torch.Tensor) -> torch.Tensor:
        return self._owner.engine.wte(ids)

    def get_input_embeddings(self):
        self._owner._pull_mapper()          # (after full-scope train steps the current wte lives on the device)
        w = self._owner._sd["gpt.transformer.wte.weight"]
        return SimpleNamespace(weight=w.to(torch.device("cuda", self._owner._device_index)))

    def eval(self):
        return self

    def __call__(self, inputs_embeds: torch.Tensor = None, **kw):
        if inputs_embeds is None or kw.get("labels") is not None:
            raise CapdecError("model.gpt(...) supports inference on inputs_embeds only")
        return SimpleNamespace(logits=self._owner.engine.gpt2_logits(inputs_embeds, all_positions=True))
