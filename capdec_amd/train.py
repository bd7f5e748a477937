"""Drop-in surface of reference ``train.py``: ``noise_injection`` (:27-39), ``get_uniform_ball_noise`` (:18-24),
``MappingType`` (:42-44), the ``ClipCaptionModel`` ctor spelling with ``prefix_size`` (:262), and the train loop of
:317-392: ``AdamW``, ``get_linear_schedule_with_warmup``, ``train_step`` (= :345-353 as one device call) and ``train``
(with the validation pass and ``loss_per_epoch.json`` of :373-391).  Both configurations of the reference: ``--only_prefix``
(``ClipCaptionPrefix``: the mapper is trained, GPT-2 frozen and in eval mode) and the default (``ClipCaptionModel``: GPT-2
is trained too, with transformers' dropouts of 0.1 while the model is in ``train()`` mode)."""
from __future__ import annotations

import math
from typing import Optional

import torch

from ._capi import CapdecError
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


# ---------------------------------------------------------------------------------------------------------------------
# the train loop of reference train.py:317-392 for a frozen GPT-2 (--only_prefix)
# ---------------------------------------------------------------------------------------------------------------------
class AdamW:
    """``transformers.AdamW(params, lr, betas, eps, weight_decay, correct_bias)`` as reference train.py:326 constructs it
    (transformers 4.24 defaults: eps 1e-6, no weight decay, bias correction).  The moments live in the HIP context of the
    model; this object carries the hyper-parameters and the current learning rate (``param_groups[0]["lr"]``, which the
    scheduler rewrites).  ``step`` / ``zero_grad`` exist so the reference's loop reads the same, but the update itself is
    applied by ``train_step`` (gradient and update are one device call)."""

    def __init__(self, params=None, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-6, weight_decay: float = 0.0,
                 correct_bias: bool = True):
        if not correct_bias:
            raise CapdecError("AdamW: correct_bias=False is not implemented")
        if params is not None:
            list(params)                       # (a generator of device handles: nothing to keep)
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, correct_bias=True)
        self.param_groups = [dict(self.defaults, initial_lr=lr)]

    def step(self):
        pass

    def zero_grad(self):
        pass


class _LinearSchedule:
    def __init__(self, optimizer: AdamW, num_warmup_steps: int, num_training_steps: int):
        self.optimizer, self.warmup, self.total, self.last_epoch = optimizer, num_warmup_steps, num_training_steps, 0
        self._apply()

    def _factor(self) -> float:
        k = self.last_epoch
        if k < self.warmup:
            return float(k) / float(max(1, self.warmup))
        return max(0.0, float(self.total - k) / float(max(1, self.total - self.warmup)))

    def _apply(self):
        for g in self.optimizer.param_groups:
            g["lr"] = g["initial_lr"] * self._factor()

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]


def get_linear_schedule_with_warmup(optimizer: AdamW, num_warmup_steps: int, num_training_steps: int) -> _LinearSchedule:
    """``transformers.get_linear_schedule_with_warmup`` (reference train.py:329-331): like torch's LambdaLR it sets the lr of
    step 0 at construction -- 0 when there is a warm-up, so the reference's first update moves nothing"""
    return _LinearSchedule(optimizer, num_warmup_steps, num_training_steps)


def train_step(model: ClipCaptionModel, optimizer: AdamW, tokens: torch.Tensor, mask: Optional[torch.Tensor],
               prefix: torch.Tensor, *, apply_update: bool = True, dropout_masks: Optional[torch.Tensor] = None,
               wait: bool = True) -> Optional[float]:
    """reference train.py:345-351 and :353 for one batch -- ``model.zero_grad(); outputs = model(tokens, prefix, mask);
    loss = cross_entropy(outputs.logits[:, P-1:-1], tokens, ignore_index=0); loss.backward(); optimizer.step();
    optimizer.zero_grad()`` -- as ONE device call (capdec_train_step); returns ``loss.item()``.  ``prefix`` is the batch
    after ``noise_injection`` (:347); the caller steps the scheduler afterwards (:352), as the reference does.
    ``mask`` must be the dataset's right-padding mask (or None): see ClipCaptionModel.forward.

    A ``ClipCaptionPrefix`` trains its mapper against a frozen GPT-2 in eval mode; a plain ``ClipCaptionModel`` trains
    GPT-2 too (reference train.py:306-308) and, while ``model.training``, applies GPT-2's dropouts
    (``model.gpt.config.resid_pdrop`` = embd_pdrop = attn_pdrop, 0.1 like ``GPT2LMHeadModel.from_pretrained('gpt2')``)
    from a Philox stream seeded from torch's global seed -- or from ``dropout_masks`` (uint8, 1 = keep: every site of the
    step concatenated in call order, include/capdec.h), the hook the parity test uses.  ``wait=False`` enqueues the step
    and returns None (``model.engine.train_loss()`` reads losses later)."""
    full = not isinstance(model, ClipCaptionPrefix)      # a plain ClipCaptionModel trains GPT-2 too (reference train.py:306-308)
    if model._train_gpt != full:
        model._pull_mapper()
        model._train_gpt = full
    eng = model.engine
    eng.train_set_scope(full)
    p_drop = float(model.gpt.config.resid_pdrop) if (full and model.training) else 0.0
    # (the probability and the Philox key live in the HIP CONTEXT and a new context starts at p = 0: what was last set is
    #  remembered on the Engine object, so a model that moved to another device / build sets its dropouts up again)
    if full and (p_drop, model._drop_seed_epoch) != getattr(eng, "_drop_state", None):
        if not (model.gpt.config.embd_pdrop == model.gpt.config.attn_pdrop == model.gpt.config.resid_pdrop):
            raise CapdecError("train_step: one dropout probability for embd / attn / resid (transformers' defaults are equal)")
        eng.train_set_dropout(p_drop, _next_seed())
        eng._drop_state = (p_drop, model._drop_seed_epoch)
    if dropout_masks is not None:
        if p_drop <= 0.0:
            raise CapdecError("train_step: dropout_masks need a ClipCaptionModel in train() mode with dropout > 0")
        eng.train_set_dropout_masks(dropout_masks)
    if not tokens.is_cuda and tokens.numel() and (int(tokens.min()) < 0 or int(tokens.max()) >= model.gpt_dims.vocab):
        raise IndexError("train_step: token id out of range (the reference's embedding lookup raises here too)")
    if mask is not None:
        # (checked where the tensors ARE: the DataLoader's host tensors cost no device round trip, so train()'s steps stay
        #  enqueued; device tensors are checked on the device)
        m = mask > 0
        tokens = tokens.to(m.device)
        P = model.prefix_length
        if m.shape != (tokens.shape[0], P + tokens.shape[1]) or bool((m[:, 1:] & ~m[:, :-1]).any()) or not bool(m[:, :P].all()):
            raise CapdecError("train_step: only the reference dataset's right-padding mask is supported")
        if bool(((tokens != 0) & ~m[:, P:]).any()):
            raise CapdecError("train_step: a non-zero token under a zero mask (the loss would read it, the reference's "
                              "attention would not)")
    tokens = tokens.to(torch.device("cuda", model._device_index))
    g = optimizer.param_groups[0]
    loss = eng.train_step(prefix, tokens, g["lr"], g["betas"], g["eps"], g["weight_decay"], apply_update, wait=wait)
    if loss is not None and math.isnan(loss) and bool(((tokens < 0) | (tokens >= model.gpt_dims.vocab)).any()):
        raise IndexError("train_step: token id out of range (the reference's embedding lookup raises here too); "
                         "no weight was updated")
    if apply_update:
        model._device_ahead = True
    return loss


def mapper_gradients(model: ClipCaptionModel):
    """``{name: p.grad}`` of the mapper after the last ``train_step`` (names as in ``clip_project.state_dict()``)"""
    return model.engine.mapper_gradients(model._mapper_shapes())


def all_gradients(model: ClipCaptionModel):
    """``{state-dict name: p.grad}`` of every tensor of the model's current train scope (mapper; GPT-2 too for a plain
    ClipCaptionModel) after the last ``train_step``"""
    return model.engine.mapper_gradients(model._train_shapes())


def validation_loss(model: ClipCaptionModel, val_dataset, batch_size: int, prefix_length: Optional[int] = None) -> float:
    """the validation pass of reference train.py:373-388: eval mode, no noise injection, ``model(tokens, prefix, mask)`` ->
    ``cross_entropy(logits[:, P-1:-1], tokens, ignore_index=0)`` per batch (the device kernel capdec_cross_entropy),
    averaged over the batches of a shuffled, drop_last DataLoader"""
    from torch.utils.data import DataLoader
    P = model.prefix_length if prefix_length is None else prefix_length
    loader = DataLoader(val_dataset, batch_size=batch_size, shuffle=True, drop_last=True)
    was_training = model.training
    model.eval()
    val_loss = 0.0
    try:
        for tokens, mask, prefix in loader:
            prefix = prefix.to(device, dtype=torch.float32)
            outputs = model(tokens, prefix, mask)
            logits = outputs.logits[:, P - 1:-1]
            val_loss += float(model.engine.cross_entropy(logits, tokens, ignore_index=0))
    finally:
        if was_training:
            model.train()
    return val_loss / max(1, len(loader))


def train(dataset, model: ClipCaptionModel, args, warmup_steps: int = 5000, output_dir: str = ".", output_prefix: str = "",
          val_dataset=None):
    """reference train.py:317-392: ``dataset`` yields ``(tokens, mask, prefix)`` like train.ClipCocoDataset.__getitem__
    (:66-75); ``args`` needs ``bs, epochs, lr, noise_variance, uniform_noise, dont_norm, save_every`` (and optionally
    ``modality_offset``: the tensor the reference reads from others/CLIP_embeddings_centers_info.pkl at :333-337).
    ``val_dataset`` (same item layout) stands for ``ClipCocoDataset(args.val_pt, ...)`` of :374 -- parsing the embedding
    pickle is the caller's; with it every epoch ends with the validation pass of :373-388, and ``loss_per_epoch.json``
    (:390-391: ``{'train': [...], 'val': [...]}``) is rewritten after every epoch either way.  The steps of an epoch are
    enqueued without waiting for the device; the epoch's mean loss is the device's running sum (capdec_train_loss)."""
    import json
    import os
    import sys
    from torch.utils.data import DataLoader
    os.makedirs(output_dir, exist_ok=True)
    if getattr(args, "val_pt", "") and val_dataset is None:
        raise CapdecError("train: args.val_pt names an embedding pickle -- build the validation dataset from it and pass "
                          "val_dataset= (dataset parsing is outside this package)")
    model.train()
    optimizer = AdamW(model.parameters(), lr=args.lr)
    loader = DataLoader(dataset, batch_size=args.bs, shuffle=True, drop_last=True)
    scheduler = get_linear_schedule_with_warmup(optimizer, num_warmup_steps=warmup_steps,
                                                num_training_steps=args.epochs * len(loader))
    modality_offset = getattr(args, "modality_offset", None)
    loss_per_epoch_train = []
    loss_per_epoch_val = []
    for epoch in range(args.epochs):
        print(f">>> Training epoch {epoch} / {args.epochs}")
        sys.stdout.flush()
        model.engine.train_loss(reset=True)              # the device's running loss sum starts at 0
        for idx, (tokens, mask, prefix) in enumerate(loader):
            prefix = prefix.to(device, dtype=torch.float32)
            prefix = noise_injection(prefix, args.noise_variance, modality_offset=modality_offset,
                                     uniform_noise=args.uniform_noise, dont_norm=args.dont_norm)
            train_step(model, optimizer, tokens, mask, prefix, wait=False)
            scheduler.step()
            if (idx + 1) % 10000 == 0:
                torch.save(model.state_dict(), os.path.join(output_dir, f"{output_prefix}_latest.pt"))
        _, total, counted = model.engine.train_loss()
        if counted != len(loader):      # a step whose token ids were out of range is flagged on the device and left out of the sum
            raise IndexError(f"train: {len(loader) - counted} batch(es) of epoch {epoch} held a token id out of range (the "
                             "reference's embedding lookup raises on the first of them); no weight was updated by those steps")
        loss_per_epoch_train.append(total / max(1, len(loader)))
        print('loss_per_epoch_train: ', loss_per_epoch_train)
        if epoch % args.save_every == 0 or epoch == args.epochs - 1:
            torch.save(model.state_dict(), os.path.join(output_dir, f"{output_prefix}-{epoch:03d}.pt"))
        if val_dataset is not None:
            loss_per_epoch_val.append(validation_loss(model, val_dataset, args.bs, getattr(dataset, "prefix_length", None)))
            print('loss_per_epoch_val: ', loss_per_epoch_val)
        with open(os.path.join(output_dir, "loss_per_epoch.json"), 'w') as f:
            json.dump({'train': loss_per_epoch_train, 'val': loss_per_epoch_val}, f)
    return model
