"""Host-side driver of one MI355X: owns a ``capdec_ctx`` and hands device pointers of torch
CUDA(HIP) tensors to the C ABI.  torch is plumbing only (device memory, streams); all
arithmetic happens in libcapdec_hip.so."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _capi
from ._capi import CapdecError, check


def _f32(t: torch.Tensor) -> np.ndarray:
    """contiguous fp32 host array view of a (CPU) tensor; fp16 pickles are upcast like the
    reference's `.float()` (train.py:70)."""
    return np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())


def _fp(a: np.ndarray):
    return a.ctypes.data_as(_capi.c_float_p)


class Engine:
    """One context per GPU / rank."""

    def __init__(self, device: int = 0, kv_budget_bytes: int = 0, measure: bool = False):
        """measure=True: the context lives in libcapdec_hip_measure.so (built with -DCAPDEC_MEASURE: ablation / override
        knobs, the diverged-beam hook) -- tools/ and bench.py's untimed tail only, never the product path"""
        if not torch.cuda.is_available():
            raise CapdecError("capdec_amd needs a HIP device (MI355X); none is visible and there is no CPU fallback")
        if measure:
            if not os.path.exists(_capi.MEASURE_LIB_PATH):
                raise CapdecError(f"{_capi.MEASURE_LIB_PATH} is missing: run `python -m capdec_amd.build --measure`")
            self.lib = _capi.load_library(_capi.MEASURE_LIB_PATH)
        else:
            self.lib = _capi.load_library()
        self.measure = bool(measure)
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        h = C.c_void_p()
        self._chk(self.lib.capdec_create(self.device_index, C.byref(h)), "capdec_create")
        self._h = h
        if kv_budget_bytes:
            self._chk(self.lib.capdec_set_kv_budget(self._h, kv_budget_bytes), "set_kv_budget")
        self.gpt_dims: Optional[Dict[str, int]] = None
        self.mapper: Optional[Dict[str, int]] = None
        self.comm: Optional[Tuple[int, int]] = None          # (rank, world) of the C-ABI RCCL communicator, if any

    def _chk(self, rc: int, what: str = ""):
        check(rc, what, self.lib)

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.capdec_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _sync_stream(self):
        """run on torch's current stream so torch copies and our kernels stay ordered"""
        s = torch.cuda.current_stream(self.device).cuda_stream
        self._chk(self.lib.capdec_set_stream(self._h, C.c_void_p(s)), "set_stream")

    def synchronize(self):
        self._chk(self.lib.capdec_synchronize(self._h), "synchronize")

    def _dev(self, t: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
        return t.to(device=self.device, dtype=dtype).contiguous()

    # ------------------------------------------------------------------ weights
    def load_gpt2(self, sd: Dict[str, torch.Tensor], prefix: str = "gpt.", n_head: int = 12, ln_eps: float = 1e-5):
        """Accepts the reference checkpoint layout (train.py:359-371): tied
        ``gpt.lm_head.weight`` is checked against wte; transformers-4.24 buffers
        ``h.{i}.attn.bias`` / ``.attn.masked_bias`` are ignored."""
        t = prefix + "transformer."
        wte = _f32(sd[t + "wte.weight"])
        lm = sd.get(prefix + "lm_head.weight")
        if lm is not None and lm.data_ptr() != sd[t + "wte.weight"].data_ptr():
            if not torch.equal(lm.float().cpu(), sd[t + "wte.weight"].float().cpu()):
                raise CapdecError("checkpoint lm_head.weight differs from wte.weight (untied heads are not supported)")
        wpe = _f32(sd[t + "wpe.weight"])
        n_layer = 0
        while f"{t}h.{n_layer}.ln_1.weight" in sd:
            n_layer += 1
        if n_layer == 0:
            raise CapdecError(f"no GPT-2 blocks under {t!r}")
        keep = [wte, wpe]
        layers = (_capi.Gpt2Layer * n_layer)()
        names = [("ln_1_w", "ln_1.weight"), ("ln_1_b", "ln_1.bias"), ("c_attn_w", "attn.c_attn.weight"),
                 ("c_attn_b", "attn.c_attn.bias"), ("c_proj_w", "attn.c_proj.weight"), ("c_proj_b", "attn.c_proj.bias"),
                 ("ln_2_w", "ln_2.weight"), ("ln_2_b", "ln_2.bias"), ("c_fc_w", "mlp.c_fc.weight"),
                 ("c_fc_b", "mlp.c_fc.bias"), ("mlp_c_proj_w", "mlp.c_proj.weight"), ("mlp_c_proj_b", "mlp.c_proj.bias")]
        for i in range(n_layer):
            for field, key in names:
                a = _f32(sd[f"{t}h.{i}.{key}"])
                keep.append(a)
                setattr(layers[i], field, _fp(a))
        d = wte.shape[1]
        if tuple(_f32(sd[f"{t}h.0.attn.c_attn.weight"]).shape) != (d, 3 * d):
            raise CapdecError("c_attn.weight must be Conv1D layout [d, 3d]")
        lnw, lnb = _f32(sd[t + "ln_f.weight"]), _f32(sd[t + "ln_f.bias"])
        keep += [lnw, lnb]
        w = _capi.Gpt2Weights(n_layer, n_head, d, wte.shape[0], wpe.shape[0], ln_eps, _fp(wte), _fp(wpe), layers,
                              _fp(lnw), _fp(lnb))
        self._chk(self.lib.capdec_load_gpt2(self._h, C.byref(w)), "capdec_load_gpt2")
        self.gpt_dims = dict(n_layer=n_layer, n_head=n_head, d=d, vocab=wte.shape[0], n_pos=wpe.shape[0])

    def load_mapper_mlp(self, sd: Dict[str, torch.Tensor], prefix: str = "clip_project."):
        w1, b1 = _f32(sd[prefix + "model.0.weight"]), _f32(sd[prefix + "model.0.bias"])
        w2, b2 = _f32(sd[prefix + "model.2.weight"]), _f32(sd[prefix + "model.2.bias"])
        d = self.gpt_dims["d"] if self.gpt_dims else 768
        hidden, D = w1.shape
        P = w2.shape[0] // d
        self._chk(self.lib.capdec_load_mapper_mlp(self._h, D, P, hidden, _fp(w1), _fp(b1), _fp(w2), _fp(b2)),
              "capdec_load_mapper_mlp")
        self.mapper = dict(kind="mlp", D=D, P=P, d=d)

    def load_mapper_transformer(self, sd: Dict[str, torch.Tensor], prefix: str = "clip_project.", num_heads: int = 8):
        lw, lb = _f32(sd[prefix + "linear.weight"]), _f32(sd[prefix + "linear.bias"])
        pc = _f32(sd[prefix + "prefix_const"])
        P, d = pc.shape
        clip_len = lw.shape[0] // d
        n_layers = 0
        while f"{prefix}transformer.layers.{n_layers}.norm1.weight" in sd:
            n_layers += 1
        keep = [lw, lb, pc]
        layers = (_capi.TMapperLayer * n_layers)()
        names = [("norm1_w", "norm1.weight"), ("norm1_b", "norm1.bias"), ("to_queries_w", "attn.to_queries.weight"),
                 ("to_keys_values_w", "attn.to_keys_values.weight"), ("project_w", "attn.project.weight"),
                 ("project_b", "attn.project.bias"), ("norm2_w", "norm2.weight"), ("norm2_b", "norm2.bias"),
                 ("fc1_w", "mlp.fc1.weight"), ("fc1_b", "mlp.fc1.bias"), ("fc2_w", "mlp.fc2.weight"),
                 ("fc2_b", "mlp.fc2.bias")]
        for i in range(n_layers):
            for field, key in names:
                a = _f32(sd[f"{prefix}transformer.layers.{i}.{key}"])
                keep.append(a)
                setattr(layers[i], field, _fp(a))
        hid = _f32(sd[f"{prefix}transformer.layers.0.mlp.fc1.weight"]).shape[0]
        w = _capi.TMapperWeights(lw.shape[1], P, clip_len, n_layers, num_heads, d, hid, _fp(lw), _fp(lb), _fp(pc), layers)
        self._chk(self.lib.capdec_load_mapper_transformer(self._h, C.byref(w)), "capdec_load_mapper_transformer")
        self.mapper = dict(kind="transformer", D=lw.shape[1], P=P, d=d, clip_length=clip_len, num_layers=n_layers)

    # ------------------------------------------------------------------ prefix stage
    def normalize_prefix(self, x: torch.Tensor, normalize: bool = True, offset: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = self._dev(x)
        n, dim = x.shape
        out = torch.empty_like(x)
        if n == 0:            # an empty shard (world > N): nothing to launch, no NULL pointer into the C ABI
            return out
        off = self._dev(offset).reshape(-1) if offset is not None else None
        self._sync_stream()
        self._chk(self.lib.capdec_normalize_prefix(self._h, x.data_ptr(), n, dim, int(normalize),
                                               off.data_ptr() if off is not None else None, out.data_ptr()),
              "capdec_normalize_prefix")
        return out

    def noise_inject(self, x: torch.Tensor, variance: float, offset: Optional[torch.Tensor] = None,
                     uniform: bool = False, dont_norm: bool = False, seed: int = 0,
                     noise: Optional[torch.Tensor] = None, u: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = self._dev(x)
        n, dim = x.shape
        out = torch.empty_like(x)
        if n == 0:
            return out
        off = self._dev(offset).reshape(-1) if offset is not None else None
        nz = self._dev(noise) if noise is not None else None
        uu = self._dev(u) if u is not None else None
        self._sync_stream()
        self._chk(self.lib.capdec_noise_inject(self._h, x.data_ptr(), n, dim, float(variance),
                                           off.data_ptr() if off is not None else None, int(uniform), int(dont_norm),
                                           int(seed) & 0xFFFFFFFFFFFFFFFF, nz.data_ptr() if nz is not None else None,
                                           uu.data_ptr() if uu is not None else None, out.data_ptr()),
              "capdec_noise_inject")
        return out

    def mapper_forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.mapper is None:
            raise CapdecError("no mapper loaded")
        x = self._dev(x)
        n = x.shape[0]
        if x.shape[1] != self.mapper["D"]:
            raise CapdecError(f"mapper expects prefix_dim {self.mapper['D']}, got {x.shape[1]}")
        out = torch.empty(n, self.mapper["P"], self.mapper["d"], device=self.device, dtype=torch.float32)
        if n:
            self._sync_stream()
            self._chk(self.lib.capdec_mapper_forward(self._h, x.data_ptr(), n, out.data_ptr()), "capdec_mapper_forward")
        return out

    # ------------------------------------------------------------------ CLIP towers
    def _clip_blocks(self, sd, prefix: str, layers: int, keep: list):
        blocks = (_capi.ClipBlock * layers)()
        names = [("ln_1_w", "ln_1.weight"), ("ln_1_b", "ln_1.bias"), ("in_proj_w", "attn.in_proj_weight"),
                 ("in_proj_b", "attn.in_proj_bias"), ("out_proj_w", "attn.out_proj.weight"),
                 ("out_proj_b", "attn.out_proj.bias"), ("ln_2_w", "ln_2.weight"), ("ln_2_b", "ln_2.bias"),
                 ("c_fc_w", "mlp.c_fc.weight"), ("c_fc_b", "mlp.c_fc.bias"), ("c_proj_w", "mlp.c_proj.weight"),
                 ("c_proj_b", "mlp.c_proj.bias")]
        for i in range(layers):
            for field, key in names:
                a = _f32(sd[f"{prefix}transformer.resblocks.{i}.{key}"])
                keep.append(a)
                setattr(blocks[i], field, _fp(a))
        return blocks

    @staticmethod
    def _count_blocks(sd, prefix: str) -> int:
        n = 0
        while f"{prefix}transformer.resblocks.{n}.ln_1.weight" in sd:
            n += 1
        return n

    def load_clip(self, sd: Dict[str, torch.Tensor], text: bool = True, vision: bool = True):
        """OpenAI CLIP state dict (what `clip.load(...)[0].state_dict()` holds; fp16 weights are upcast).  The visual
        tower is a ViT (`visual.class_embedding` present) or a ModifiedResNet (`visual.layer1...`: RN50x4, the
        reference's default backbone)."""
        keep: list = []
        if text:
            te, pe = _f32(sd["token_embedding.weight"]), _f32(sd["positional_embedding"])
            proj = _f32(sd["text_projection"])
            lw, lb = _f32(sd["ln_final.weight"]), _f32(sd["ln_final.bias"])
            keep += [te, pe, proj, lw, lb]
            layers = self._count_blocks(sd, "")
            width = te.shape[1]
            w = _capi.ClipTextWeights(pe.shape[0], te.shape[0], width, width // 64, layers, proj.shape[1], _fp(te),
                                      _fp(pe), self._clip_blocks(sd, "", layers, keep), _fp(lw), _fp(lb), _fp(proj))
            self._chk(self.lib.capdec_load_clip_text(self._h, C.byref(w)), "capdec_load_clip_text")
            self.clip_text = dict(context_length=pe.shape[0], embed_dim=proj.shape[1], vocab=te.shape[0])
        if vision and "visual.layer1.0.conv1.weight" in sd:
            self._load_clip_resnet(sd)
        elif vision:
            if "visual.conv1.weight" not in sd or "visual.class_embedding" not in sd:
                raise CapdecError("no visual tower in the state dict (neither visual.class_embedding nor visual.layer1)")
            cw = _f32(sd["visual.conv1.weight"])
            ce, pe = _f32(sd["visual.class_embedding"]), _f32(sd["visual.positional_embedding"])
            l1w, l1b = _f32(sd["visual.ln_pre.weight"]), _f32(sd["visual.ln_pre.bias"])
            l2w, l2b = _f32(sd["visual.ln_post.weight"]), _f32(sd["visual.ln_post.bias"])
            proj = _f32(sd["visual.proj"])
            keep += [cw, ce, pe, l1w, l1b, l2w, l2b, proj]
            width, patch = cw.shape[0], cw.shape[-1]
            image = int(round((pe.shape[0] - 1) ** 0.5)) * patch
            layers = self._count_blocks(sd, "visual.")
            w = _capi.ClipVisionWeights(image, patch, width, width // 64, layers, proj.shape[1], _fp(cw), _fp(ce), _fp(pe),
                                        _fp(l1w), _fp(l1b), self._clip_blocks(sd, "visual.", layers, keep), _fp(l2w),
                                        _fp(l2b), _fp(proj))
            self._chk(self.lib.capdec_load_clip_vision(self._h, C.byref(w)), "capdec_load_clip_vision")
            self.clip_vision = dict(image_size=image, embed_dim=proj.shape[1])

    def _load_clip_resnet(self, sd: Dict[str, torch.Tensor]):
        """ModifiedResNet tower (`visual.conv1..3`, `visual.layer1..4`, `visual.attnpool`): every convolution goes
        over with its BatchNorm statistics; the library folds them at load time."""
        keep: list = []

        def conv_bn(conv: str, bn: str) -> _capi.ConvBn:
            if conv + ".weight" not in sd:
                return _capi.ConvBn()
            w = _f32(sd[conv + ".weight"])
            t = [w] + [_f32(sd[f"{bn}.{k}"]) for k in ("weight", "bias", "running_mean", "running_var")]
            keep.extend(t)
            if w.shape[2] != w.shape[3]:
                raise CapdecError(f"{conv}: square kernels only")
            return _capi.ConvBn(*[_fp(x) for x in t], w.shape[1], w.shape[0], w.shape[2])

        stem = (_capi.ConvBn * 3)(*[conv_bn(f"visual.conv{i}", f"visual.bn{i}") for i in (1, 2, 3)])
        layers = [len({k.split(".")[2] for k in sd if k.startswith(f"visual.layer{li}.")}) for li in (1, 2, 3, 4)]
        blocks = []
        for li, nb in enumerate(layers, start=1):
            for b in range(nb):
                p = f"visual.layer{li}.{b}."
                blocks += [conv_bn(p + f"conv{i}", p + f"bn{i}") for i in (1, 2, 3)]
                blocks.append(conv_bn(p + "downsample.0", p + "downsample.1"))
        barr = (_capi.ConvBn * len(blocks))(*blocks)
        a = "visual.attnpool."
        pos = _f32(sd[a + "positional_embedding"])
        names = ["q_proj", "k_proj", "v_proj", "c_proj"]
        proj = [(_f32(sd[a + n + ".weight"]), _f32(sd[a + n + ".bias"])) for n in names]
        keep += [pos] + [t for pr in proj for t in pr]
        width = sd["visual.layer1.0.conv1.weight"].shape[0]
        image = int(round((pos.shape[0] - 1) ** 0.5)) * 32
        embed = proj[3][0].shape[0]
        w = _capi.ClipResNetWeights(image, width, embed, (C.c_int * 4)(*layers), stem, barr, _fp(pos),
                                    *[_fp(t) for pr in proj for t in pr])
        self._chk(self.lib.capdec_load_clip_resnet(self._h, C.byref(w)), "capdec_load_clip_resnet")
        self.clip_vision = dict(image_size=image, embed_dim=embed, kind="resnet")

    def clip_encode_text(self, tokens: torch.Tensor) -> torch.Tensor:
        t = self._dev(tokens, torch.int32)
        n = t.shape[0]
        if t.shape[1] != self.clip_text["context_length"]:
            raise CapdecError(f"token rows must have context_length {self.clip_text['context_length']}")
        out = torch.empty(n, self.clip_text["embed_dim"], device=self.device, dtype=torch.float32)
        if n:
            self._sync_stream()
            self._chk(self.lib.capdec_clip_encode_text(self._h, t.data_ptr(), n, out.data_ptr()), "capdec_clip_encode_text")
        return out

    def clip_encode_image(self, pixels: torch.Tensor) -> torch.Tensor:
        x = self._dev(pixels)
        n, S = x.shape[0], self.clip_vision["image_size"]
        if tuple(x.shape[1:]) != (3, S, S):
            raise CapdecError(f"images must be [n, 3, {S}, {S}] (already preprocessed)")
        out = torch.empty(n, self.clip_vision["embed_dim"], device=self.device, dtype=torch.float32)
        if n:
            self._sync_stream()
            self._chk(self.lib.capdec_clip_encode_image(self._h, x.data_ptr(), n, out.data_ptr()), "capdec_clip_encode_image")
        return out

    # ------------------------------------------------------------------ GPT-2
    def gpt2_logits(self, embeds: torch.Tensor, all_positions: bool = True) -> torch.Tensor:
        e = self._dev(embeds)
        n, L, d = e.shape
        V = self.gpt_dims["vocab"]
        out = torch.empty((n, L, V) if all_positions else (n, V), device=self.device, dtype=torch.float32)
        self._sync_stream()
        self._chk(self.lib.capdec_gpt2_logits(self._h, e.data_ptr(), n, L, int(all_positions), out.data_ptr()),
              "capdec_gpt2_logits")
        return out

    def cross_entropy(self, logits: torch.Tensor, labels: torch.Tensor, ignore_index: int = -100) -> torch.Tensor:
        """mean token cross-entropy of ``logits`` [..., V] against ``labels`` [...] on the device (capdec_cross_entropy);
        rows with ``labels == ignore_index`` are skipped -- what ``nnf.cross_entropy`` computes at train.py:349"""
        V = logits.shape[-1]
        lg = self._dev(logits).reshape(-1, V)
        lab = self._dev(labels.reshape(-1), torch.int32)
        out = torch.empty(1, device=self.device, dtype=torch.float32)
        self._sync_stream()
        self._chk(self.lib.capdec_cross_entropy(self._h, lg.data_ptr(), V, lab.data_ptr(), lg.shape[0], V, int(ignore_index),
                                            out.data_ptr()), "capdec_cross_entropy")
        return out[0]

    def wte(self, ids: torch.Tensor) -> torch.Tensor:
        shape = tuple(ids.shape)
        i = self._dev(ids.reshape(-1), torch.int32)
        out = torch.empty(i.numel(), self.gpt_dims["d"], device=self.device, dtype=torch.float32)
        self._sync_stream()
        self._chk(self.lib.capdec_wte_lookup(self._h, i.data_ptr(), i.numel(), out.data_ptr()), "capdec_wte_lookup")
        return out.view(*shape, -1)

    # ------------------------------------------------------------------ decode
    def decode_greedy(self, prefix_embed: torch.Tensor, stop_id: int, entry_length: int = 67,
                      alt_stop_id: int = 764) -> Tuple[torch.Tensor, torch.Tensor]:
        p = self._dev(prefix_embed)
        n, P, _ = p.shape
        ids = torch.empty(n, entry_length, device=self.device, dtype=torch.int32)
        lens = torch.empty(n, device=self.device, dtype=torch.int32)
        self._sync_stream()
        self._chk(self.lib.capdec_decode_greedy(self._h, p.data_ptr(), n, P, int(stop_id), int(alt_stop_id),
                                            int(entry_length), ids.data_ptr(), lens.data_ptr()), "capdec_decode_greedy")
        return ids, lens

    def decode_greedy_forced(self, prefix_embed: torch.Tensor, forced_ids: torch.Tensor):
        """teacher-forced greedy decode: feeds ``forced_ids`` [n, T] and returns (arg-max ids [n, T], stats [n, T, 3] =
        (top-1 logit, top-2 logit, logsumexp) of every step)"""
        p = self._dev(prefix_embed)
        f = self._dev(forced_ids, torch.int32)
        n, P, _ = p.shape
        T = f.shape[1]
        ids = torch.empty(n, T, device=self.device, dtype=torch.int32)
        stats = torch.empty(n, T, 3, device=self.device, dtype=torch.float32)
        self._sync_stream()
        self._chk(self.lib.capdec_decode_greedy_forced(self._h, p.data_ptr(), n, P, T, f.data_ptr(), ids.data_ptr(),
                                                   stats.data_ptr()), "capdec_decode_greedy_forced")
        return ids, stats

    def decode_beam(self, prefix_embed: torch.Tensor, stop_id: int, beam_size: int = 5, entry_length: int = 67,
                    temperature: float = 1.0):
        """-> ids [n, beam, T], lens [n, beam], mean-log-prob scores [n, beam] (sorted by score
        descending, like the list generate_beam returns) and order [n, beam] (reference's
        internal beam index of each returned row)."""
        p = self._dev(prefix_embed)
        n, P, _ = p.shape
        ids = torch.empty(n, beam_size, entry_length, device=self.device, dtype=torch.int32)
        lens = torch.empty(n, beam_size, device=self.device, dtype=torch.int32)
        scores = torch.empty(n, beam_size, device=self.device, dtype=torch.float32)
        order = torch.empty(n, beam_size, device=self.device, dtype=torch.int32)
        self._sync_stream()
        self._chk(self.lib.capdec_decode_beam(self._h, p.data_ptr(), n, P, int(beam_size), int(stop_id), int(entry_length),
                                          float(temperature), ids.data_ptr(), lens.data_ptr(), scores.data_ptr(),
                                          order.data_ptr()), "capdec_decode_beam")
        return ids, lens, scores, order

    # ------------------------------------------------------------------ caption-shard communicator (RCCL through the C ABI)
    def comm_unique_id(self) -> bytes:
        """rank 0: the 128-byte communicator id every rank passes to :meth:`comm_init`"""
        buf = C.create_string_buffer(128)
        self._chk(self.lib.capdec_comm_unique_id(buf), "capdec_comm_unique_id")
        return buf.raw

    def comm_init(self, rank: int, world: int, comm_id: bytes):
        if len(comm_id) != 128:
            raise CapdecError("comm_init: the communicator id is 128 bytes")
        self._chk(self.lib.capdec_comm_init(self._h, int(rank), int(world), comm_id), "capdec_comm_init")
        self.comm = (int(rank), int(world))

    def comm_info(self) -> Tuple[int, int]:
        """(rank, ranks) as RCCL itself reports them for the C-ABI communicator (ncclCommUserRank / ncclCommCount);
        (0, 1) without one"""
        r, n = C.c_int(0), C.c_int(1)
        self._chk(self.lib.capdec_comm_info(self._h, C.byref(r), C.byref(n)), "capdec_comm_info")
        return r.value, n.value

    def comm_destroy(self):
        self._chk(self.lib.capdec_comm_destroy(self._h), "capdec_comm_destroy")
        self.comm = None

    def gather_rows(self, local: torch.Tensor, n_total: int) -> torch.Tensor:
        """all-gather of this rank's row block (int32 or fp32, [n_local, ...]) in rank order -> [n_total, ...]"""
        if local.dtype not in (torch.int32, torch.float32):
            raise CapdecError("gather_rows: int32 or float32 rows")
        t = local.to(self.device).contiguous()
        row = int(np.prod(t.shape[1:])) if t.dim() > 1 else 1
        out = torch.empty((n_total,) + tuple(t.shape[1:]), dtype=t.dtype, device=self.device)
        self._sync_stream()
        self._chk(self.lib.capdec_gather_rows(self._h, t.data_ptr() if t.numel() else None, t.shape[0], max(row, 1),
                                          int(n_total), out.data_ptr() if out.numel() else None), "capdec_gather_rows")
        return out

    # ------------------------------------------------------------------ image preprocessing
    def preprocess_images(self, images, n_px: int = 224, stretch: bool = False,
                          mean=(0.48145466, 0.4578275, 0.40821073), std=(0.26862954, 0.26130258, 0.27577711)) -> torch.Tensor:
        """uint8 RGB images (list of [H, W, 3] numpy arrays / torch tensors, any sizes) -> fp32 [n, 3, n_px, n_px] on the
        device: PIL-exact bicubic Resize(n_px) + CenterCrop(n_px) (or stretch to n_px x n_px) + ToTensor + Normalize
        (reference predictions_runner.py:116-122,212; embeddings_generator.py:72), one HIP launch pair per batch."""
        arrs = []
        for im in images:
            a = im.detach().cpu().numpy() if isinstance(im, torch.Tensor) else np.asarray(im)
            if a.ndim != 3 or a.shape[2] != 3 or a.dtype != np.uint8:
                raise CapdecError(f"preprocess_images: expected uint8 [H, W, 3], got {a.dtype} {a.shape}")
            arrs.append(np.ascontiguousarray(a))
        n = len(arrs)
        out = torch.empty(n, 3, n_px, n_px, device=self.device, dtype=torch.float32)
        if n == 0:
            return out
        sizes = np.array([a.size for a in arrs], dtype=np.int64)
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        flat = torch.from_numpy(np.concatenate([a.reshape(-1) for a in arrs])).to(self.device)
        hs = np.array([a.shape[0] for a in arrs], dtype=np.int32)
        ws = np.array([a.shape[1] for a in arrs], dtype=np.int32)
        m = np.asarray(mean, dtype=np.float32)
        sd = np.asarray(std, dtype=np.float32)
        self._sync_stream()
        self._chk(self.lib.capdec_preprocess_images(
            self._h, flat.data_ptr(), offs.ctypes.data_as(C.POINTER(C.c_int64)), hs.ctypes.data_as(C.POINTER(C.c_int32)),
            ws.ctypes.data_as(C.POINTER(C.c_int32)), n, int(n_px), int(bool(stretch)), _fp(m), _fp(sd), out.data_ptr()),
            "capdec_preprocess_images")
        return out

    def timer_start(self):
        """hipEvent on the engine's stream (capdec_timer_start): the reference Timer's ``starter.record()``"""
        self._sync_stream()
        self._chk(self.lib.capdec_timer_start(self._h), "timer_start")

    def timer_stop_ms(self) -> float:
        """records the end event, synchronises, returns the elapsed milliseconds (capdec_timer_stop_ms)"""
        ms = C.c_float(0.0)
        self._chk(self.lib.capdec_timer_stop_ms(self._h, C.byref(ms)), "timer_stop_ms")
        return float(ms.value)

    def decode_stats(self) -> Dict[str, int]:
        """steps run / compactions / activation row-steps of the last decode call"""
        a, b, r = C.c_int(0), C.c_int(0), C.c_longlong(0)
        self._chk(self.lib.capdec_decode_stats(self._h, C.byref(a), C.byref(b), C.byref(r)), "decode_stats")
        return dict(steps=a.value, compactions=b.value, row_steps=r.value)

    def set_compact(self, on: bool = True):
        """finished-caption compaction at the decode loop's poll points on / off (capdec_set_compact; default on)"""
        self._chk(self.lib.capdec_set_compact(self._h, int(bool(on))), "set_compact")

    def decode_step_rows(self):
        """activation rows of every decode step (after the prefill) of the last decode call (capdec_decode_step_rows)"""
        buf, n = (C.c_int * 1024)(), C.c_int(0)
        self._chk(self.lib.capdec_decode_step_rows(self._h, buf, 1024, C.byref(n)), "decode_step_rows")
        return [int(buf[i]) for i in range(min(n.value, 1024))]

    def decode_counters(self) -> Dict[str, float]:
        """kv_slots_per_position: mean number of distinct K/V slots a (caption, position) of the last beam decode read
        (1 = beams share their whole history, beam = nothing); saturated_quads: GEMM-operand quads clamped to the fp16
        range since the last call (the call resets the counter)"""
        kv, sat = C.c_double(0.0), C.c_longlong(0)
        self._chk(self.lib.capdec_decode_counters(self._h, C.byref(kv), C.byref(sat)), "decode_counters")
        return dict(kv_slots_per_position=kv.value, saturated_quads=sat.value)

    # ---- the train step (reference train.py:344-354): scope 0 = --only_prefix, scope 1 = the default run
    @staticmethod
    def train_tensor_names(mapping: str, num_layers: int = 8):
        """the mapper's trainable tensors in the order capdec_train_get indexes them (names as in ``clip_project.state_dict()``)"""
        if mapping == "mlp":
            return ["model.0.weight", "model.0.bias", "model.2.weight", "model.2.bias"]
        names = ["linear.weight", "linear.bias", "prefix_const"]
        for i in range(num_layers):
            names += [f"transformer.layers.{i}.{n}" for n in (
                "norm1.weight", "norm1.bias", "attn.to_queries.weight", "attn.to_keys_values.weight", "attn.project.weight",
                "attn.project.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight",
                "mlp.fc2.bias")]
        return names

    def train_step(self, prefix: torch.Tensor, tokens: torch.Tensor, lr: float, betas=(0.9, 0.999), eps: float = 1e-6,
                   weight_decay: float = 0.0, apply_update: bool = True, wait: bool = True) -> Optional[float]:
        """one iteration on the device-resident weights (capdec_train_step): ``prefix`` [B, D] AFTER noise injection,
        ``tokens`` [B, L] right-padded with 0; returns the loss of train.py:349.  ``wait=False`` only enqueues the step and
        returns None (``train_loss`` reads the loss, and the running sum, later: no device round trip per step)"""
        self._sync_stream()
        x = prefix.to(self.device, torch.float32).contiguous()
        tok = tokens.to(self.device, torch.int32).contiguous()
        if x.dim() != 2 or tok.dim() != 2 or x.shape[0] != tok.shape[0]:
            raise CapdecError("train_step: prefix [B, D] and tokens [B, L] expected")
        loss = C.c_float(0.0)
        self._chk(self.lib.capdec_train_step(self._h, x.data_ptr(), tok.data_ptr(), tok.shape[0], tok.shape[1], float(lr),
                                             float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
                                             int(bool(apply_update)), C.byref(loss) if wait else None), "train_step")
        if not wait:
            x.record_stream(torch.cuda.current_stream(self.device))      # (no-ops on the stream they were made on: kept
            tok.record_stream(torch.cuda.current_stream(self.device))    #  for callers that build batches on a side stream)
            return None
        return float(loss.value)

    def train_loss(self, reset: bool = False):
        """(loss of the last step, sum of the losses since the last reset, number of those steps) -- capdec_train_loss"""
        last, total, n = C.c_float(0.0), C.c_double(0.0), C.c_longlong(0)
        self._chk(self.lib.capdec_train_loss(self._h, C.byref(last), C.byref(total), C.byref(n), int(bool(reset))), "train_loss")
        return float(last.value), float(total.value), int(n.value)

    def train_set_dropout(self, p: float, seed: int = 0):
        """GPT-2's dropouts in scope 1 (transformers' default 0.1); keep-masks from the Philox stream keyed by ``seed``"""
        self._chk(self.lib.capdec_train_set_dropout(self._h, float(p), int(seed) & 0xFFFFFFFFFFFFFFFF), "train_set_dropout")

    def train_set_dropout_masks(self, masks: torch.Tensor):
        """keep-masks (uint8, 1 = keep) of the NEXT train step, all sites concatenated in call order (capdec.h)"""
        self._sync_stream()
        m = masks.to(self.device, torch.uint8).contiguous().flatten()
        self._chk(self.lib.capdec_train_set_dropout_masks(self._h, m.data_ptr(), m.numel()), "train_set_dropout_masks")

    def train_get_dropout_masks(self, n: int) -> torch.Tensor:
        """the mask stream the last train step used (uint8 [n])"""
        self._sync_stream()
        out = torch.empty(n, device=self.device, dtype=torch.uint8)
        self._chk(self.lib.capdec_train_get_dropout_masks(self._h, out.data_ptr(), n), "train_get_dropout_masks")
        return out

    @staticmethod
    def dropout_stream_size(B: int, S: int, d: int, n_head: int, n_layer: int) -> int:
        """bytes of one step's mask stream: embd [B, S, d] + per block attn [B, H, S, S], resid [B, S, d], mlp [B, S, d]"""
        return B * S * d + n_layer * (B * n_head * S * S + 2 * B * S * d)

    def _train_get(self, kind: int, shapes) -> Dict[str, torch.Tensor]:
        """``shapes``: ordered {name: shape} in the order of train_tensor_names"""
        out = {}
        for i, name in enumerate(shapes):
            t = torch.empty(shapes[name], device=self.device, dtype=torch.float32)
            self._chk(self.lib.capdec_train_get(self._h, kind, i, t.data_ptr(), t.numel()), "train_get")
            out[name] = t
        return out

    def mapper_parameters(self, shapes) -> Dict[str, torch.Tensor]:
        """the mapper's current tensors on the device (after train steps: the updated values), keyed like MLP.state_dict"""
        return self._train_get(0, shapes)

    def mapper_gradients(self, shapes) -> Dict[str, torch.Tensor]:
        """d loss / d tensor of the last train_step (what loss.backward() leaves in .grad, train.py:350)"""
        return self._train_get(1, shapes)

    @staticmethod
    def train_gpt2_tensor_names(n_layer: int):
        """the GPT-2 tensors that follow the mapper's in capdec_train_get when the scope includes GPT-2 (state-dict names
        without the ``gpt.`` prefix)"""
        names = ["transformer.wte.weight", "transformer.wpe.weight"]
        for i in range(n_layer):
            names += [f"transformer.h.{i}.{n}" for n in (
                "ln_1.weight", "ln_1.bias", "attn.c_attn.weight", "attn.c_attn.bias", "attn.c_proj.weight", "attn.c_proj.bias",
                "ln_2.weight", "ln_2.bias", "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias")]
        return names + ["transformer.ln_f.weight", "transformer.ln_f.bias"]

    def train_set_scope(self, train_gpt: bool):
        """False: the mapper only, GPT-2 frozen and in eval mode (--only_prefix); True: GPT-2 as well (the reference's
        default run; dropout per train_set_dropout)"""
        self._chk(self.lib.capdec_train_set_scope(self._h, int(bool(train_gpt))), "train_set_scope")

    def train_reset(self):
        """a fresh optimizer: drops the AdamW moments and the step count (scope and dropout setting stay)"""
        self._chk(self.lib.capdec_train_reset(self._h), "train_reset")

    def second_pass_rows(self) -> int:
        """(row, step) pairs of the last decode call whose lm_head top 5 went through the exact second pass (capdec.h)"""
        n = C.c_longlong(0)
        self._chk(self.lib.capdec_decode_second_pass_rows(self._h, C.byref(n)), "decode_second_pass_rows")
        return n.value

    def set_batch_invariant(self, on: bool = True):
        """results independent of batch size / chunking / sharding (no split-K, pinned kernel variants); see capdec.h"""
        self._chk(self.lib.capdec_set_batch_invariant(self._h, int(bool(on))), "set_batch_invariant")

    def set_debug_diverge(self, on: bool = True):
        """measurement builds only (Engine(measure=True)): beams never share history -- the worst-case decode-attention
        traffic; the shipped library does not export the hook"""
        if not hasattr(self.lib, "capdec_set_debug_diverge") or not self.measure:
            raise CapdecError("capdec_set_debug_diverge exists only in the measurement build: Engine(measure=True)")
        self._chk(self.lib.capdec_set_debug_diverge(self._h, int(bool(on))), "set_debug_diverge")

    # ------------------------------------------------------------------ hooks
    def gemm(self, a: torch.Tensor, bt: torch.Tensor, bias=None, resid=None, act: int = 0) -> torch.Tensor:
        a, bt = self._dev(a), self._dev(bt)
        M, K = a.shape
        N = bt.shape[0]
        out = torch.empty(M, N, device=self.device, dtype=torch.float32)
        b = self._dev(bias) if bias is not None else None
        r = self._dev(resid) if resid is not None else None
        self._sync_stream()
        self._chk(self.lib.capdec_gemm_f32(self._h, a.data_ptr(), K, bt.data_ptr(), K, out.data_ptr(), N, M, N, K,
                                       b.data_ptr() if b is not None else None,
                                       r.data_ptr() if r is not None else None, N, act), "capdec_gemm_f32")
        return out

    def set_gemm_mode(self, mode: str):
        """'f16x2' (default: fp32-accurate, operands as two fp16 planes, 3 MFMAs per product), 'bf16x3' (fp32-accurate,
        three bf16 planes, 6 MFMAs per product), 'f32' (native fp32 MFMA), 'bf16' (bf16 GEMM operands and KV cache, fp32
        accumulate: BASELINE configs[1]) or 'f16' (fp16 GEMM operands: the reference's CLIP-tower arithmetic on a GPU)"""
        self._chk(self.lib.capdec_set_gemm_mode(self._h, {"f32": 0, "bf16x3": 1, "bf16": 2, "f16x2": 3, "f16": 4}[mode]), "set_gemm_mode")

    def gemm_mode(self) -> str:
        return ["f32", "bf16x3", "bf16", "f16x2", "f16"][self.lib.capdec_get_gemm_mode(self._h)]

    def profile_enable(self, on=True):
        """True / 1: time every launch; N > 1: every N-th launch of each kernel family (sampling); False: off"""
        self._chk(self.lib.capdec_profile_enable(self._h, int(on)), "profile_enable")

    def profile_reset(self):
        self._chk(self.lib.capdec_profile_reset(self._h), "profile_reset")

    def profile_get(self) -> Dict[str, Dict[str, float]]:
        cnt = C.c_int(24)                 # in: capacity of the arrays below, out: families filled
        names = (C.c_char_p * 24)()
        ms = (C.c_float * 24)()
        launches = (C.c_int64 * 24)()
        flops = (C.c_double * 24)()
        calls = (C.c_int64 * 24)()
        self._chk(self.lib.capdec_profile_get(self._h, C.byref(cnt), names, ms, launches, flops, calls), "profile_get")
        return {names[i].decode(): dict(ms=float(ms[i]), launches=int(launches[i]), flops=float(flops[i]),
                                        calls=int(calls[i]))
                for i in range(cnt.value)}


_engines: Dict[int, Engine] = {}


def get_engine(device: int = 0) -> Engine:
    """process-wide engine per device index"""
    if device not in _engines:
        _engines[device] = Engine(device)
    return _engines[device]
