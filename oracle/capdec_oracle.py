"""CPU oracle for the CapDec caption hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this file; ``capdec_amd`` never does (the product path is HIP-only and raises when
the extension is missing).

It is a plain-torch (CPU, fp32) restatement of the reference algorithm, function by
function, each citing the reference ``file:line`` it follows.  Two families:

* reference-shaped (batch 1, NO KV cache, lm_head on every position) -- ``generate2_ref`` /
  ``generate_beam_ref``: the algorithm of reference gpt2_prefix_eval.py:50-198 exactly as
  the reference executes it; this is what ``cpu_baseline`` times.
* KV-cached, batched -- ``greedy_cached`` / ``beam_cached``: the algorithm the HIP
  kernels implement; same results, O(T) instead of O(T^2); used for larger parity cases.

Pinning: the reference has no tests or golden vectors of its own, and
the GPT-2 arithmetic lives in the un-vendored dependency ``transformers`` (pinned 4.24.0 in
reference requirments.txt:12; 5.15.0 installed here).  The oracle is therefore pinned
against outputs of the reference itself, imported in the build container by
``tools/gen_golden.py`` -> ``tests/golden/*.npz`` (checked by
``tests/test_oracle_vs_golden.py``).

CLIP (the reference's dependency ``clip`` = openai/CLIP, NOT installed here): the towers are restated from the published
model under OpenAI state-dict names -- PARITY UNPINNED against the package itself.  What pins them instead: the ViT-B/32
text / image towers against ``transformers.CLIPModel`` (fp32 and fp16; ``tests/golden/clip_{tiny,b32}.npz``), the
ModifiedResNet (RN50x4) tower against a ``torch.nn`` module witness built to the published structure into which the
weights load with ``strict=True`` (``tools/gen_golden.py:_RnTower`` -> ``tests/golden/clip_resnet.npz``).
"""
from __future__ import annotations

import contextlib
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ----------------------------------------------------------------------------------------
# prefix stage
# ----------------------------------------------------------------------------------------
def normalize_prefix(prefix: Tensor, offset: Optional[Tensor] = None, dont_normalize: bool = False) -> Tensor:
    """reference predictions_runner.py:221-224: ``prefix / prefix.norm(2, -1)`` (no eps)
    then ``+ offset_to_add_in_inference``.  Written for B == 1 in the reference; batched here
    with keepdim."""
    if not dont_normalize:
        prefix = prefix / prefix.norm(2, -1, keepdim=True)
    if offset is not None:
        prefix = prefix + offset
    return prefix


def uniform_ball_noise(shape, radius: float, gauss: Tensor, u: Tensor) -> Tensor:
    """reference train.py:18-24 with the two random draws supplied by the caller
    (``gauss`` ~ randn(shape), ``u`` ~ rand(shape[0]))."""
    sphere = F.normalize(gauss, dim=1)
    u = u ** (1.0 / shape[1])
    return (sphere.T * u * radius).T


def noise_injection(x: Tensor, variance: float = 0.001, modality_offset: Optional[Tensor] = None,
                    uniform_noise: bool = False, dont_norm: bool = False, *,
                    noise: Optional[Tensor] = None, u: Optional[Tensor] = None) -> Tensor:
    """reference train.py:27-39.  ``variance == 0`` returns x UNCHANGED (not normalised).
    ``noise`` stands for the ``torch.randn(x.shape)`` draw (unit variance, scaled by std here),
    ``u`` for the ``torch.rand`` draw of the uniform-ball variant."""
    if variance == 0.0:
        return x
    std = math.sqrt(variance)
    if not dont_norm:
        x = F.normalize(x, dim=1)
    if noise is None:
        noise = torch.randn(x.shape)
    if uniform_noise:
        if u is None:
            u = torch.rand(x.shape[0])
        x = x + uniform_ball_noise(x.shape, std, noise, u)
    else:
        x = x + noise * std
    if modality_offset is not None:
        x = x + modality_offset
    return F.normalize(x, dim=1)


# ----------------------------------------------------------------------------------------
# mapping networks
# ----------------------------------------------------------------------------------------
def mlp_mapper(x: Tensor, sd: SD, pfx: str = "clip_project.") -> Tensor:
    """reference gpt2_prefix.py:114-126 built at :167-168: Linear -> Tanh -> Linear.
    x [B, D] -> [B, P*768]."""
    h = torch.tanh(F.linear(x, sd[pfx + "model.0.weight"], sd[pfx + "model.0.bias"]))
    return F.linear(h, sd[pfx + "model.2.weight"], sd[pfx + "model.2.bias"])


def _mapper_attention(x: Tensor, sd: SD, l: str, num_heads: int = 8) -> Tensor:
    """reference transformer_mapper.py:22-51 (self-attention, no mask, q/kv bias=False)."""
    b, n, c = x.shape
    hd = c // num_heads
    q = F.linear(x, sd[l + "attn.to_queries.weight"]).reshape(b, n, num_heads, hd)
    kv = F.linear(x, sd[l + "attn.to_keys_values.weight"]).reshape(b, n, 2, num_heads, hd)
    k, v = kv[:, :, 0], kv[:, :, 1]
    att = torch.einsum("bnhd,bmhd->bnmh", q, k) * (hd ** -0.5)
    att = att.softmax(dim=2)
    out = torch.einsum("bnmh,bmhd->bnhd", att, v).reshape(b, n, c)
    return F.linear(out, sd[l + "attn.project.weight"], sd[l + "attn.project.bias"])


def transformer_mapper(x: Tensor, sd: SD, clip_length: int, num_layers: int = 8,
                       pfx: str = "clip_project.") -> Tensor:
    """reference transformer_mapper.py:113-127 -> Transformer :76-110 -> TransformerLayer :54-73.
    x [B, D] -> [B, P, 768] (rows clip_length: of the sequence)."""
    d = sd[pfx + "prefix_const"].shape[1]
    h = F.linear(x, sd[pfx + "linear.weight"], sd[pfx + "linear.bias"]).view(x.shape[0], clip_length, d)
    const = sd[pfx + "prefix_const"].unsqueeze(0).expand(x.shape[0], -1, -1)
    h = torch.cat((h, const), dim=1)
    for i in range(num_layers):
        l = f"{pfx}transformer.layers.{i}."
        a = F.layer_norm(h, (d,), sd[l + "norm1.weight"], sd[l + "norm1.bias"], 1e-5)
        h = h + _mapper_attention(a, sd, l)
        m = F.layer_norm(h, (d,), sd[l + "norm2.weight"], sd[l + "norm2.bias"], 1e-5)
        m = F.linear(torch.relu(F.linear(m, sd[l + "mlp.fc1.weight"], sd[l + "mlp.fc1.bias"])),
                     sd[l + "mlp.fc2.weight"], sd[l + "mlp.fc2.bias"])
        h = h + m
    return h[:, clip_length:]


def clip_project(x: Tensor, sd: SD, mapping_type: str, prefix_length: int, clip_length: int = 10,
                 num_layers: int = 8) -> Tensor:
    """``model.clip_project(prefix).reshape(B, P, -1)`` (reference predictions_runner.py:228)."""
    if mapping_type == "mlp":
        return mlp_mapper(x, sd).reshape(x.shape[0], prefix_length, -1)
    return transformer_mapper(x, sd, clip_length, num_layers).reshape(x.shape[0], prefix_length, -1)


# ----------------------------------------------------------------------------------------
# GPT-2 (third-party transformers.GPT2LMHeadModel; formulae per SURVEY.md section 3.4)
# ----------------------------------------------------------------------------------------
def gelu_new(x: Tensor) -> Tensor:
    """transformers activations.py:65-66 (NewGELUActivation)."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


# bf16 mode (BASELINE configs[1]; include/capdec.h gemm mode 2): inside ``bf16_gemm_operands()`` both operands of every
# GPT-2 projection and of the lm_head are rounded to bf16 (round-to-nearest-even) before an fp32 matmul, and K / V are
# rounded to bf16 when they are produced (the bf16 KV cache: the rounded value is what the producing step and every
# later step attend to) -- the arithmetic of the HIP bf16 mode (one bf16 MFMA per product, fp32 accumulate; residual
# stream, LayerNorm, softmax fp32).  ``bf16_gemm_operands(dtype=torch.float16)`` is the fp16-operand variant (CLIP
# towers; no KV cache there).  Default: no rounding, the reference's fp32 path.
_GEMM_BF16 = False
_GEMM_DT = torch.bfloat16
_RW_CACHE: dict = {}       # id(weight) -> (weight, rounded weight): a GPT-2-small step would otherwise re-round 124 M values


@contextlib.contextmanager
def bf16_gemm_operands(dtype=torch.bfloat16):
    global _GEMM_BF16, _GEMM_DT
    old, old_dt = _GEMM_BF16, _GEMM_DT
    _GEMM_BF16, _GEMM_DT = True, dtype
    _RW_CACHE.clear()
    try:
        yield
    finally:
        _GEMM_BF16, _GEMM_DT = old, old_dt
        _RW_CACHE.clear()


def _r(x: Tensor) -> Tensor:
    """GEMM-input activation (and K / V at cache-write time): rounded to 16 bits inside ``bf16_gemm_operands()``"""
    return x.to(_GEMM_DT).to(torch.float32) if _GEMM_BF16 else x


def _rw(w: Tensor) -> Tensor:
    """weight operand: as ``_r`` but cached for the lifetime of the context (the key keeps the tensor alive, so an id
    cannot be recycled)"""
    if not _GEMM_BF16:
        return w
    hit = _RW_CACHE.get(id(w))
    if hit is None or hit[0] is not w:
        hit = (w, w.to(_GEMM_DT).to(torch.float32))
        _RW_CACHE[id(w)] = hit
    return hit[1]


def _n_layer(sd: SD, g: str) -> int:
    n = 0
    while f"{g}transformer.h.{n}.ln_1.weight" in sd:
        n += 1
    return n


def gpt2_hidden(embeds: Tensor, sd: SD, n_head: int = 12, g: str = "gpt.", pos0: int = 0,
                cache: Optional[list] = None) -> Tensor:
    """GPT2Model.forward on ``inputs_embeds`` [N, L, d]: h = x + wpe[pos]; 12 x
    {h += c_proj(causal MHA(c_attn(ln_1 h))); h += mlp.c_proj(gelu_new(c_fc(ln_2 h)))}; ln_f.
    Conv1D = addmm(bias, x, W) with W [in, out] (transformers pytorch_utils.py:117-121).
    With ``cache`` (list of per-layer [K, V] tensors [N, heads, L_past, hd]) the new rows sit
    at positions pos0.. and attend to the cached past plus themselves (causal)."""
    N, L, d = embeds.shape
    hd = d // n_head
    t = g + "transformer."
    h = embeds + sd[t + "wpe.weight"][pos0:pos0 + L]
    for i in range(_n_layer(sd, g)):
        b = f"{t}h.{i}."
        a = F.layer_norm(h, (d,), sd[b + "ln_1.weight"], sd[b + "ln_1.bias"], 1e-5)
        qkv = torch.addmm(sd[b + "attn.c_attn.bias"], _r(a.reshape(-1, d)), _rw(sd[b + "attn.c_attn.weight"])).view(N, L, 3 * d)
        q, k, v = qkv.split(d, dim=2)
        q = q.view(N, L, n_head, hd).transpose(1, 2)
        k = _r(k.view(N, L, n_head, hd).transpose(1, 2))          # bf16 mode: K / V live in bf16 from the moment they exist
        v = _r(v.view(N, L, n_head, hd).transpose(1, 2))
        if cache is not None:
            if cache[i] is not None:
                k = torch.cat((cache[i][0], k), dim=2)
                v = torch.cat((cache[i][1], v), dim=2)
            cache[i] = [k, v]
        Lk = k.shape[2]
        w = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd)
        causal = torch.ones(Lk, Lk, dtype=torch.bool).tril()[Lk - L:, :]
        w = torch.where(causal, w, torch.full((), torch.finfo(w.dtype).min))
        w = w.softmax(dim=-1)
        o = torch.matmul(w, v).transpose(1, 2).reshape(N, L, d)
        o = torch.addmm(sd[b + "attn.c_proj.bias"], _r(o.reshape(-1, d)), _rw(sd[b + "attn.c_proj.weight"])).view(N, L, d)
        h = h + o
        m = F.layer_norm(h, (d,), sd[b + "ln_2.weight"], sd[b + "ln_2.bias"], 1e-5)
        m = torch.addmm(sd[b + "mlp.c_fc.bias"], _r(m.reshape(-1, d)), _rw(sd[b + "mlp.c_fc.weight"]))
        m = gelu_new(m)
        m = torch.addmm(sd[b + "mlp.c_proj.bias"], _r(m), _rw(sd[b + "mlp.c_proj.weight"])).view(N, L, d)
        h = h + m
    return F.layer_norm(h, (d,), sd[t + "ln_f.weight"], sd[t + "ln_f.bias"], 1e-5)


def gpt2_logits(embeds: Tensor, sd: SD, n_head: int = 12, g: str = "gpt.") -> Tensor:
    """``model.gpt(inputs_embeds=x).logits`` -- ALL positions [N, L, V], tied lm_head
    (what the reference computes every step, gpt2_prefix_eval.py:76-77,163-164)."""
    h = gpt2_hidden(embeds, sd, n_head, g)
    return torch.matmul(_r(h), _rw(sd[g + "transformer.wte.weight"]).t())


def wte(ids: Tensor, sd: SD, g: str = "gpt.") -> Tensor:
    return sd[g + "transformer.wte.weight"][ids]


def train_forward(sd: SD, tokens: Tensor, prefix: Tensor, mapping_type: str, prefix_length: int, clip_length: int = 10,
                  num_layers: int = 8, n_head: int = 12) -> Tensor:
    """ClipCaptionModel.forward of the train step (reference train.py:251-260): logits [B, P + L, V] of
    cat(clip_project(prefix).view(-1, P, d), wte(tokens)).  The dataset pads on the right (train.py:52-63), so under the
    causal mask the attention_mask only changes the PADDED positions, which the loss ignores (train.py:349)."""
    pe = clip_project(prefix, sd, mapping_type, prefix_length, clip_length, num_layers)
    return gpt2_logits(torch.cat((pe, wte(tokens.long(), sd)), dim=1), sd, n_head)


# ----------------------------------------------------------------------------------------
# the train step with a frozen GPT-2 (reference train.py:344-354 run with --only_prefix: ClipCaptionPrefix,
# :279-287 -- parameters() = the mapper's, GPT-2 in eval mode, so no dropout anywhere and the step is deterministic)
# ----------------------------------------------------------------------------------------
# Backward passes are written out by hand (no autograd): this is the algorithm the HIP train step implements, and
# tests/test_oracle_vs_golden.py checks it against gradients the reference's own loss.backward() produced
# (tools/gen_golden.py:gen_train_step -> tests/golden/train_step_*.npz).
def _ln_fwd(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def _ln_bwd(x: Tensor, w: Tensor, dy: Tensor, eps: float = 1e-5) -> Tuple[Tensor, Tensor, Tensor]:
    """LayerNorm backward: (dx, dw, db) for y = xhat * w + b, xhat = (x - mean) * rstd (biased variance)"""
    mu = x.mean(-1, keepdim=True)
    rstd = torch.rsqrt(((x - mu) ** 2).mean(-1, keepdim=True) + eps)
    xhat = (x - mu) * rstd
    g = dy * w
    dx = rstd * (g - g.mean(-1, keepdim=True) - xhat * (g * xhat).mean(-1, keepdim=True))
    red = tuple(range(x.dim() - 1))
    return dx, (dy * xhat).sum(red), dy.sum(red)


def gelu_new_grad(x: Tensor) -> Tensor:
    """d/dx of transformers activations.py:65-66"""
    c = math.sqrt(2.0 / math.pi)
    t = torch.tanh(c * (x + 0.044715 * x ** 3))
    return 0.5 * (1.0 + t) + 0.5 * x * (1.0 - t * t) * c * (1.0 + 3.0 * 0.044715 * x * x)


def dropout_sites(n_layer: int, N: int, L: int, d: int, n_head: int) -> List[Tuple[str, Tuple[int, ...]]]:
    """The dropout calls of one GPT-2 forward in train() mode, in call order (transformers GPT2Model: ``drop`` after
    inputs_embeds + position_embeds; per block ``attn_dropout`` on the softmax weights, ``resid_dropout`` after attn.c_proj,
    the MLP's ``dropout`` after mlp.c_proj): (name, mask shape).  The mask stream of capdec_train_set_dropout_masks is
    these masks flattened and concatenated in this order, one byte per element (1 = keep)."""
    sites = [("embd", (N, L, d))]
    for i in range(n_layer):
        sites += [(f"h.{i}.attn", (N, n_head, L, L)), (f"h.{i}.resid", (N, L, d)), (f"h.{i}.mlp", (N, L, d))]
    return sites


def gpt2_forward_saved(embeds: Tensor, sd: SD, n_head: int = 12, g: str = "gpt.",
                       drop: Optional[Tuple[float, Sequence[Tensor]]] = None) -> Tuple[Tensor, list]:
    """gpt2_hidden (above) keeping what the backward pass needs per layer; returns (ln_f output, saved).
    ``drop`` = (p, masks): GPT-2 in train() mode (the reference's default run, train.py:306-308 + model.train() at :321) --
    every nn.Dropout / F.dropout of the stack multiplies by ``mask / (1 - p)`` with the keep-masks of ``dropout_sites``
    in call order (torch's dropout: Bernoulli(1 - p) keep mask, survivors scaled by 1 / (1 - p))."""
    N, L, d = embeds.shape
    hd = d // n_head
    t = g + "transformer."
    mi = iter(drop[1]) if drop is not None else None
    keep = 1.0 - drop[0] if drop is not None else 1.0
    dm = (lambda x: x * (next(mi).to(x.dtype) / keep)) if drop is not None else (lambda x: x)      # noqa: E731
    h = dm(embeds + sd[t + "wpe.weight"][:L])
    causal = torch.ones(L, L, dtype=torch.bool).tril()
    saved = []
    for i in range(_n_layer(sd, g)):
        b = f"{t}h.{i}."
        a1 = _ln_fwd(h, sd[b + "ln_1.weight"], sd[b + "ln_1.bias"])
        qkv = torch.addmm(sd[b + "attn.c_attn.bias"], a1.reshape(-1, d), sd[b + "attn.c_attn.weight"]).view(N, L, 3 * d)
        q, k, v = (x.view(N, L, n_head, hd).transpose(1, 2) for x in qkv.split(d, dim=2))
        w = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd)
        w = torch.where(causal, w, torch.full((), torch.finfo(w.dtype).min, dtype=w.dtype)).softmax(dim=-1)
        m_att = (next(mi).to(w.dtype) / keep) if drop is not None else None
        wd = w * m_att if drop is not None else w                   # the weights that multiply V
        att = torch.matmul(wd, v).transpose(1, 2).reshape(N, L, d)
        m_res = (next(mi).to(w.dtype) / keep) if drop is not None else None
        y = torch.addmm(sd[b + "attn.c_proj.bias"], att.reshape(-1, d), sd[b + "attn.c_proj.weight"]).view(N, L, d)
        h_mid = h + (y * m_res if drop is not None else y)
        a2 = _ln_fwd(h_mid, sd[b + "ln_2.weight"], sd[b + "ln_2.bias"])
        fc = torch.addmm(sd[b + "mlp.c_fc.bias"], a2.reshape(-1, d), sd[b + "mlp.c_fc.weight"])
        m_mlp = (next(mi).to(w.dtype) / keep) if drop is not None else None
        y2 = torch.addmm(sd[b + "mlp.c_proj.bias"], gelu_new(fc), sd[b + "mlp.c_proj.weight"]).view(N, L, d)
        h_out = h_mid + (y2 * m_mlp if drop is not None else y2)
        saved.append(dict(h=h, q=q, k=k, v=v, w=w, wd=wd, h_mid=h_mid, fc=fc, m_att=m_att, m_res=m_res, m_mlp=m_mlp))
        h = h_out
    saved.append(dict(h=h, m_embd=(drop[1][0].to(embeds.dtype) / keep) if drop is not None else None))
    return _ln_fwd(h, sd[t + "ln_f.weight"], sd[t + "ln_f.bias"]), saved


def gpt2_backward_dx(dhf: Tensor, saved: list, sd: SD, n_head: int = 12, g: str = "gpt.",
                     wgrads: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """d loss / d inputs_embeds given d loss / d (ln_f output).  With a frozen GPT-2 (``wgrads`` None) no weight gradient
    is formed; with ``wgrads`` (a dict) the gradients of every tensor of the block stack, ln_f and wpe are stored there
    under their state-dict names (the reference's default run trains them too, train.py:326).
    Conv1D y = x W + b with W [in, out]  =>  dx = dy W^T, dW = x^T dy, db = sum dy."""
    t = g + "transformer."
    N, L, d = dhf.shape
    hd = d // n_head
    f2 = lambda x: x.reshape(-1, x.shape[-1])                       # noqa: E731
    dh, gw, gb = _ln_bwd(saved[-1]["h"], sd[t + "ln_f.weight"], dhf)
    if wgrads is not None:
        wgrads[t + "ln_f.weight"], wgrads[t + "ln_f.bias"] = gw, gb
    for i in reversed(range(_n_layer(sd, g))):
        b, s = f"{t}h.{i}.", saved[i]
        dy2 = dh * s["m_mlp"] if s.get("m_mlp") is not None else dh          # through the MLP's dropout
        dg = dy2.reshape(-1, d) @ sd[b + "mlp.c_proj.weight"].t()
        dfc = dg * gelu_new_grad(s["fc"])
        da2 = (dfc @ sd[b + "mlp.c_fc.weight"].t()).view(N, L, d)
        dx2, g2w, g2b = _ln_bwd(s["h_mid"], sd[b + "ln_2.weight"], da2)
        dh_mid = dh + dx2
        dy1 = dh_mid * s["m_res"] if s.get("m_res") is not None else dh_mid  # through resid_dropout
        datt_m = dy1.reshape(-1, d) @ sd[b + "attn.c_proj.weight"].t()
        datt = datt_m.view(N, L, n_head, hd).transpose(1, 2)
        # softmax attention backward: dV = Pd^T dO (Pd = the dropped weights); dPd = dO V^T; dP = dPd mask / keep;
        # dS = P (dP - rowsum(P dP)); dQ = dS K / sqrt(hd); dK = dS^T Q / sqrt(hd)
        dv = torch.matmul(s.get("wd", s["w"]).transpose(-1, -2), datt)
        dp = torch.matmul(datt, s["v"].transpose(-1, -2))
        if s.get("m_att") is not None:
            dp = dp * s["m_att"]
        ds = s["w"] * (dp - (s["w"] * dp).sum(-1, keepdim=True))
        dq = torch.matmul(ds, s["k"]) / math.sqrt(hd)
        dk = torch.matmul(ds.transpose(-1, -2), s["q"]) / math.sqrt(hd)
        dqkv = torch.cat([x.transpose(1, 2).reshape(N, L, d) for x in (dq, dk, dv)], dim=2)
        da1 = (dqkv.reshape(-1, 3 * d) @ sd[b + "attn.c_attn.weight"].t()).view(N, L, d)
        dx1, g1w, g1b = _ln_bwd(s["h"], sd[b + "ln_1.weight"], da1)
        if wgrads is not None:
            a1 = _ln_fwd(s["h"], sd[b + "ln_1.weight"], sd[b + "ln_1.bias"])
            a2 = _ln_fwd(s["h_mid"], sd[b + "ln_2.weight"], sd[b + "ln_2.bias"])
            att = torch.matmul(s.get("wd", s["w"]), s["v"]).transpose(1, 2).reshape(N * L, d)
            wgrads[b + "mlp.c_proj.weight"], wgrads[b + "mlp.c_proj.bias"] = gelu_new(s["fc"]).t() @ f2(dy2), f2(dy2).sum(0)
            wgrads[b + "mlp.c_fc.weight"], wgrads[b + "mlp.c_fc.bias"] = f2(a2).t() @ dfc, dfc.sum(0)
            wgrads[b + "ln_2.weight"], wgrads[b + "ln_2.bias"] = g2w, g2b
            wgrads[b + "attn.c_proj.weight"], wgrads[b + "attn.c_proj.bias"] = att.t() @ f2(dy1), f2(dy1).sum(0)
            wgrads[b + "attn.c_attn.weight"], wgrads[b + "attn.c_attn.bias"] = f2(a1).t() @ f2(dqkv), f2(dqkv).sum(0)
            wgrads[b + "ln_1.weight"], wgrads[b + "ln_1.bias"] = g1w, g1b
        dh = dh_mid + dx1
    if saved[-1].get("m_embd") is not None:
        dh = dh * saved[-1]["m_embd"]                                  # through the embedding dropout
    if wgrads is not None:
        wgrads[t + "wpe.weight"] = torch.zeros_like(sd[t + "wpe.weight"])
        wgrads[t + "wpe.weight"][:L] = dh.sum(0)
    return dh                                          # (frozen wpe: the position term only adds a constant)


def mlp_mapper_backward(x: Tensor, dy: Tensor, sd: SD, pfx: str = "clip_project.") -> Dict[str, Tensor]:
    """gradients of the MLP mapper's four tensors (reference gpt2_prefix.py:114-126: Linear -> Tanh -> Linear)"""
    hmid = torch.tanh(F.linear(x, sd[pfx + "model.0.weight"], sd[pfx + "model.0.bias"]))
    dh = dy @ sd[pfx + "model.2.weight"]
    da = dh * (1.0 - hmid * hmid)
    return {pfx + "model.0.weight": da.t() @ x, pfx + "model.0.bias": da.sum(0),
            pfx + "model.2.weight": dy.t() @ hmid, pfx + "model.2.bias": dy.sum(0)}


def transformer_mapper_backward(x: Tensor, dout: Tensor, sd: SD, clip_length: int, num_layers: int = 8,
                                pfx: str = "clip_project.", num_heads: int = 8) -> Dict[str, Tensor]:
    """gradients of every tensor of the TransformerMapper (reference transformer_mapper.py:113-127 -> :76-110 -> :54-73 ->
    :22-51, :4-19) given d loss / d output [B, P, d]; forward as ``transformer_mapper`` above, kept per layer"""
    d = sd[pfx + "prefix_const"].shape[1]
    B, hd = x.shape[0], d // num_heads
    h = torch.cat((F.linear(x, sd[pfx + "linear.weight"], sd[pfx + "linear.bias"]).view(B, clip_length, d),
                   sd[pfx + "prefix_const"].unsqueeze(0).expand(B, -1, -1)), dim=1)
    S = h.shape[1]
    saved = []
    for i in range(num_layers):
        l = f"{pfx}transformer.layers.{i}."
        a1 = _ln_fwd(h, sd[l + "norm1.weight"], sd[l + "norm1.bias"])
        q = F.linear(a1, sd[l + "attn.to_queries.weight"]).reshape(B, S, num_heads, hd)
        kv = F.linear(a1, sd[l + "attn.to_keys_values.weight"]).reshape(B, S, 2, num_heads, hd)
        k, v = kv[:, :, 0], kv[:, :, 1]
        w = (torch.einsum("bnhd,bmhd->bnmh", q, k) * (hd ** -0.5)).softmax(dim=2)
        att = torch.einsum("bnmh,bmhd->bnhd", w, v).reshape(B, S, d)
        h_mid = h + F.linear(att, sd[l + "attn.project.weight"], sd[l + "attn.project.bias"])
        a2 = _ln_fwd(h_mid, sd[l + "norm2.weight"], sd[l + "norm2.bias"])
        r = torch.relu(F.linear(a2, sd[l + "mlp.fc1.weight"], sd[l + "mlp.fc1.bias"]))
        saved.append(dict(h=h, a1=a1, q=q, k=k, v=v, w=w, att=att, h_mid=h_mid, a2=a2, r=r))
        h = h_mid + F.linear(r, sd[l + "mlp.fc2.weight"], sd[l + "mlp.fc2.bias"])
    g: Dict[str, Tensor] = {}
    dh = torch.zeros_like(h)
    dh[:, clip_length:] = dout
    f2 = lambda t: t.reshape(-1, t.shape[-1])                       # noqa: E731
    for i in reversed(range(num_layers)):
        l, s = f"{pfx}transformer.layers.{i}.", saved[i]
        g[l + "mlp.fc2.weight"], g[l + "mlp.fc2.bias"] = f2(dh).t() @ f2(s["r"]), f2(dh).sum(0)
        dr = (dh @ sd[l + "mlp.fc2.weight"]) * (s["r"] > 0)
        g[l + "mlp.fc1.weight"], g[l + "mlp.fc1.bias"] = f2(dr).t() @ f2(s["a2"]), f2(dr).sum(0)
        dx, g[l + "norm2.weight"], g[l + "norm2.bias"] = _ln_bwd(s["h_mid"], sd[l + "norm2.weight"], dr @ sd[l + "mlp.fc1.weight"])
        dh_mid = dh + dx
        g[l + "attn.project.weight"], g[l + "attn.project.bias"] = f2(dh_mid).t() @ f2(s["att"]), f2(dh_mid).sum(0)
        datt = (dh_mid @ sd[l + "attn.project.weight"]).reshape(B, S, num_heads, hd)
        dv = torch.einsum("bnmh,bnhd->bmhd", s["w"], datt)
        dp = torch.einsum("bnhd,bmhd->bnmh", datt, s["v"])
        ds = s["w"] * (dp - (s["w"] * dp).sum(2, keepdim=True))
        dq = torch.einsum("bnmh,bmhd->bnhd", ds, s["k"]) * (hd ** -0.5)
        dk = torch.einsum("bnmh,bnhd->bmhd", ds, s["q"]) * (hd ** -0.5)
        dq2, dkv2 = dq.reshape(B * S, d), torch.stack((dk, dv), dim=2).reshape(B * S, 2 * d)
        g[l + "attn.to_queries.weight"] = dq2.t() @ f2(s["a1"])
        g[l + "attn.to_keys_values.weight"] = dkv2.t() @ f2(s["a1"])
        da1 = (dq2 @ sd[l + "attn.to_queries.weight"] + dkv2 @ sd[l + "attn.to_keys_values.weight"]).view(B, S, d)
        dx, g[l + "norm1.weight"], g[l + "norm1.bias"] = _ln_bwd(s["h"], sd[l + "norm1.weight"], da1)
        dh = dh_mid + dx
    g[pfx + "prefix_const"] = dh[:, clip_length:].sum(0)
    dlin = dh[:, :clip_length].reshape(B, clip_length * d)
    g[pfx + "linear.weight"], g[pfx + "linear.bias"] = dlin.t() @ x, dlin.sum(0)
    return g


def train_step_loss_and_grads(sd: SD, tokens: Tensor, prefix: Tensor, mapping_type: str, prefix_length: int,
                              n_head: int = 12, clip_length: int = 10, num_layers: int = 8,
                              train_gpt: bool = False,
                              drop: Optional[Tuple[float, Sequence[Tensor]]] = None) -> Tuple[Tensor, Dict[str, Tensor]]:
    """loss of reference train.py:348-349 (cross_entropy(logits[:, P-1:-1], tokens, ignore_index=0), mean over the
    labels != 0) and its gradients with respect to the mapper's parameters (what loss.backward() leaves in .grad of
    ClipCaptionPrefix.parameters(), :350).  ``prefix`` is the embedding batch AFTER noise_injection (:347).
    ``train_gpt``: the reference's DEFAULT run (ClipCaptionModel: ``model.parameters()`` includes GPT-2) -- the
    gradients of every GPT-2 tensor are returned too (the tied wte collects the lm_head's and the token lookup's).
    ``drop`` = (p, keep-masks in the order of ``dropout_sites``): GPT-2's dropouts (the reference trains with transformers'
    default 0.1: embd_pdrop = attn_pdrop = resid_pdrop) with the masks supplied -- how the fixture
    tests/golden/train_full_dropout_tiny.npz pins it against the reference's own loss.backward()."""
    P, (B, L) = prefix_length, tokens.shape
    d = sd["gpt.transformer.wte.weight"].shape[1]
    pe = clip_project(prefix, sd, mapping_type, P, clip_length, num_layers)
    embeds = torch.cat((pe, wte(tokens.long(), sd)), dim=1)
    hf, saved = gpt2_forward_saved(embeds, sd, n_head, drop=drop)
    W = sd["gpt.transformer.wte.weight"]
    logits = hf[:, P - 1:-1] @ W.t()                                   # [B, L, V]: the rows the loss reads
    labels = tokens.long()
    valid = labels != 0
    n = int(valid.sum())
    lse = torch.logsumexp(logits, -1)
    picked = logits.gather(-1, labels.unsqueeze(-1)).squeeze(-1)
    loss = ((lse - picked) * valid).sum() / n
    dlogits = torch.softmax(logits, -1)
    dlogits.scatter_add_(-1, labels.unsqueeze(-1), -torch.ones_like(picked).unsqueeze(-1))
    dlogits = dlogits * (valid.unsqueeze(-1) / n)
    dhf = torch.zeros_like(hf)
    dhf[:, P - 1:-1] = dlogits @ W
    grads: Dict[str, Tensor] = {}
    dembeds = gpt2_backward_dx(dhf, saved, sd, n_head, wgrads=grads if train_gpt else None)
    if train_gpt:
        gw = dlogits.reshape(-1, dlogits.shape[-1]).t() @ hf[:, P - 1:-1].reshape(-1, d)          # lm_head (tied)
        gw.index_add_(0, labels.reshape(-1), dembeds[:, P:].reshape(-1, d))                        # token lookup
        grads["gpt.transformer.wte.weight"] = gw
    if mapping_type == "mlp":
        grads.update(mlp_mapper_backward(prefix, dembeds[:, :P].reshape(B, P * d), sd))
    else:
        grads.update(transformer_mapper_backward(prefix, dembeds[:, :P], sd, clip_length, num_layers))
    return loss, grads


def linear_schedule_with_warmup(step: int, num_warmup_steps: int, num_training_steps: int) -> float:
    """the lr factor of transformers.get_linear_schedule_with_warmup (reference train.py:329-331) at scheduler step
    ``step``.  The reference calls optimizer.step() BEFORE scheduler.step() (:351-352), so the k-th update (k = 0, 1, ...)
    runs at factor(k): the very first update has lr 0."""
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))


def adamw_transformers(p: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, step: int, lr: float,
                       betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-6, weight_decay: float = 0.0,
                       correct_bias: bool = True) -> None:
    """One update of ``transformers.AdamW`` (the optimizer reference train.py:326 constructs: ``AdamW(model.parameters(),
    lr=args.lr)``; transformers pinned 4.24.0 in reference requirments.txt:12 -- the class no longer exists in the 5.15
    installed here, so this is a restatement of the published algorithm, transformers/optimization.py ``AdamW.step``;
    PARITY UNPINNED against the class itself).  Differs from torch.optim.AdamW: eps defaults to 1e-6 and is added to
    sqrt(v) BEFORE the bias correction is applied through the step size; weight decay (default 0) is applied after the
    update with the uncorrected lr.  In place; ``step`` = 1 for the first update."""
    b1, b2 = betas
    exp_avg.mul_(b1).add_(grad, alpha=1.0 - b1)
    exp_avg_sq.mul_(b2).addcmul_(grad, grad, value=1.0 - b2)
    denom = exp_avg_sq.sqrt().add_(eps)
    step_size = lr
    if correct_bias:
        step_size = step_size * math.sqrt(1.0 - b2 ** step) / (1.0 - b1 ** step)
    p.addcdiv_(exp_avg, denom, value=-step_size)
    if weight_decay > 0.0:
        p.add_(p, alpha=-lr * weight_decay)


def train_steps(sd: SD, batches: Sequence[Tuple[Tensor, Tensor]], mapping_type: str, prefix_length: int, lr: float,
                num_warmup_steps: int, num_training_steps: int, n_head: int = 12, clip_length: int = 10,
                num_layers: int = 8, train_gpt: bool = False,
                drops: Optional[Sequence[Tuple[float, Sequence[Tensor]]]] = None) -> Tuple[List[float], SD]:
    """``len(batches)`` iterations of reference train.py:344-354 (frozen GPT-2) from the weights ``sd`` (not modified):
    each batch = (tokens [B, L] right-padded with 0, prefix [B, D] after noise injection).  Returns the per-step losses
    and the final state dict."""
    sd = {k: (v.clone() if (train_gpt or k.startswith("clip_project.")) else v) for k, v in sd.items()}
    if train_gpt and "gpt.lm_head.weight" in sd:
        sd["gpt.lm_head.weight"] = sd["gpt.transformer.wte.weight"]                 # tied: one tensor
    state: Dict[str, Tuple[Tensor, Tensor]] = {}
    losses = []
    for it, (tokens, prefix) in enumerate(batches):
        loss, grads = train_step_loss_and_grads(sd, tokens, prefix, mapping_type, prefix_length, n_head, clip_length, num_layers,
                                                train_gpt=train_gpt, drop=drops[it] if drops is not None else None)
        losses.append(float(loss))
        cur_lr = lr * linear_schedule_with_warmup(it, num_warmup_steps, num_training_steps)
        for k, gk in grads.items():
            if k not in state:
                state[k] = (torch.zeros_like(sd[k]), torch.zeros_like(sd[k]))
            adamw_transformers(sd[k], gk, state[k][0], state[k][1], it + 1, cur_lr)
    return losses, sd


# ----------------------------------------------------------------------------------------
# reference-shaped decode (batch 1, no KV cache): what the reference executes
# ----------------------------------------------------------------------------------------
def generate2_ref(sd: SD, embed: Tensor, stop_id: int = 13, entry_length: int = 67, top_p: float = 0.8,
                  temperature: float = 1.0, alt_stop_id: int = 764, n_head: int = 12,
                  margins: Optional[list] = None, on_step=None) -> List[int]:
    """reference gpt2_prefix_eval.py:118-198 with ``embed`` [1, P, d] given.  Keeps the sort /
    cumsum / top-p masking the reference performs (the arg-max is unaffected: rank 0 is
    never removed, :172).  Returns the id list INCLUDING the stop token."""
    generated = embed
    tokens: List[int] = []
    for _ in range(entry_length):
        logits = gpt2_logits(generated, sd, n_head)[:, -1, :] / (temperature if temperature > 0 else 1.0)
        sorted_logits, sorted_indices = torch.sort(logits, descending=True)
        if margins is not None:
            margins.append(float(sorted_logits[0, 0] - sorted_logits[0, 1]))
        cumulative = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1)
        remove = cumulative > top_p
        remove[..., 1:] = remove[..., :-1].clone()
        remove[..., 0] = 0
        logits[:, sorted_indices[remove]] = -float("inf")
        nxt = torch.argmax(logits, -1).unsqueeze(0)
        generated = torch.cat((generated, wte(nxt, sd)), dim=1)
        tokens.append(int(nxt.item()))
        if tokens[-1] == stop_id or tokens[-1] == alt_stop_id:
            break
        if on_step is not None and on_step(len(tokens) - 1, 1, generated.shape[1] - 1):
            break
    return tokens


def generate_beam_ref(sd: SD, embed: Tensor, beam_size: int = 5, stop_id: int = 13, entry_length: int = 67,
                      temperature: float = 1.0, n_head: int = 12, on_step=None) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """reference gpt2_prefix_eval.py:50-115 with ``embed`` [1, P, d] given (``on_step`` as in
    generate2_ref).  Returns the
    function's final internal state: ``tokens`` int64 [beam, T], ``seq_lengths`` fp32 [beam],
    mean-log-prob ``scores`` fp32 [beam] (after ``scores / seq_lengths``, :110) and
    ``order = scores.argsort(descending=True)`` (:113); the reference returns
    ``decode(tokens[b, :int(seq_lengths[b])])`` for b in order."""
    tokens = None
    scores = None
    seq_lengths = torch.ones(beam_size)
    is_stopped = torch.zeros(beam_size, dtype=torch.bool)
    generated = embed
    for it in range(entry_length):
        rows_in, L_in = generated.shape[0], generated.shape[1]
        logits = gpt2_logits(generated, sd, n_head)[:, -1, :] / (temperature if temperature > 0 else 1.0)
        logits = logits.softmax(-1).log()
        if scores is None:
            scores, next_tokens = logits.topk(beam_size, -1)
            generated = generated.expand(beam_size, *generated.shape[1:])
            next_tokens, scores = next_tokens.permute(1, 0), scores.squeeze(0)
            tokens = next_tokens
        else:
            logits[is_stopped] = -float("inf")
            logits[is_stopped, 0] = 0
            scores_sum = scores[:, None] + logits
            seq_lengths[~is_stopped] += 1
            scores_sum_average = scores_sum / seq_lengths[:, None]
            scores_sum_average, next_tokens = scores_sum_average.view(-1).topk(beam_size, -1)
            source = next_tokens // scores_sum.shape[1]
            seq_lengths = seq_lengths[source]
            next_tokens = (next_tokens % scores_sum.shape[1]).unsqueeze(1)
            tokens = torch.cat((tokens[source], next_tokens), dim=1)
            generated = generated[source]
            scores = scores_sum_average * seq_lengths
            is_stopped = is_stopped[source]
        emb = wte(next_tokens.squeeze(), sd).view(generated.shape[0], 1, -1)
        generated = torch.cat((generated, emb), dim=1)
        is_stopped = is_stopped + next_tokens.eq(stop_id).squeeze()
        if is_stopped.all():
            break
        if on_step is not None and on_step(it, rows_in, L_in):
            break
    scores = scores / seq_lengths
    order = scores.argsort(descending=True)
    return tokens, seq_lengths, scores, order


# ----------------------------------------------------------------------------------------
# KV-cached batched decode: the algorithm the HIP kernels implement
# ----------------------------------------------------------------------------------------
def greedy_cached(sd: SD, prefix: Tensor, stop_id: int = 13, entry_length: int = 67, alt_stop_id: int = 764,
                  n_head: int = 12) -> Tuple[Tensor, Tensor]:
    """Batched greedy with a KV cache.  prefix [N, P, d] -> ids int32 [N, entry_length]
    (zero padded) and lens int32 [N] (count INCLUDING the stop token), equal row by row
    to ``generate2_ref``."""
    N, P, d = prefix.shape
    g = "gpt."
    cache: list = [None] * _n_layer(sd, g)
    W = sd[g + "transformer.wte.weight"]
    Wr = _rw(W)
    ids = torch.zeros(N, entry_length, dtype=torch.int32)
    lens = torch.zeros(N, dtype=torch.int32)
    done = torch.zeros(N, dtype=torch.bool)
    h = gpt2_hidden(prefix, sd, n_head, g, 0, cache)[:, -1]
    for i in range(entry_length):
        nxt = torch.argmax(_r(h) @ Wr.t(), -1)
        ids[~done, i] = nxt[~done].to(torch.int32)
        lens[~done] += 1
        done = done | (nxt == stop_id) | (nxt == alt_stop_id)
        if bool(done.all()) or i == entry_length - 1:
            break
        h = gpt2_hidden(W[nxt].unsqueeze(1), sd, n_head, g, P + i, cache)[:, -1]
    return ids, lens


def greedy_forced(sd: SD, prefix: Tensor, forced: Tensor, n_head: int = 12) -> Tuple[Tensor, Tensor]:
    """Teacher-forced KV-cached greedy decode (capdec_decode_greedy_forced): step i feeds ``forced[:, i]`` whatever the
    arg-max was.  -> (arg-max ids int32 [N, T], stats fp32 [N, T, 3] = top-1 logit, top-2 logit, logsumexp)."""
    N, P, d = prefix.shape
    T = forced.shape[1]
    g = "gpt."
    cache: list = [None] * _n_layer(sd, g)
    W = sd[g + "transformer.wte.weight"]
    Wr = _rw(W)
    ids = torch.zeros(N, T, dtype=torch.int32)
    stats = torch.zeros(N, T, 3)
    h = gpt2_hidden(prefix, sd, n_head, g, 0, cache)[:, -1]
    for i in range(T):
        logits = _r(h) @ Wr.t()
        top = logits.topk(2, -1)
        ids[:, i] = top.indices[:, 0].to(torch.int32)
        stats[:, i, 0], stats[:, i, 1] = top.values[:, 0], top.values[:, 1]
        stats[:, i, 2] = torch.logsumexp(logits, -1)
        if i == T - 1:
            break
        h = gpt2_hidden(W[forced[:, i].long()].unsqueeze(1), sd, n_head, g, P + i, cache)[:, -1]
    return ids, stats


def beam_cached(sd: SD, prefix: Tensor, beam_size: int = 5, stop_id: int = 13, entry_length: int = 67,
                temperature: float = 1.0, n_head: int = 12, margins: Optional[list] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """Batched beam search with a KV cache, per caption identical in arithmetic to
    ``generate_beam_ref`` (same fp32 op order for sum / mean / sum score juggling).
    prefix [N, P, d] -> tokens int32 [N, beam, entry_length] (zero padded), seq_lengths int32
    [N, beam], scores fp32 [N, beam]; rows in the reference's INTERNAL beam order -- sort by
    ``scores`` descending (stable) to get the returned order.
    ``margins`` (a list): receives one tensor [N] = the smallest gap, over the caption's live steps, between the
    last selected and the first rejected candidate key -- a caption whose margin is at fp32 round-off level
    (< ~1e-5) is a numerical tie: another summation order may legitimately keep a different beam."""
    N, P, d = prefix.shape
    g = "gpt."
    B = beam_size
    nl = _n_layer(sd, g)
    cache: list = [None] * nl
    W = sd[g + "transformer.wte.weight"]
    V = W.shape[0]
    temp = temperature if temperature > 0 else 1.0
    h = gpt2_hidden(prefix, sd, n_head, g, 0, cache)[:, -1]
    Wr = _rw(W)
    logp = ((_r(h) @ Wr.t()) / temp).softmax(-1).log()
    scores, nxt = logp.topk(B, -1)                      # [N, B]
    if margins is not None and V > B:
        t6 = logp.topk(B + 1, -1).values
        margin = t6[:, B - 1] - t6[:, B]
    else:
        margin = torch.full((N,), float("inf"))
    tokens = torch.zeros(N, B, entry_length, dtype=torch.int64)
    tokens[:, :, 0] = nxt
    seq = torch.ones(N, B)
    stopped = nxt.eq(stop_id)
    for i in range(nl):                                 # expand the prefix cache to beams
        cache[i] = [c.repeat_interleave(B, dim=0) for c in cache[i]]
    alive = ~stopped.all(dim=1)                         # captions whose loop has not broken
    for i in range(1, entry_length):
        if not bool(alive.any()):
            break
        x = W[nxt.reshape(-1)].unsqueeze(1)
        h = gpt2_hidden(x, sd, n_head, g, P + i - 1, cache)[:, -1]
        logp = ((_r(h) @ Wr.t()) / temp).softmax(-1).log().view(N, B, V)
        logp[stopped] = -float("inf")
        logp[stopped, 0] = 0
        ssum = scores[:, :, None] + logp
        seq_new = seq + (~stopped).float()
        avg = ssum / seq_new[:, :, None]
        avg_top, flat = avg.view(N, -1).topk(B, -1)
        if margins is not None:
            t6 = avg.view(N, -1).topk(B + 1, -1).values
            gap = t6[:, B - 1] - t6[:, B]
            margin = torch.where(alive & (gap < margin), gap, margin)
        src = flat // V
        tok = flat % V
        seq_sel = torch.gather(seq_new, 1, src)
        tok_hist = torch.gather(tokens, 1, src[:, :, None].expand(-1, -1, entry_length)).clone()
        tok_hist[:, :, i] = tok
        stopped_sel = torch.gather(stopped, 1, src) | tok.eq(stop_id)
        # captions that already broke out of the reference loop keep their state
        a = alive
        tokens[a] = tok_hist[a]
        seq[a] = seq_sel[a]
        scores[a] = (avg_top * seq_sel)[a]
        stopped[a] = stopped_sel[a]
        nxt = torch.where(a[:, None], tok, nxt)
        rows = (torch.arange(N)[:, None] * B + torch.where(a[:, None], src, torch.arange(B)[None, :])).reshape(-1)
        for l in range(nl):
            cache[l] = [c[rows] for c in cache[l]]
        alive = alive & ~stopped.all(dim=1)
    final = scores / seq
    if margins is not None:
        margins.append(margin)
    return tokens.to(torch.int32), seq.to(torch.int32), final


def beam_output_order(scores: Tensor) -> Tensor:
    """``scores.argsort(descending=True)`` of reference gpt2_prefix_eval.py:113, per caption."""
    return scores.argsort(dim=-1, descending=True)


# ----------------------------------------------------------------------------------------
# CLIP ViT-B/32 towers (third-party `clip` = openai/CLIP, un-vendored and NOT installed here;
# reference call sites embeddings_generator.py:86,89, predictions_runner.py:218,220).  Restated
# from the published model (clip/model.py: Transformer / ResidualAttentionBlock / QuickGELU /
# VisionTransformer / CLIP.encode_text / encode_image) on OpenAI state-dict names.  Pinned only
# against the independent HF `CLIPModel` stand-in (tests/golden/clip_*.npz): PARITY UNPINNED
# with respect to the reference's own dependency.
# ----------------------------------------------------------------------------------------
def _clip_resblocks(x: Tensor, sd: SD, prefix: str, n_head: int, causal: bool) -> Tensor:
    N, L, d = x.shape
    hd = d // n_head
    i = 0
    while f"{prefix}transformer.resblocks.{i}.ln_1.weight" in sd:
        b = f"{prefix}transformer.resblocks.{i}."
        a = F.layer_norm(x, (d,), sd[b + "ln_1.weight"], sd[b + "ln_1.bias"], 1e-5)
        # (_r / _rw: 16-bit GEMM operands inside ``bf16_gemm_operands(dtype)`` -- the HIP f16 / bf16 tower modes)
        qkv = F.linear(_r(a), _rw(sd[b + "attn.in_proj_weight"]), sd[b + "attn.in_proj_bias"])
        q, k, v = qkv.split(d, dim=2)
        q = q.view(N, L, n_head, hd).transpose(1, 2)
        k = k.view(N, L, n_head, hd).transpose(1, 2)
        v = v.view(N, L, n_head, hd).transpose(1, 2)
        if _GEMM_BF16:
            # the towers' 16-bit modes (HIP attn_prefill_mfma_kernel<.., SPLIT = false>): both attention products take fp16
            # operands too -- q (scaled by 1 / sqrt(hd) and log2 e: the scores live in the log2 domain), k, v and the
            # UN-NORMALISED weights 2^(s - max) are rounded to fp16 (whatever the GEMM dtype), products and sums fp32, the
            # row sum is taken over the unrounded weights and divides the result
            h16 = lambda t: t.to(torch.float16).to(torch.float32)         # noqa: E731
            w = torch.matmul(h16(q * (hd ** -0.5 * 1.4426950408889634)), h16(k).transpose(-1, -2))
            if causal:
                w = w + torch.full((L, L), float("-inf")).triu_(1)
            pw = torch.exp2(w - w.max(dim=-1, keepdim=True).values)
            o = (torch.matmul(h16(pw), h16(v)) / pw.sum(dim=-1, keepdim=True)).transpose(1, 2).reshape(N, L, d)
        else:
            q = q * (hd ** -0.5)                                          # nn.MultiheadAttention scales q
            w = torch.matmul(q, k.transpose(-1, -2))
            if causal:                                                    # build_attention_mask: -inf above the diagonal
                w = w + torch.full((L, L), float("-inf")).triu_(1)
            o = torch.matmul(w.softmax(dim=-1), v).transpose(1, 2).reshape(N, L, d)
        x = x + F.linear(_r(o), _rw(sd[b + "attn.out_proj.weight"]), sd[b + "attn.out_proj.bias"])
        m = F.layer_norm(x, (d,), sd[b + "ln_2.weight"], sd[b + "ln_2.bias"], 1e-5)
        m = F.linear(_r(m), _rw(sd[b + "mlp.c_fc.weight"]), sd[b + "mlp.c_fc.bias"])
        m = m * torch.sigmoid(1.702 * m)                                  # QuickGELU
        x = x + F.linear(_r(m), _rw(sd[b + "mlp.c_proj.weight"]), sd[b + "mlp.c_proj.bias"])
        i += 1
    return x


def clip_encode_text(text: Tensor, sd: SD, n_head: int = 8) -> Tensor:
    """CLIP.encode_text: token ids int [N, 77] -> [N, embed_dim] (NOT normalised, like
    reference embeddings_generator.py:86-87)."""
    x = sd["token_embedding.weight"][text] + sd["positional_embedding"]
    x = _clip_resblocks(x, sd, "", n_head, causal=True)
    d = x.shape[-1]
    x = F.layer_norm(x, (d,), sd["ln_final.weight"], sd["ln_final.bias"], 1e-5)
    x = x[torch.arange(x.shape[0]), text.argmax(dim=-1)]                  # EOT token = highest id
    return x @ sd["text_projection"]


def clip_encode_image(image: Tensor, sd: SD, n_head: int = 12) -> Tensor:
    """CLIP.encode_image (VisionTransformer): pixels fp32 [N, 3, 224, 224] -> [N, embed_dim]."""
    w = sd["visual.conv1.weight"]
    x = F.conv2d(image, w, stride=w.shape[-1])                            # [N, width, 7, 7]
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)            # [N, 49, width]
    cls = sd["visual.class_embedding"].expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"]
    d = x.shape[-1]
    x = F.layer_norm(x, (d,), sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"], 1e-5)
    x = _clip_resblocks(x, sd, "visual.", n_head, causal=False)
    x = F.layer_norm(x[:, 0, :], (d,), sd["visual.ln_post.weight"], sd["visual.ln_post.bias"], 1e-5)
    return x @ sd["visual.proj"]


# ----------------------------------------------------------------------------------------
# CLIP ModifiedResNet image tower (`clip.load("RN50x4")`: the reference's default backbone, predictions_runner.py:158,
# 220; embeddings_generator.py:89,113).  The `clip` package is not installed and NO stand-in for this architecture
# exists in the container (transformers has no ModifiedResNet): the functions below restate the published openai/CLIP
# model (clip/model.py: Bottleneck, AttentionPool2d, ModifiedResNet) -- PARITY UNPINNED against the package; pinned against
# the torch.nn module witness of tools/gen_golden.py (tests/golden/clip_resnet.npz), see DESIGN.md section 2.
#   stem: 3 x (conv3x3 -> BatchNorm -> ReLU) with the first conv at stride 2, then AvgPool2d(2);
#   4 stages of Bottleneck blocks (1x1 -> 3x3 -> [AvgPool2d(stride)] -> 1x1 (x4 planes), BatchNorm after every conv,
#   anti-aliased striding: the stride is an average pool, never a strided conv; downsample = AvgPool2d(stride) ->
#   conv1x1 -> BatchNorm whenever stride > 1 or the channel count changes; out = ReLU(out + identity));
#   attention pool: tokens = [mean token; HW tokens] + positional embedding, ONE query (the mean token) attends over
#   all tokens with nn.MultiheadAttention semantics (separate q/k/v projections with bias, q scaled by head_dim^-0.5),
#   c_proj -> output_dim.
# ----------------------------------------------------------------------------------------
def _bn2d(x: Tensor, sd: SD, p: str) -> Tensor:
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False, 0.0, 1e-5)


def _rn_bottleneck(x: Tensor, sd: SD, p: str, stride: int) -> Tensor:
    out = F.relu(_bn2d(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1."))
    out = F.relu(_bn2d(F.conv2d(out, sd[p + "conv2.weight"], padding=1), sd, p + "bn2."))
    if stride > 1:
        out = F.avg_pool2d(out, stride)
    out = _bn2d(F.conv2d(out, sd[p + "conv3.weight"]), sd, p + "bn3.")
    if p + "downsample.0.weight" in sd:
        idt = F.avg_pool2d(x, stride) if stride > 1 else x
        idt = _bn2d(F.conv2d(idt, sd[p + "downsample.0.weight"]), sd, p + "downsample.1.")
    else:
        idt = x
    return F.relu(out + idt)


def clip_resnet_features(image: Tensor, sd: SD) -> Tensor:
    """ModifiedResNet up to (not including) the attention pool: [N, 3, S, S] -> [N, 32 * width, S/32, S/32]."""
    v = "visual."
    x = F.relu(_bn2d(F.conv2d(image, sd[v + "conv1.weight"], stride=2, padding=1), sd, v + "bn1."))
    x = F.relu(_bn2d(F.conv2d(x, sd[v + "conv2.weight"], padding=1), sd, v + "bn2."))
    x = F.relu(_bn2d(F.conv2d(x, sd[v + "conv3.weight"], padding=1), sd, v + "bn3."))
    x = F.avg_pool2d(x, 2)
    for li in range(1, 5):
        b = 0
        while f"{v}layer{li}.{b}.conv1.weight" in sd:
            x = _rn_bottleneck(x, sd, f"{v}layer{li}.{b}.", 2 if (b == 0 and li > 1) else 1)
            b += 1
    return x


def clip_attention_pool(x: Tensor, sd: SD, p: str = "visual.attnpool.") -> Tensor:
    """AttentionPool2d: [N, C, H, W] -> [N, output_dim]; heads = C // 64."""
    N, C = x.shape[0], x.shape[1]
    heads, hd = C // 64, 64
    t = x.flatten(2).permute(0, 2, 1)                                   # [N, HW, C]
    t = torch.cat([t.mean(dim=1, keepdim=True), t], dim=1) + sd[p + "positional_embedding"][None]
    q = (F.linear(t[:, :1], sd[p + "q_proj.weight"], sd[p + "q_proj.bias"]) * hd ** -0.5).view(N, 1, heads, hd).transpose(1, 2)
    k = F.linear(t, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"]).view(N, -1, heads, hd).transpose(1, 2)
    v = F.linear(t, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"]).view(N, -1, heads, hd).transpose(1, 2)
    w = torch.matmul(q, k.transpose(-1, -2)).softmax(dim=-1)
    o = torch.matmul(w, v).transpose(1, 2).reshape(N, C)
    return F.linear(o, sd[p + "c_proj.weight"], sd[p + "c_proj.bias"])


def clip_encode_image_resnet(image: Tensor, sd: SD) -> Tensor:
    """CLIP.encode_image for a ModifiedResNet visual tower (RN50x4: [N, 3, 288, 288] -> [N, 640])."""
    return clip_attention_pool(clip_resnet_features(image, sd), sd)


# ----------------------------------------------------------------------------------------
# Image preprocessing in front of encode_image: the `preprocess` callable that
# `clip.load` returns and the reference applies to every PIL image (predictions_runner.py:212,
# embeddings_generator.py:72) = torchvision Compose[Resize(n_px, BICUBIC), CenterCrop(n_px),
# convert("RGB"), ToTensor(), Normalize(mean, std)]; `clip_transform_full` (predictions_runner.py:116-122)
# is the stretch-to-square variant.  torchvision is not installed here; on PIL images its Resize is
# `Image.resize(size, BICUBIC)`, restated below from Pillow's Resample.c (8-bit path: separable two-pass
# convolution, antialiased support = 2 * max(scale, 1), coefficients in 22-bit fixed point, horizontal
# pass first, uint8 between the passes).  Pinned against PIL itself (tests/golden/preprocess.npz).
# ----------------------------------------------------------------------------------------
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # predictions_runner.py:121
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
_PIL_PRECISION_BITS = 32 - 8 - 2


def _bicubic_filter(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_resample_coeffs(in_size: int, out_size: int):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the bicubic filter over the whole axis:
    per output index (xmin, [int coefficients])."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ss = 1.0 / filterscale
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        kk = [int(-0.5 + v * (1 << _PIL_PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PIL_PRECISION_BITS)) for v in w]
        out.append((xmin, kk))
    return out


def pil_bicubic_resize(img, out_w: int, out_h: int):
    """``Image.fromarray(img).resize((out_w, out_h), Image.BICUBIC)`` for a uint8 [H, W, C] array (numpy)."""
    import numpy as np
    img = np.asarray(img)
    H, W, C = img.shape
    half = 1 << (_PIL_PRECISION_BITS - 1)

    def clip8(v):
        return np.clip(v >> _PIL_PRECISION_BITS, 0, 255).astype(np.uint8)

    cur = img
    if out_w != W:                                            # horizontal pass first
        co = pil_resample_coeffs(W, out_w)
        tmp = np.empty((H, out_w, C), np.uint8)
        for xx, (xmin, kk) in enumerate(co):
            k = np.asarray(kk, np.int64)
            acc = half + (cur[:, xmin:xmin + len(kk), :].astype(np.int64) * k[None, :, None]).sum(1)
            tmp[:, xx, :] = clip8(acc)
        cur = tmp
    if out_h != H:
        co = pil_resample_coeffs(H, out_h)
        tmp = np.empty((out_h, cur.shape[1], C), np.uint8)
        for yy, (ymin, kk) in enumerate(co):
            k = np.asarray(kk, np.int64)
            acc = half + (cur[ymin:ymin + len(kk), :, :].astype(np.int64) * k[:, None, None]).sum(0)
            tmp[yy] = clip8(acc)
        cur = tmp
    return cur


def clip_preprocess_geometry(H: int, W: int, n_px: int = 224, stretch: bool = False):
    """resized (h, w) and crop origin (top, left) of torchvision Resize(n_px) + CenterCrop(n_px)
    (functional `_compute_resized_output_size`: the longer side is int(n_px * long / short); center_crop uses
    int(round((size - n_px) / 2.0)) with Python's round-half-even); stretch = Resize((n_px, n_px)), no crop."""
    if stretch:
        return n_px, n_px, 0, 0
    if W <= H:
        rw, rh = n_px, int(n_px * H / W)
    else:
        rh, rw = n_px, int(n_px * W / H)
    return rh, rw, int(round((rh - n_px) / 2.0)), int(round((rw - n_px) / 2.0))


def clip_preprocess(img, n_px: int = 224, stretch: bool = False) -> Tensor:
    """uint8 RGB image [H, W, 3] -> fp32 [3, n_px, n_px]: bicubic resize, centre crop, /255, (x - mean) / std."""
    import numpy as np
    img = np.asarray(img)
    rh, rw, top, left = clip_preprocess_geometry(img.shape[0], img.shape[1], n_px, stretch)
    r = pil_bicubic_resize(img, rw, rh)[top:top + n_px, left:left + n_px]
    x = torch.from_numpy(np.ascontiguousarray(r)).permute(2, 0, 1).to(torch.float32).div(255)
    mean = torch.tensor(CLIP_MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(CLIP_STD, dtype=torch.float32).view(3, 1, 1)
    return x.sub(mean).div(std)
