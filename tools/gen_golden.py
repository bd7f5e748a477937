#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE implementation (imported from
/root/reference, build container only) on seeded inputs and seeded "hot" weights.

Only inputs, seeds and outputs are written -- never reference source.  The reference has
no tests of its own; these files are the pins the oracle
(oracle/capdec_oracle.py) and the HIP path are checked against.

Shim sequence = SURVEY.md Appendix A (transformers>=5 has no AdamW; clip / pycocotools are
not installed; train.py hard-codes cuda:0; no 'gpt2' checkpoint offline).

usage: python tools/gen_golden.py [--only NAME ...]
"""
from __future__ import annotations

import argparse
import os
import pickle
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from capdec_amd import synth  # noqa: E402


# ---------------------------------------------------------------------------- reference import
def import_reference():
    sys.path.insert(0, REF)
    import transformers
    from transformers import GPT2LMHeadModel, GPT2Tokenizer, get_linear_schedule_with_warmup
    stub = types.ModuleType("transformers")
    stub.GPT2LMHeadModel, stub.GPT2Tokenizer = GPT2LMHeadModel, GPT2Tokenizer
    stub.AdamW, stub.get_linear_schedule_with_warmup = torch.optim.AdamW, get_linear_schedule_with_warmup
    real = sys.modules["transformers"]
    sys.modules["transformers"] = stub
    sys.modules["clip"] = types.ModuleType("clip")
    pc, pcc = types.ModuleType("pycocotools"), types.ModuleType("pycocotools.coco")
    pcc.COCO = object
    sys.modules["pycocotools"], sys.modules["pycocotools.coco"] = pc, pcc
    import gpt2_prefix, gpt2_prefix_eval, transformer_mapper, train as ref_train  # noqa: E401
    sys.modules["transformers"] = real
    ref_train.device = torch.device("cpu")
    return gpt2_prefix, gpt2_prefix_eval, transformer_mapper, ref_train


class FakeTok:
    """generate_* only need encode('.')[0] and decode(ids)."""

    def __init__(self, stop=13):
        self.stop = stop

    def encode(self, s):
        return [self.stop]

    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)


def build_ref_model(gpt2_prefix, dims, mapping_type, prefix_dim, P, clip_length=10, num_layers=8, seed=42):
    from transformers import GPT2Config, GPT2LMHeadModel
    cfg = GPT2Config(n_layer=dims.n_layer, n_head=dims.n_head, n_embd=dims.n_embd, vocab_size=dims.vocab,
                     n_positions=dims.n_pos)
    GPT2LMHeadModel.from_pretrained = staticmethod(lambda name, *a, **k: GPT2LMHeadModel(cfg))
    mt = {"mlp": gpt2_prefix.MappingType.MLP, "transformer_encoder": gpt2_prefix.MappingType.TransformerEncoder}[mapping_type]
    model = gpt2_prefix.ClipCaptionModel(P, clip_length=clip_length, prefix_dim=prefix_dim, num_layers=num_layers,
                                         mapping_type=mt).eval()
    sd = synth.hot_state_dict(seed, mapping_type, prefix_dim, P, clip_length, num_layers, dims)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(".attn.bias" in m or ".attn.masked_bias" in m for m in missing), missing
    return model, sd


def capture_beam(gpt2_prefix_eval, model, tok, embed, entry_length):
    """Run reference generate_beam and capture its final locals via sys.setprofile."""
    got = {}

    def prof(frame, event, arg):
        if event == "return" and frame.f_code.co_name == "generate_beam":
            loc = frame.f_locals
            got["tokens"] = loc["tokens"].cpu().numpy().astype(np.int64)
            got["seq_lengths"] = loc["seq_lengths"].cpu().numpy().astype(np.float32)
            got["scores"] = loc["scores"].cpu().numpy().astype(np.float32)
            got["order"] = loc["order"].cpu().numpy().astype(np.int64)

    sys.setprofile(prof)
    try:
        texts = gpt2_prefix_eval.generate_beam(model, tok, embed=embed, entry_length=entry_length)
    finally:
        sys.setprofile(None)
    got["texts"] = texts
    return got


def pad_tokens(tok, T, beam=5):
    out = np.zeros((beam, T), np.int64)
    out[:, :tok.shape[1]] = tok
    return out


# ---------------------------------------------------------------------------- fixture writers
def gen_mappers(refs):
    gpt2_prefix, _, transformer_mapper, _ = refs
    out = {}
    for D in (512, 640):
        x = synth.synthetic_clip_embeddings(4, D, seed=10 + D)
        out[f"x_{D}"] = x.numpy()
        sd = synth.hot_mlp_mapper_state_dict(43, D, 10)
        m = gpt2_prefix.MLP((D, 3840, 7680)).eval()
        m.load_state_dict({k[len("clip_project."):]: v for k, v in sd.items()})
        with torch.no_grad():
            out[f"mlp_{D}"] = m(x).numpy()
        out[f"mlp_{D}_crc"] = np.uint32(synth.state_dict_checksum(sd))
        sd = synth.hot_transformer_mapper_state_dict(43, D, 10, 10, 8)
        m = transformer_mapper.TransformerMapper(D, 768, 10, 10, 8).eval()
        m.load_state_dict({k[len("clip_project."):]: v for k, v in sd.items()})
        with torch.no_grad():
            out[f"tm_{D}"] = m(x).numpy()
        out[f"tm_{D}_crc"] = np.uint32(synth.state_dict_checksum(sd))
    # a second geometry: prefix_length 5, clip_length 7, 3 layers (ragged vs the default)
    x = synth.synthetic_clip_embeddings(3, 512, seed=77)
    sd = synth.hot_transformer_mapper_state_dict(44, 512, 5, 7, 3)
    m = transformer_mapper.TransformerMapper(512, 768, 5, 7, 3).eval()
    m.load_state_dict({k[len("clip_project."):]: v for k, v in sd.items()})
    with torch.no_grad():
        out["x_p5"], out["tm_p5"] = x.numpy(), m(x).numpy()
    np.savez_compressed(os.path.join(OUT, "mappers.npz"), **out)
    print("mappers.npz", {k: getattr(v, "shape", v) for k, v in out.items()})


def gen_noise(refs):
    _, _, _, ref_train = refs
    out = {}
    centers = pickle.load(open(os.path.join(REF, "others", "CLIP_embeddings_centers_info.pkl"), "rb"))
    off = centers["offset_to_add_in_training"].float()
    out["offset_to_add_in_training"] = off.numpy()
    out["offset_to_add_in_inference"] = centers["offset_to_add_in_inference"].float().numpy()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 640, generator=g) * 3.0
    noise = torch.randn(6, 640, generator=g)
    u = torch.rand(6, generator=g)
    out["x"], out["noise"], out["u"] = x.numpy(), noise.numpy(), u.numpy()
    real_randn, real_rand = torch.randn, torch.rand
    try:
        torch.randn = lambda *a, **k: noise.clone()
        torch.rand = lambda *a, **k: u.clone()
        out["v0"] = ref_train.noise_injection(x.clone(), 0.0).numpy()
        out["v016"] = ref_train.noise_injection(x.clone(), 0.016).numpy()
        out["v016_off"] = ref_train.noise_injection(x.clone(), 0.016, modality_offset=off).numpy()
        out["v016_dontnorm"] = ref_train.noise_injection(x.clone(), 0.016, dont_norm=True).numpy()
        out["v016_uniform"] = ref_train.noise_injection(x.clone(), 0.016, uniform_noise=True).numpy()
        out["ball"] = ref_train.get_uniform_ball_noise((6, 640), radius=0.3).numpy()
    finally:
        torch.randn, torch.rand = real_randn, real_rand
    np.savez_compressed(os.path.join(OUT, "noise.npz"), **out)
    print("noise.npz", {k: v.shape for k, v in out.items()})


def gen_gpt2_logits(refs, dims, tag):
    gpt2_prefix = refs[0]
    model, sd = build_ref_model(gpt2_prefix, dims, "mlp", 512, 10)
    out = {"gpt_crc": np.uint32(synth.state_dict_checksum({k: v for k, v in sd.items() if k.startswith("gpt.")}))}
    g = torch.Generator().manual_seed(7)
    for L in (1, 10, 23, 77):
        x = torch.randn(2, L, 768, generator=g) * 0.5
        with torch.no_grad():
            logits = model.gpt(inputs_embeds=x).logits  # [2, L, V]
        last = logits[:, -1, :]
        out[f"x_L{L}"] = x.numpy()
        out[f"last_sub_L{L}"] = last[:, ::5].numpy()
        tv, ti = last.topk(8, -1)
        out[f"top_v_L{L}"], out[f"top_i_L{L}"] = tv.numpy(), ti.numpy()
        out[f"lse_L{L}"] = torch.logsumexp(last, -1).numpy()
        # logits of every position for a few columns (checks non-last rows / causal mask)
        out[f"allpos_sub_L{L}"] = logits[:, :, :: max(1, dims.vocab // 64)].numpy()
    np.savez_compressed(os.path.join(OUT, f"gpt2_logits_{tag}.npz"), **out)
    print(f"gpt2_logits_{tag}.npz written")


def gen_decode(refs, dims, tag, n_greedy, n_beam, lengths):
    gpt2_prefix, gpt2_prefix_eval = refs[0], refs[1]
    out = {}
    # ---- greedy: config 1 shape -- 8 x 640-d embeddings, MLP mapper, P = 10
    model, sd = build_ref_model(gpt2_prefix, dims, "mlp", 640, 10)
    out["greedy_sd_crc"] = np.uint32(synth.state_dict_checksum(sd))
    x = synth.synthetic_clip_embeddings(n_greedy, 640, seed=1)
    out["greedy_x"] = x.numpy()
    T = max(lengths)
    ids_free = np.zeros((n_greedy, T), np.int64)
    with torch.no_grad():
        pe = model.clip_project(x).reshape(n_greedy, 10, -1)
        out["greedy_prefix_embed"] = pe.numpy()
        # pass 1: stop id that never fires (-> full-length sequences)
        for r in range(n_greedy):
            txt = gpt2_prefix_eval.generate2(model, FakeTok(stop=dims.vocab + 5), embed=pe[r:r + 1], entry_length=T)
            ids_free[r] = [int(t) for t in txt.split()]
    out["greedy_ids_nostop"] = ids_free
    # pass 2: choose a stop id that occurs mid-sequence in several rows
    vals, counts = np.unique(ids_free[:, 2:], return_counts=True)
    stop = int(vals[np.argmax(counts)])
    out["greedy_stop_id"] = np.int64(stop)
    for el in lengths:
        ids = np.zeros((n_greedy, el), np.int64)
        lens = np.zeros(n_greedy, np.int64)
        with torch.no_grad():
            for r in range(n_greedy):
                txt = gpt2_prefix_eval.generate2(model, FakeTok(stop=stop), embed=pe[r:r + 1], entry_length=el)
                t = [int(v) for v in txt.split()]
                ids[r, :len(t)], lens[r] = t, len(t)
        out[f"greedy_ids_T{el}"], out[f"greedy_lens_T{el}"] = ids, lens
    # ---- beam: config 3 shape -- 512-d embeddings, TransformerMapper(8 layers), P = 10, beam 5
    model, sd = build_ref_model(gpt2_prefix, dims, "transformer_encoder", 512, 10)
    out["beam_sd_crc"] = np.uint32(synth.state_dict_checksum(sd))
    x = synth.synthetic_clip_embeddings(n_beam, 512, seed=2)
    out["beam_x"] = x.numpy()
    with torch.no_grad():
        pe = model.clip_project(x).reshape(n_beam, 10, -1)
        out["beam_prefix_embed"] = pe.numpy()
        first = [capture_beam(gpt2_prefix_eval, model, FakeTok(stop=dims.vocab + 5), pe[r:r + 1], min(lengths))
                 for r in range(n_beam)]
    allt = np.concatenate([f["tokens"][:, 1:].reshape(-1) for f in first])
    vals, counts = np.unique(allt, return_counts=True)
    stop = int(vals[np.argmax(counts)])
    out["beam_stop_id"] = np.int64(stop)
    for el in lengths:
        for name, st in (("nostop", dims.vocab + 5), ("stop", stop)):
            toks = np.zeros((n_beam, 5, el), np.int64)
            seql = np.zeros((n_beam, 5), np.float32)
            scs = np.zeros((n_beam, 5), np.float32)
            order = np.zeros((n_beam, 5), np.int64)
            with torch.no_grad():
                for r in range(n_beam):
                    got = capture_beam(gpt2_prefix_eval, model, FakeTok(stop=st), pe[r:r + 1], el)
                    toks[r] = pad_tokens(got["tokens"], el)
                    seql[r], scs[r], order[r] = got["seq_lengths"], got["scores"], got["order"]
                    # the texts the function returns must equal decode(tokens[b,:len]) in `order`
                    exp = [" ".join(str(int(v)) for v in got["tokens"][b, :int(got["seq_lengths"][b])]) for b in got["order"]]
                    assert exp == got["texts"], (exp, got["texts"])
            out[f"beam_{name}_tokens_T{el}"] = toks
            out[f"beam_{name}_seqlen_T{el}"] = seql
            out[f"beam_{name}_scores_T{el}"] = scs
            out[f"beam_{name}_order_T{el}"] = order
    np.savez_compressed(os.path.join(OUT, f"decode_{tag}.npz"), **out)
    print(f"decode_{tag}.npz written; greedy stop {out['greedy_stop_id']}, beam stop {out['beam_stop_id']}")


def gen_decode_p40(refs, dims, tag):
    """the notebook's geometry (others/CapDec_inference.ipynb: prefix_length 40, RN50x4 640-d): contexts up to
    P + T - 1 = 106 tokens; MLP mapper 640 -> 15360 -> 30720, TransformerMapper with 80-token sequences"""
    gpt2_prefix, gpt2_prefix_eval = refs[0], refs[1]
    out = {}
    for mapping, n in (("mlp", 3), ("transformer_encoder", 3)):
        model, sd = build_ref_model(gpt2_prefix, dims, mapping, 640, 40, clip_length=40, num_layers=2, seed=11)
        out[f"{mapping}_crc"] = np.uint32(synth.state_dict_checksum(sd))
        x = synth.synthetic_clip_embeddings(n, 640, seed=21)
        with torch.no_grad():
            pe = model.clip_project(x).reshape(n, 40, -1)
            out[f"{mapping}_x"], out[f"{mapping}_prefix_embed"] = x.numpy(), pe.numpy()
            ids = np.zeros((n, 67), np.int64)
            lens = np.zeros(n, np.int64)
            for r in range(n):
                t = [int(v) for v in gpt2_prefix_eval.generate2(model, FakeTok(stop=dims.vocab + 5), embed=pe[r:r + 1]).split()]
                ids[r, :len(t)], lens[r] = t, len(t)
            out[f"{mapping}_greedy_ids"], out[f"{mapping}_greedy_lens"] = ids, lens
            toks = np.zeros((n, 5, 67), np.int64); seql = np.zeros((n, 5), np.float32)
            scs = np.zeros((n, 5), np.float32); order = np.zeros((n, 5), np.int64)
            stop = int(ids[0, 5])
            out[f"{mapping}_beam_stop_id"] = np.int64(stop)
            for r in range(n):
                got = capture_beam(gpt2_prefix_eval, model, FakeTok(stop=stop), pe[r:r + 1], 67)
                toks[r] = pad_tokens(got["tokens"], 67)
                seql[r], scs[r], order[r] = got["seq_lengths"], got["scores"], got["order"]
            out[f"{mapping}_beam_tokens"], out[f"{mapping}_beam_seqlen"] = toks, seql
            out[f"{mapping}_beam_scores"], out[f"{mapping}_beam_order"] = scs, order
    np.savez_compressed(os.path.join(OUT, f"decode_p40_{tag}.npz"), **out)
    print(f"decode_p40_{tag}.npz written")


class PromptTok(FakeTok):
    """tokenizer stand-in for the prompt / tokens entry of generate2 / generate_beam: a prompt is a string of ids."""

    def encode(self, s):
        return [self.stop] if s == "." else [int(v) for v in s.split()]


def gen_prompt(refs, dims, tag):
    """generate2(tokens=...), generate2(prompt=...), generate_beam(prompt=...) (reference gpt2_prefix_eval.py:70-74,
    86-89,141-151): the prefix is wte(prompt ids) instead of a mapped CLIP embedding, and the prompt ids stay in the
    output (generate_beam even slices the concatenated row by the GENERATED length only, :111)."""
    gpt2_prefix, gpt2_prefix_eval = refs[0], refs[1]
    model, sd = build_ref_model(gpt2_prefix, dims, "mlp", 512, 10)
    out = {"sd_crc": np.uint32(synth.state_dict_checksum(sd))}
    g = torch.Generator().manual_seed(31)
    prompts = [torch.randint(1, dims.vocab - 1, (L,), generator=g).tolist() for L in (1, 3, 7, 5)]
    out["prompt_lens"] = np.array([len(p) for p in prompts], np.int64)
    pp = np.zeros((len(prompts), 8), np.int64)
    for i, p in enumerate(prompts):
        pp[i, :len(p)] = p
    out["prompts"] = pp
    T = 12
    free = []
    with torch.no_grad():
        for p in prompts:          # pass 1: never stop
            txt = gpt2_prefix_eval.generate2(model, PromptTok(stop=dims.vocab + 5), tokens=torch.tensor([p]), entry_length=T)
            free.append([int(v) for v in txt.split()])
    gen_part = np.concatenate([np.array(f[len(p) + 2:]) for f, p in zip(free, prompts)])
    vals, counts = np.unique(gen_part, return_counts=True)
    stop = int(vals[np.argmax(counts)])
    out["stop_id"] = np.int64(stop)
    for name, st in (("nostop", dims.vocab + 5), ("stop", stop)):
        texts_tok, texts_prompt, beams = [], [], []
        with torch.no_grad():
            for p in prompts:
                a = gpt2_prefix_eval.generate2(model, PromptTok(stop=st), tokens=torch.tensor([p]), entry_length=T)
                b = gpt2_prefix_eval.generate2(model, PromptTok(stop=st), prompt=" ".join(str(v) for v in p), entry_length=T)
                assert a == b
                texts_tok.append(a)
                texts_prompt.append(b)
                beams.append(gpt2_prefix_eval.generate_beam(model, PromptTok(stop=st), prompt=" ".join(str(v) for v in p),
                                                            entry_length=T))
        out[f"generate2_{name}"] = np.array(texts_tok)
        out[f"generate_beam_{name}"] = np.array(beams)          # [n_prompts, 5] strings, best first
    np.savez_compressed(os.path.join(OUT, f"prompt_{tag}.npz"), **out)
    print(f"prompt_{tag}.npz written; stop {stop}", out["generate2_stop"], out["generate_beam_stop"][1])


def gen_train_forward(refs, dims, tag):
    """train.ClipCaptionModel.forward (train.py:251-260) on a right-padded batch exactly as train.ClipCocoDataset builds
    it (:52-63), and the loss of the train step (:349: cross_entropy(logits[:, P-1:-1], tokens, ignore_index=0))."""
    ref_train = refs[3]
    from transformers import GPT2Config, GPT2LMHeadModel
    cfg = GPT2Config(n_layer=dims.n_layer, n_head=dims.n_head, n_embd=dims.n_embd, vocab_size=dims.vocab,
                     n_positions=dims.n_pos)
    GPT2LMHeadModel.from_pretrained = staticmethod(lambda name, *a, **k: GPT2LMHeadModel(cfg))
    P = 10
    model = ref_train.ClipCaptionModel(P, clip_length=10, prefix_size=512, num_layers=8,
                                       mapping_type=ref_train.MappingType.MLP).eval()
    sd = synth.hot_state_dict(42, "mlp", 512, P, dims=dims)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(".attn.bias" in m or ".attn.masked_bias" in m for m in missing)
    g = torch.Generator().manual_seed(17)
    lens = [9, 4, 7]
    L = max(lens)
    tokens = torch.zeros(len(lens), L, dtype=torch.int64)
    mask = torch.zeros(len(lens), P + L)
    mask[:, :P] = 1
    for r, n in enumerate(lens):
        tokens[r, :n] = torch.randint(1, dims.vocab, (n,), generator=g)
        mask[r, P:P + n] = 1
    prefix = synth.synthetic_clip_embeddings(len(lens), 512, seed=5)
    with torch.no_grad():
        out = model(tokens, prefix, mask)
        logits = out.logits
        loss = torch.nn.functional.cross_entropy(logits[:, P - 1:-1].reshape(-1, logits.shape[-1]), tokens.flatten(),
                                                 ignore_index=0)
        out_l = model(tokens, prefix, mask, labels=tokens)          # GPT2LMHeadModel's own shifted loss
    step = max(1, dims.vocab // 97)
    res = {"sd_crc": np.uint32(synth.state_dict_checksum(sd)), "tokens": tokens.numpy(), "mask": mask.numpy(),
           "prefix": prefix.numpy(), "lens": np.array(lens), "logits_sub": logits[:, :, ::step].numpy(),
           "argmax": logits.argmax(-1).numpy(), "lse": torch.logsumexp(logits, -1).numpy(),
           "train_loss": np.float32(loss), "hf_loss_all_positions": np.float32(out_l.loss)}
    np.savez_compressed(os.path.join(OUT, f"train_forward_{tag}.npz"), **res)
    print(f"train_forward_{tag}.npz written; loss {float(loss):.5f}")


def gen_train_step(refs, dims, tag, mapping="mlp"):
    """The train step with a frozen GPT-2 (train.py:344-354 with --only_prefix): train.ClipCaptionPrefix (:279-287) in
    train() mode, the batch of gen_train_forward, loss as at :349, the reference's own loss.backward() (:350) -> the
    gradients of ClipCaptionPrefix.parameters().  Three iterations follow with the lr of the real
    transformers.get_linear_schedule_with_warmup; the optimizer update itself is the oracle's restatement of
    transformers-4.24 AdamW (the class is gone from the installed 5.15: that part is NOT pinned by this file -- what is
    pinned is every gradient, computed by autograd at the weights the previous updates produced, and the losses)."""
    ref_train = refs[3]
    from transformers import GPT2Config, GPT2LMHeadModel, get_linear_schedule_with_warmup
    from oracle import capdec_oracle as O
    cfg = GPT2Config(n_layer=dims.n_layer, n_head=dims.n_head, n_embd=dims.n_embd, vocab_size=dims.vocab,
                     n_positions=dims.n_pos)
    GPT2LMHeadModel.from_pretrained = staticmethod(lambda name, *a, **k: GPT2LMHeadModel(cfg))
    P, D = 10, 512
    mt = ref_train.MappingType.MLP if mapping == "mlp" else ref_train.MappingType.Transformer
    nlay = 8 if mapping == "mlp" else 3                      # (three mapper layers keep the fixture small)
    model = ref_train.ClipCaptionPrefix(P, clip_length=10, prefix_size=D, num_layers=nlay, mapping_type=mt)
    model.train()
    assert not model.gpt.training and model.clip_project.training
    sd = synth.hot_state_dict(42, mapping, D, P, 10, nlay, dims)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(".attn.bias" in m or ".attn.masked_bias" in m for m in missing)
    names = [k for k, _ in model.named_parameters() if k.startswith("clip_project.")]
    params = list(model.parameters())
    assert len(params) == len(names) == (4 if mapping == "mlp" else 3 + 12 * nlay)
    g = torch.Generator().manual_seed(23)
    lr, warm, total, iters = 2e-3, 2, 8, 4
    dummy = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=lr)     # only carries the lr for the real scheduler
    sched = get_linear_schedule_with_warmup(dummy, num_warmup_steps=warm, num_training_steps=total)
    state = {k: (torch.zeros_like(q), torch.zeros_like(q)) for k, q in zip(names, params)}
    res = {"sd_crc": np.uint32(synth.state_dict_checksum(sd)), "lr": np.float64(lr), "warmup": np.int32(warm),
           "total": np.int32(total), "names": np.array(names)}
    losses, lrs = [], []
    for it in range(iters):
        lens = [[9, 4, 7], [5, 8, 3], [6, 6, 9], [2, 7, 5]][it]
        L = max(lens)
        tokens = torch.zeros(len(lens), L, dtype=torch.int64)
        mask = torch.zeros(len(lens), P + L)
        mask[:, :P] = 1
        for r, n in enumerate(lens):
            tokens[r, :n] = torch.randint(1, dims.vocab, (n,), generator=g)
            mask[r, P:P + n] = 1
        if it == 1:
            tokens[1, 2] = 0                 # a REAL token with id 0 is ignored by the loss as well (ignore_index=0)
        prefix = synth.synthetic_clip_embeddings(len(lens), D, seed=50 + it)
        model.zero_grad()
        out = model(tokens, prefix, mask)
        logits = out.logits[:, P - 1:-1]
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), tokens.flatten(), ignore_index=0)
        loss.backward()
        cur_lr = dummy.param_groups[0]["lr"]
        res[f"tokens_{it}"], res[f"mask_{it}"], res[f"prefix_{it}"] = tokens.numpy(), mask.numpy(), prefix.numpy()
        for k, q in zip(names, params):
            gk = q.grad.detach()
            flat = gk.flatten()
            res[f"grad_{it}_{k}_sub"] = flat[::max(1, flat.numel() // 4096)].numpy().copy()
            res[f"grad_{it}_{k}_norm"] = np.float64(gk.double().norm())
            with torch.no_grad():
                O.adamw_transformers(q.data, gk, state[k][0], state[k][1], it + 1, cur_lr)
        losses.append(float(loss.detach()))
        lrs.append(cur_lr)
        dummy.step()
        sched.step()
    res["losses"], res["lrs"] = np.array(losses, np.float32), np.array(lrs, np.float64)
    for k, q in zip(names, params):
        flat = q.detach().flatten()
        res[f"final_{k}_sub"] = flat[::max(1, flat.numel() // 4096)].numpy().copy()
    np.savez_compressed(os.path.join(OUT, f"train_step_{tag}.npz"), **res)
    print(f"train_step_{tag}.npz written; losses {losses} lrs {lrs}")


def gen_train_full(refs, dims, tag):
    """The reference's DEFAULT train step (train.py:344-350 with train.ClipCaptionModel: model.parameters() includes
    GPT-2): gradients of EVERY tensor -- mapper, all GPT-2 blocks, ln_f, wpe and the tied wte -- from the reference's own
    loss.backward(), with GPT-2's dropouts set to 0 (the reference trains with transformers' default 0.1, which no fixture
    can pin).  One iteration; pins oracle/capdec_oracle.py train_step_loss_and_grads(train_gpt=True)."""
    ref_train = refs[3]
    from transformers import GPT2Config, GPT2LMHeadModel
    cfg = GPT2Config(n_layer=dims.n_layer, n_head=dims.n_head, n_embd=dims.n_embd, vocab_size=dims.vocab,
                     n_positions=dims.n_pos, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)
    GPT2LMHeadModel.from_pretrained = staticmethod(lambda name, *a, **k: GPT2LMHeadModel(cfg))
    P, D = 10, 512
    model = ref_train.ClipCaptionModel(P, clip_length=10, prefix_size=D, num_layers=8, mapping_type=ref_train.MappingType.MLP)
    model.train()
    assert model.gpt.training
    sd = synth.hot_state_dict(42, "mlp", D, P, dims=dims)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(".attn.bias" in m or ".attn.masked_bias" in m for m in missing)
    g = torch.Generator().manual_seed(29)
    lens = [7, 9, 3, 6]
    L = max(lens)
    tokens = torch.zeros(len(lens), L, dtype=torch.int64)
    mask = torch.zeros(len(lens), P + L)
    mask[:, :P] = 1
    for r, n in enumerate(lens):
        tokens[r, :n] = torch.randint(1, dims.vocab, (n,), generator=g)
        mask[r, P:P + n] = 1
    tokens[2, 1] = tokens[0, 3]                                  # a token id that occurs twice: the lookup gradient adds up
    prefix = synth.synthetic_clip_embeddings(len(lens), D, seed=61)
    model.zero_grad()
    out = model(tokens, prefix, mask)
    logits = out.logits[:, P - 1:-1]
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), tokens.flatten(), ignore_index=0)
    loss.backward()
    names = [k for k, q in model.named_parameters() if q.grad is not None]
    res = {"sd_crc": np.uint32(synth.state_dict_checksum(sd)), "tokens": tokens.numpy(), "mask": mask.numpy(),
           "prefix": prefix.numpy(), "loss": np.float32(loss.detach()), "names": np.array(names)}
    for k, q in model.named_parameters():
        if q.grad is None:
            continue
        flat = q.grad.detach().flatten()
        res[f"grad_{k}_sub"] = flat[::max(1, flat.numel() // 1024)].numpy().copy()
        res[f"grad_{k}_norm"] = np.float64(q.grad.detach().double().norm())
    np.savez_compressed(os.path.join(OUT, f"train_full_{tag}.npz"), **res)
    print(f"train_full_{tag}.npz written; loss {float(loss):.5f}; {len(names)} tensors")


def gen_train_full_dropout(refs, dims, tag):
    """The reference's DEFAULT train step as it really runs: train.ClipCaptionModel in train() mode with transformers'
    default dropouts (embd_pdrop = attn_pdrop = resid_pdrop = 0.1).  torch.nn.functional.dropout is patched for the
    duration of the forward so that every call draws its keep-mask from a seeded generator and the mask is RECORDED
    (same arithmetic as torch's: x * mask / (1 - p)); the eager attention path is forced so the attention dropout is a
    visible F.dropout call too.  Written: the masks (bit-packed, in call order), the loss and a subsample + norm of the
    gradient of every tensor from the reference's own loss.backward() -- two iterations' worth of independent batches
    (no update in between: the update rule is pinned elsewhere)."""
    ref_train = refs[3]
    from transformers import GPT2Config, GPT2LMHeadModel
    from oracle import capdec_oracle as O
    cfg = GPT2Config(n_layer=dims.n_layer, n_head=dims.n_head, n_embd=dims.n_embd, vocab_size=dims.vocab,
                     n_positions=dims.n_pos)
    assert cfg.resid_pdrop == cfg.embd_pdrop == cfg.attn_pdrop == 0.1
    cfg._attn_implementation = "eager"
    GPT2LMHeadModel.from_pretrained = staticmethod(lambda name, *a, **k: GPT2LMHeadModel(cfg))
    P, D = 10, 512
    model = ref_train.ClipCaptionModel(P, clip_length=10, prefix_size=D, num_layers=8, mapping_type=ref_train.MappingType.MLP)
    model.train()
    assert model.gpt.training
    sd = synth.hot_state_dict(42, "mlp", D, P, dims=dims)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(".attn.bias" in m or ".attn.masked_bias" in m for m in missing)
    res = {"sd_crc": np.uint32(synth.state_dict_checksum(sd)), "p": np.float64(0.1)}
    g = torch.Generator().manual_seed(31)
    real_dropout = torch.nn.functional.dropout
    for it, lens in enumerate([[7, 9, 3, 6], [5, 2, 8]]):
        L = max(lens)
        tokens = torch.zeros(len(lens), L, dtype=torch.int64)
        mask = torch.zeros(len(lens), P + L)
        mask[:, :P] = 1
        for r, n in enumerate(lens):
            tokens[r, :n] = torch.randint(1, dims.vocab, (n,), generator=g)
            mask[r, P:P + n] = 1
        prefix = synth.synthetic_clip_embeddings(len(lens), D, seed=71 + it)
        recorded = []
        mg = torch.Generator().manual_seed(900 + it)

        def recording_dropout(x, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return x
            keep = torch.rand(x.shape, generator=mg) >= p
            recorded.append(keep)
            return x * (keep.to(x.dtype) / (1.0 - p))

        model.zero_grad()
        torch.nn.functional.dropout = recording_dropout
        try:
            out = model(tokens, prefix, mask)
        finally:
            torch.nn.functional.dropout = real_dropout
        want_sites = O.dropout_sites(dims.n_layer, len(lens), P + L, dims.n_embd, dims.n_head)
        assert [tuple(m.shape) for m in recorded] == [s for _, s in want_sites], \
            ([tuple(m.shape) for m in recorded], want_sites)
        logits = out.logits[:, P - 1:-1]
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), tokens.flatten(), ignore_index=0)
        loss.backward()
        names = [k for k, q in model.named_parameters() if q.grad is not None]
        flatmask = torch.cat([m.flatten() for m in recorded]).numpy()
        res[f"tokens_{it}"], res[f"mask_{it}"], res[f"prefix_{it}"] = tokens.numpy(), mask.numpy(), prefix.numpy()
        res[f"drop_bits_{it}"], res[f"drop_n_{it}"] = np.packbits(flatmask), np.int64(flatmask.size)
        res[f"loss_{it}"] = np.float32(loss.detach())
        res["names"] = np.array(names)
        for k, q in model.named_parameters():
            if q.grad is None:
                continue
            flat = q.grad.detach().flatten()
            res[f"grad_{it}_{k}_sub"] = flat[::max(1, flat.numel() // 1024)].numpy().copy()
            res[f"grad_{it}_{k}_norm"] = np.float64(q.grad.detach().double().norm())
        print(f"  iteration {it}: loss {float(loss):.5f}, {len(recorded)} dropout calls, keep rate {flatmask.mean():.4f}")
    np.savez_compressed(os.path.join(OUT, f"train_full_dropout_{tag}.npz"), **res)
    print(f"train_full_dropout_{tag}.npz written; {len(names)} tensors")


def openai_to_hf_clip(sd, dims):
    """Map an OpenAI-CLIP-named state dict onto transformers.CLIPModel names (the independent
    stand-in used to pin the CLIP kernels: the reference's own `clip` package is not installed)."""
    out = {"logit_scale": sd["logit_scale"],
           "text_model.embeddings.token_embedding.weight": sd["token_embedding.weight"],
           "text_model.embeddings.position_embedding.weight": sd["positional_embedding"],
           "text_model.final_layer_norm.weight": sd["ln_final.weight"],
           "text_model.final_layer_norm.bias": sd["ln_final.bias"],
           "text_projection.weight": sd["text_projection"].t().contiguous(),
           "vision_model.embeddings.class_embedding": sd["visual.class_embedding"],
           "vision_model.embeddings.patch_embedding.weight": sd["visual.conv1.weight"],
           "vision_model.embeddings.position_embedding.weight": sd["visual.positional_embedding"],
           "vision_model.pre_layrnorm.weight": sd["visual.ln_pre.weight"],
           "vision_model.pre_layrnorm.bias": sd["visual.ln_pre.bias"],
           "vision_model.post_layernorm.weight": sd["visual.ln_post.weight"],
           "vision_model.post_layernorm.bias": sd["visual.ln_post.bias"],
           "visual_projection.weight": sd["visual.proj"].t().contiguous()}
    for tower, pfx, width, layers in (("text_model", "", dims.text_width, dims.text_layers),
                                      ("vision_model", "visual.", dims.vision_width, dims.vision_layers)):
        for i in range(layers):
            b, h = f"{pfx}transformer.resblocks.{i}.", f"{tower}.encoder.layers.{i}."
            wq, wk, wv = sd[b + "attn.in_proj_weight"].split(width, dim=0)
            bq, bk, bv = sd[b + "attn.in_proj_bias"].split(width, dim=0)
            out.update({h + "self_attn.q_proj.weight": wq, h + "self_attn.k_proj.weight": wk,
                        h + "self_attn.v_proj.weight": wv, h + "self_attn.q_proj.bias": bq,
                        h + "self_attn.k_proj.bias": bk, h + "self_attn.v_proj.bias": bv,
                        h + "self_attn.out_proj.weight": sd[b + "attn.out_proj.weight"],
                        h + "self_attn.out_proj.bias": sd[b + "attn.out_proj.bias"],
                        h + "layer_norm1.weight": sd[b + "ln_1.weight"], h + "layer_norm1.bias": sd[b + "ln_1.bias"],
                        h + "layer_norm2.weight": sd[b + "ln_2.weight"], h + "layer_norm2.bias": sd[b + "ln_2.bias"],
                        h + "mlp.fc1.weight": sd[b + "mlp.c_fc.weight"], h + "mlp.fc1.bias": sd[b + "mlp.c_fc.bias"],
                        h + "mlp.fc2.weight": sd[b + "mlp.c_proj.weight"], h + "mlp.fc2.bias": sd[b + "mlp.c_proj.bias"]})
    return out


def gen_clip(dims, tag, n_text, n_img):
    """CLIP ViT-B/32 text / image features from transformers.CLIPModel (random 'hot' weights)."""
    from transformers import CLIPConfig, CLIPModel
    cfg = CLIPConfig()
    cfg.text_config.num_hidden_layers = dims.text_layers
    cfg.vision_config.num_hidden_layers = dims.vision_layers
    model = CLIPModel(cfg).eval()
    sd = synth.hot_clip_state_dict(43, dims)
    missing, unexpected = model.load_state_dict(openai_to_hf_clip(sd, dims), strict=False)
    assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
    toks = synth.synthetic_clip_tokens(n_text, seed=2)
    imgs = synth.synthetic_images(n_img, seed=4)
    with torch.no_grad():
        tf = model.get_text_features(input_ids=toks)
        vf = model.get_image_features(pixel_values=imgs)
    tf = getattr(tf, "pooler_output", tf)
    vf = getattr(vf, "pooler_output", vf)
    out = {"crc": np.uint32(synth.state_dict_checksum(sd)), "tokens": toks.numpy(), "text_features": tf.numpy(),
           "image_seed": np.int64(4), "image_features": vf.numpy()}
    # the same stand-in in half precision = the arithmetic class of the reference on a GPU (`clip.load(..., device=cuda)`
    # converts the model to fp16, predictions_runner.py:218,220 cast the result back with .float())
    try:
        mh = model.half()
        with torch.no_grad():
            tfh = mh.get_text_features(input_ids=toks)
            vfh = mh.get_image_features(pixel_values=imgs.half())
        out["text_features_fp16"] = getattr(tfh, "pooler_output", tfh).float().numpy()
        out["image_features_fp16"] = getattr(vfh, "pooler_output", vfh).float().numpy()
    except Exception as ex:                      # CPU half kernels missing: keep the fp32 pin only
        print("fp16 stand-in not available:", ex)
    np.savez_compressed(os.path.join(OUT, f"clip_{tag}.npz"), **out)
    print(f"clip_{tag}.npz", tf.shape, vf.shape, float(tf.norm(dim=1).mean()), float(vf.norm(dim=1).mean()))


class _RnBlock(torch.nn.Module):
    """bottleneck of the published ModifiedResNet: 1x1 -> 3x3 -> (AvgPool2d(stride)) -> 1x1 (x4), anti-aliased strided
    shortcut (AvgPool2d, 1x1 conv, BatchNorm) when the shape changes; submodule names = OpenAI state-dict names"""

    def __init__(self, inplanes, planes, stride):
        super().__init__()
        nn = torch.nn
        self.conv1, self.bn1 = nn.Conv2d(inplanes, planes, 1, bias=False), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False), nn.BatchNorm2d(planes)
        self.avgpool = nn.AvgPool2d(stride) if stride > 1 else nn.Identity()
        self.conv3, self.bn3 = nn.Conv2d(planes, planes * 4, 1, bias=False), nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if stride > 1 or inplanes != planes * 4:
            from collections import OrderedDict
            self.downsample = nn.Sequential(OrderedDict([("-1", nn.AvgPool2d(stride)),
                                                         ("0", nn.Conv2d(inplanes, planes * 4, 1, bias=False)),
                                                         ("1", nn.BatchNorm2d(planes * 4))]))

    def forward(self, x):
        y = torch.relu(self.bn1(self.conv1(x)))
        y = torch.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(self.avgpool(y)))
        return torch.relu(y + (x if self.downsample is None else self.downsample(x)))


class _RnAttnPool(torch.nn.Module):
    def __init__(self, tokens, dim, heads, out_dim):
        super().__init__()
        nn = torch.nn
        self.positional_embedding = nn.Parameter(torch.zeros(tokens, dim))
        self.q_proj, self.k_proj, self.v_proj = nn.Linear(dim, dim), nn.Linear(dim, dim), nn.Linear(dim, dim)
        self.c_proj = nn.Linear(dim, out_dim)
        self.heads = heads

    def forward(self, x):
        x = x.flatten(2).permute(2, 0, 1)                                   # [HW, N, C]
        x = torch.cat([x.mean(dim=0, keepdim=True), x], dim=0) + self.positional_embedding[:, None, :]
        out, _ = torch.nn.functional.multi_head_attention_forward(
            query=x[:1], key=x, value=x, embed_dim_to_check=x.shape[-1], num_heads=self.heads,
            q_proj_weight=self.q_proj.weight, k_proj_weight=self.k_proj.weight, v_proj_weight=self.v_proj.weight,
            in_proj_weight=None, in_proj_bias=torch.cat([self.q_proj.bias, self.k_proj.bias, self.v_proj.bias]),
            bias_k=None, bias_v=None, add_zero_attn=False, dropout_p=0.0, out_proj_weight=self.c_proj.weight,
            out_proj_bias=self.c_proj.bias, use_separate_proj_weight=True, training=False, need_weights=False)
        return out.squeeze(0)


class _RnTower(torch.nn.Module):
    """CLIP's ModifiedResNet built from torch.nn modules as published (3-convolution stem with average pool, four stages
    of bottlenecks with strides 1/2/2/2, attention pool).  openai/CLIP is not installed here: this module is the independent
    witness of the oracle's functional restatement, and `load_state_dict(strict=True)` of the synthetic weights is the check
    that their names and shapes are those of an OpenAI state dict."""

    def __init__(self, dims):
        super().__init__()
        nn, w = torch.nn, dims.width
        self.conv1, self.bn1 = nn.Conv2d(3, w // 2, 3, stride=2, padding=1, bias=False), nn.BatchNorm2d(w // 2)
        self.conv2, self.bn2 = nn.Conv2d(w // 2, w // 2, 3, padding=1, bias=False), nn.BatchNorm2d(w // 2)
        self.conv3, self.bn3 = nn.Conv2d(w // 2, w, 3, padding=1, bias=False), nn.BatchNorm2d(w)
        self.avgpool = nn.AvgPool2d(2)
        inplanes = w
        for li, (planes, blocks) in enumerate(zip((w, 2 * w, 4 * w, 8 * w), dims.layers), start=1):
            layer = []
            for b in range(blocks):
                layer.append(_RnBlock(inplanes, planes, 2 if (b == 0 and li > 1) else 1))
                inplanes = planes * 4
            setattr(self, f"layer{li}", nn.Sequential(*layer))
        sp = dims.image_size // 32
        self.attnpool = _RnAttnPool(sp * sp + 1, dims.feat_dim, dims.heads, dims.embed_dim)

    def forward(self, x):
        for conv, bn in ((self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)):
            x = torch.relu(bn(conv(x)))
        x = self.avgpool(x)
        for li in (1, 2, 3, 4):
            x = getattr(self, f"layer{li}")(x)
        return self.attnpool(x)


def gen_clip_resnet():
    """RN50x4-class image towers: features of the torch.nn witness above on the synthetic OpenAI-named weights."""
    out = {}
    for tag, dims, n in (("tiny", synth.CLIP_RN_TINY, 5), ("rn50x4", synth.CLIP_RN50X4, 2)):
        sd = synth.hot_clip_resnet_state_dict(44, dims)
        tower = _RnTower(dims).eval()
        tower.load_state_dict({k[len("visual."):]: v for k, v in sd.items()}, strict=True)
        imgs = synth.synthetic_images(n, seed=12 if tag == "tiny" else 13, size=dims.image_size)
        with torch.no_grad():
            f = tower(imgs)
        out[f"crc_{tag}"] = np.uint32(synth.state_dict_checksum(sd))
        out[f"features_{tag}"] = f.numpy()
        out[f"image_seed_{tag}"] = np.int64(12 if tag == "tiny" else 13)
        print("clip_resnet", tag, tuple(f.shape), float(f.abs().max()))
    np.savez_compressed(os.path.join(OUT, "clip_resnet.npz"), **out)


def gen_preprocess():
    """CLIP `preprocess` (Resize(224, BICUBIC) -> CenterCrop(224) -> ToTensor -> Normalize) and the stretch variant
    `clip_transform_full` (predictions_runner.py:116-122) computed with PIL itself (torchvision is not installed: its
    Resize / CenterCrop on PIL images are Image.resize / Image.crop with the size arithmetic restated here) on the
    seeded images of synth.synthetic_photo.  Stored: crc32 of the uint8 crop and every 29th value of the float tensor."""
    import zlib
    from PIL import Image
    mean = torch.tensor((0.48145466, 0.4578275, 0.40821073)).view(3, 1, 1)
    std = torch.tensor((0.26862954, 0.26130258, 0.27577711)).view(3, 1, 1)
    n_px, out = 224, {}
    for i, (h, w) in enumerate(synth.PREPROCESS_SIZES):
        img = Image.fromarray(synth.synthetic_photo(h, w, 100 + i))
        for stretch in (0, 1):
            if stretch:
                r = img.resize((n_px, n_px), Image.BICUBIC)
            else:
                if w <= h:
                    rw, rh = n_px, int(n_px * h / w)
                else:
                    rh, rw = n_px, int(n_px * w / h)
                top, left = int(round((rh - n_px) / 2.0)), int(round((rw - n_px) / 2.0))
                r = img.resize((rw, rh), Image.BICUBIC).crop((left, top, left + n_px, top + n_px))
            u8 = np.ascontiguousarray(np.asarray(r.convert("RGB")))
            x = torch.from_numpy(u8.copy()).permute(2, 0, 1).to(torch.float32).div(255).sub(mean).div(std)
            out[f"crc_{i}_{stretch}"] = np.uint32(zlib.crc32(u8.tobytes()))
            out[f"sub_{i}_{stretch}"] = x.reshape(-1)[::29].numpy()
    out["pil_version"] = np.array(Image.__version__ if hasattr(Image, "__version__") else "?")
    np.savez_compressed(os.path.join(OUT, "preprocess.npz"), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(os.cpu_count() or 8)
    refs = None if args.only and set(args.only) <= {"preprocess", "clip_resnet", "clip_tiny", "clip_b32"} else import_reference()
    jobs = {
        "mappers": lambda: gen_mappers(refs),
        "noise": lambda: gen_noise(refs),
        "logits_tiny": lambda: gen_gpt2_logits(refs, synth.GPT2_TINY, "tiny"),
        "logits_small": lambda: gen_gpt2_logits(refs, synth.GPT2_SMALL, "small"),
        "decode_tiny": lambda: gen_decode(refs, synth.GPT2_TINY, "tiny", 8, 6, (12, 67)),
        "decode_small": lambda: gen_decode(refs, synth.GPT2_SMALL, "small", 8, 4, (12, 67)),
        "decode_p40_tiny": lambda: gen_decode_p40(refs, synth.GPT2_TINY, "tiny"),
        "train_forward_tiny": lambda: gen_train_forward(refs, synth.GPT2_TINY, "tiny"),
        "train_forward_small": lambda: gen_train_forward(refs, synth.GPT2_SMALL, "small"),
        "train_step_tiny": lambda: gen_train_step(refs, synth.GPT2_TINY, "tiny"),
        "train_step_small": lambda: gen_train_step(refs, synth.GPT2_SMALL, "small"),
        "train_full_tiny": lambda: gen_train_full(refs, synth.GPT2_TINY, "tiny"),
        "train_full_dropout_tiny": lambda: gen_train_full_dropout(refs, synth.GPT2_TINY, "tiny"),
        "train_step_tm_tiny": lambda: gen_train_step(refs, synth.GPT2_TINY, "tm_tiny", "transformer_encoder"),
        "prompt_tiny": lambda: gen_prompt(refs, synth.GPT2_TINY, "tiny"),
        "prompt_small": lambda: gen_prompt(refs, synth.GPT2_SMALL, "small"),
        "clip_tiny": lambda: gen_clip(synth.CLIP_TINY, "tiny", 6, 3),
        "clip_b32": lambda: gen_clip(synth.CLIP_VIT_B32, "b32", 6, 3),
        "clip_resnet": gen_clip_resnet,
        "preprocess": gen_preprocess,
    }
    for name, fn in jobs.items():
        if args.only and name not in args.only:
            continue
        print("==", name)
        fn()


if __name__ == "__main__":
    main()
