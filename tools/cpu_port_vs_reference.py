#!/usr/bin/env python
"""One-off (build container only: needs /root/reference): the oracle's reference-shaped CPU port
(oracle.capdec_oracle.generate_beam_ref / generate2_ref, what bench.py's `cpu_baseline` times on the GPU box) timed
beside the REAL reference import (gpt2_prefix_eval.generate_beam / generate2 on the reference's own ClipCaptionModel)
on the same seeded captions, same weights, same thread count -- SURVEY.md section 8 D.5.  Writes
profiles/r3_cpu_port_vs_reference.json.  usage: python tools/cpu_port_vs_reference.py [n_captions]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import gen_golden as G
from capdec_amd import synth
from oracle import capdec_oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
refs = G.import_reference()
gpt2_prefix, gpt2_prefix_eval = refs[0], refs[1]
torch.set_num_threads(os.cpu_count())
P, T, D = 10, 67, 512
model, sd = G.build_ref_model(gpt2_prefix, synth.GPT2_SMALL, "transformer_encoder", D, P)
tok = G.FakeTok(13)
x = synth.synthetic_clip_embeddings(n + 1, D, seed=0)
res = {"threads": torch.get_num_threads(), "host_cpus": os.cpu_count(), "captions": n, "workload": "TransformerMapper(8) -> GPT-2 small, prefix 10, entry_length 67 (no caption stops early with the hot-init weights)"}
with torch.no_grad():
    pe_ref = [model.clip_project(O.normalize_prefix(x[i:i + 1])).reshape(1, P, -1) for i in range(n + 1)]
    pe_port = [O.clip_project(O.normalize_prefix(x[i:i + 1]), sd, "transformer_encoder", P) for i in range(n + 1)]
    for kind, ref_fn, port_fn in (
            ("beam5", lambda e: gpt2_prefix_eval.generate_beam(model, tok, embed=e, entry_length=T),
             lambda e: O.generate_beam_ref(sd, e, 5, 13, T)),
            ("greedy", lambda e: gpt2_prefix_eval.generate2(model, tok, embed=e, entry_length=T),
             lambda e: O.generate2_ref(sd, e, 13, T))):
        ref_fn(pe_ref[n]); port_fn(pe_port[n])                      # warm-up caption (lazy init), discarded
        t0 = time.perf_counter(); outs_ref = [ref_fn(pe_ref[i]) for i in range(n)]; t_ref = time.perf_counter() - t0
        t0 = time.perf_counter(); outs_port = [port_fn(pe_port[i]) for i in range(n)]; t_port = time.perf_counter() - t0
        if kind == "beam5":
            same = all(o[0].split() == [str(int(v)) for v in p[0][p[3][0]][:int(p[1][p[3][0]])]] for o, p in zip(outs_ref, outs_port))
        else:
            same = all(o.split() == [str(int(v)) for v in p] for o, p in zip(outs_ref, outs_port))
        res[kind] = {"reference_s_per_caption": round(t_ref / n, 3), "port_s_per_caption": round(t_port / n, 3),
                     "port_over_reference": round(t_port / t_ref, 3), "same_token_ids": bool(same)}
        print(kind, res[kind], flush=True)
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "profiles", "r3_cpu_port_vs_reference.json"), "w"), indent=1)
print(json.dumps(res))
