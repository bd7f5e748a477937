#!/usr/bin/env python
"""GPU box: does overlapping the power-bound GEMM phases of one half-batch with the HBM-bound attention phases of the
other pay?  Two engines (two contexts, two HIP streams, weights replicated), each decoding half of the captions from
its own host thread, against one engine decoding all of them.  usage: overlap_probe.py [captions] [ways]"""
import sys, os, json, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capdec_amd import synth
from capdec_amd.gpt2_prefix import ClipCaptionModel, MappingType
from capdec_amd.predictions_runner import caption_ids

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
ways = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
sd = synth.hot_state_dict(42, "transformer_encoder", 512, 10)
models = []
for i in range(ways):
    m = ClipCaptionModel(10, clip_length=10, prefix_dim=512, num_layers=8, mapping_type=MappingType.TransformerEncoder).to(dev).eval()
    m.load_state_dict(sd)
    models.append(m)
emb = synth.synthetic_clip_embeddings(n, 512, seed=0, normalize=False).to(dev)
streams = [torch.cuda.Stream(dev) for _ in range(ways)]


def run_one(i, lo, hi, out):
    with torch.cuda.stream(streams[i]):
        ids, lens, scores = caption_ids(models[i], emb[lo:hi], 13, beam=True, beam_size=5, entry_length=67)
        streams[i].synchronize()
    out[i] = (ids, lens, scores)


def single():
    out = [None]
    run_one(0, 0, n, out)
    return out[0]


def multi():
    out = [None] * ways
    bounds = [(n * i // ways, n * (i + 1) // ways) for i in range(ways)]
    th = [threading.Thread(target=run_one, args=(i, lo, hi, out)) for i, (lo, hi) in enumerate(bounds)]
    for t in th: t.start()
    for t in th: t.join()
    return out


res = {"captions": n, "ways": ways}
single(); torch.cuda.synchronize()
t0 = time.perf_counter(); ref = single(); torch.cuda.synchronize(); res["single_s"] = round(time.perf_counter() - t0, 4)
multi(); torch.cuda.synchronize()
t0 = time.perf_counter(); out = multi(); torch.cuda.synchronize(); res["multi_s"] = round(time.perf_counter() - t0, 4)
t0 = time.perf_counter(); out = multi(); torch.cuda.synchronize(); res["multi_s_again"] = round(time.perf_counter() - t0, 4)
ids = torch.cat([o[0] for o in out])
res["captions_per_s_single"] = round(n / res["single_s"], 1)
res["captions_per_s_multi"] = round(n / min(res["multi_s"], res["multi_s_again"]), 1)
res["identical_captions"] = round(float((ids == ref[0]).flatten(1).all(1).float().mean()), 4)
print(json.dumps(res))
