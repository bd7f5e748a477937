#!/usr/bin/env python
"""GEMM at decode shapes with HOT vs COLD weights (GPU box only): the decode loop walks 12 layers x 4 matrices (340 MB of
packed weights > the 256 MB Infinity Cache), so in-chain every weight tile comes from HBM; a micro-benchmark that repeats
one matrix reads it from L2 / Infinity Cache.  usage: CAPDEC_HOOK_PACKA=1 CAPDEC_HOOK_CACHE=1 gemm_cold.py M"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capdec_amd.engine import Engine
eng = Engine(0, measure=os.environ.get("CAPDEC_MEASURE_LIB") == "1")   # CAPDEC_MEASURE_LIB=1: the -DCAPDEC_MEASURE build (ablation knobs)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 3125
g = torch.Generator().manual_seed(0)
res = {}
for (n, k) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    a = (torch.rand(M, k, generator=g) * 2 - 1).cuda()
    ws = [(torch.rand(n, k, generator=g) * 2 - 1).cuda() for _ in range(48)]      # 48 x 7-9 MB packed: > Infinity Cache
    for w in ws:
        eng.gemm(a, w)
    torch.cuda.synchronize()
    for name, seq in (("hot", [ws[0]] * 48), ("cold", ws)):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for w in seq:
            eng.gemm(a, w)
        e.record(); torch.cuda.synchronize()
        res[f"{M}x{n}x{k}_{name}_us"] = round(s.elapsed_time(e) / len(seq) * 1e3, 2)
    del ws
print(json.dumps(res))
