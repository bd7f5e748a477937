#!/usr/bin/env python
"""Micro-benchmark of the packed-operand MFMA GEMMs on the shapes the decode loop issues (GPU box only).
usage: CAPDEC_HOOK_PACKA=1 CAPDEC_HOOK_CACHE=1 gemm_bench.py [M ...]   (mode from CAPDEC_GEMM_MODE)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capdec_amd.engine import Engine

def main():
    eng = Engine(0, measure=os.environ.get("CAPDEC_MEASURE_LIB") == "1")   # CAPDEC_MEASURE_LIB=1: the -DCAPDEC_MEASURE build (ablation knobs)
    Ms = [int(v) for v in sys.argv[1:]] or [25000]
    g = torch.Generator().manual_seed(0)
    res = {"mode": eng.gemm_mode()}
    for M in Ms:
        shapes = [(M, 2304, 768), (M, 768, 768), (M, 3072, 768), (M, 768, 3072), (M, 50257, 768)]
        if M == Ms[0]:
            shapes.append((4096, 4096, 4096))
        for (m, n, k) in shapes:
            a = (torch.rand(m, k, generator=g) * 2 - 1).cuda()
            bt = (torch.rand(n, k, generator=g) * 2 - 1).cuda()
            for _ in range(2):
                out = eng.gemm(a, bt)
            torch.cuda.synchronize()
            iters = 5 if n > 10000 else 20
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                out = eng.gemm(a, bt)
            e.record(); torch.cuda.synchronize()
            ms = s.elapsed_time(e) / iters
            res[f"{m}x{n}x{k}"] = dict(ms=round(ms, 4), tflops=round(2.0 * m * n * k / ms / 1e9, 1))
            del a, bt, out
    print(json.dumps(res))

if __name__ == "__main__":
    main()
