#!/usr/bin/env python
"""Run ONE GEMM shape a few times (for rocprofv3 --pmc passes).  usage: gemm_one.py M N K [iters] [random|zeros]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capdec_amd.engine import Engine
eng = Engine(0, measure=os.environ.get("CAPDEC_MEASURE_LIB") == "1")   # CAPDEC_MEASURE_LIB=1: the -DCAPDEC_MEASURE build (ablation knobs)
m, n, k = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
data = sys.argv[5] if len(sys.argv) > 5 else "random"
g = torch.Generator().manual_seed(0)
if data == "zeros":
    a, bt = torch.zeros(m, k).cuda(), torch.zeros(n, k).cuda()
else:
    a = (torch.rand(m, k, generator=g) * 2 - 1).cuda()
    bt = (torch.rand(n, k, generator=g) * 2 - 1).cuda()
for _ in range(iters):
    out = eng.gemm(a, bt)
torch.cuda.synchronize()
