#!/usr/bin/env python
"""GPU box: per-block phase stamps of the ping-pong GEMM (CAPDEC_PP_STAMPS): where a short launch spends its time.
usage: CAPDEC_H2W=10 python tools/pp_stamps.py "3125,2304,768;3125,3072,768" """
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = "/tmp/pp_stamps.txt"
os.environ["CAPDEC_PP_STAMPS"] = path
os.environ["CAPDEC_MEASURE_LIB"] = "1"
os.environ.setdefault("CAPDEC_HOOK_PACKA", "1")
os.environ["CAPDEC_HOOK_CACHE"] = "1"
os.environ.setdefault("CAPDEC_SPLITK", "0")
import numpy as np
import torch
from capdec_amd.engine import Engine

eng = Engine(0, measure=os.environ.get("CAPDEC_MEASURE_LIB") == "1")   # CAPDEC_MEASURE_LIB=1: the -DCAPDEC_MEASURE build (ablation knobs)
g = torch.Generator().manual_seed(0)
for t in sys.argv[1].split(";"):
    m, n, k = (int(v) for v in t.split(","))
    a = (torch.rand(m, k, generator=g) * 2 - 1).cuda()
    bt = (torch.rand(n, k, generator=g) * 2 - 1).cuda()
    if os.path.exists(path):
        os.remove(path)
    evs = []
    for _ in range(6):
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        eng.gemm(a, bt)
        e0.record()
        evs.append((s0, e0))
    torch.cuda.synchronize()
    print("event-timed launches (us, includes the stamp read-back sync):", [round(a_.elapsed_time(b_) * 1e3, 1) for a_, b_ in evs])
    launches, cur = [], None
    for line in open(path):
        if line.startswith("launch"):
            cur = []
            launches.append((line.strip(), cur))
        else:
            cur.append([int(v) for v in line.split()])
    for hdr, rows in launches[-2:]:
        s = np.array(rows, dtype=np.int64)
        s = s[s[:, 0] > 0]
        t0 = s[:, 0].min()
        us = (s - t0) / 100.0          # 100 MHz wall clock
        print(hdr)
        for name, col in (("entry", 0), ("first tile landed", 1), ("main loop done", 2), ("epilogue done", 3)):
            v = us[:, col]
            print(f"   {name:18s} min {v.min():7.2f}  median {np.median(v):7.2f}  max {v.max():7.2f} us after the first block's entry")
        print(f"   per block: prologue {np.median(us[:,1]-us[:,0]):6.2f}  main loop {np.median(us[:,2]-us[:,1]):6.2f}  epilogue {np.median(us[:,3]-us[:,2]):6.2f} us (medians)")
