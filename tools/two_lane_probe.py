#!/usr/bin/env python
"""Feasibility probe (GPU box only): does running two independent half-batches of the beam decode on two HIP
streams (two contexts, two host threads) beat one full batch on one stream?  Prints captions/s for both."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capdec_amd import synth
from capdec_amd.engine import Engine

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 67
    sd = synth.hot_state_dict(42, "mlp", 512, 10)
    pe = (torch.randn(n, 10, 768, generator=torch.Generator().manual_seed(1)) * 0.3).cuda()
    engs = []
    for i in range(lanes):
        e = Engine(0)
        e.load_gpt2(sd)
        engs.append(e)
    streams = [torch.cuda.Stream() for _ in range(lanes)]

    def run_single():
        return engs[0].decode_beam(pe, 13, 5, T)

    def run_lanes():
        outs = [None] * lanes
        bounds = [n * i // lanes for i in range(lanes + 1)]
        def work(i):
            with torch.cuda.stream(streams[i]):
                outs[i] = engs[i].decode_beam(pe[bounds[i]:bounds[i + 1]], 13, 5, T)
        th = [threading.Thread(target=work, args=(i,)) for i in range(lanes)]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        return outs

    for name, fn in (("single", run_single), (f"{lanes}-lane", run_lanes)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            r = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        print(f"{name}: {dt*1e3:.1f} ms  {n/dt:.1f} captions/s", flush=True)
    a = run_single(); b = run_lanes()
    ids = torch.cat([o[0] for o in b]); print("ids equal:", bool((ids == a[0]).all()))

if __name__ == "__main__":
    main()
