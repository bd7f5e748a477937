#!/usr/bin/env python
"""GPU box: correctness (vs fp64) and timing of the f16x2 GEMM kernels on the decode loop's shapes, for the geometry
CAPDEC_H2W selects (0 = round-2 kernel, 2.. = the wide single-accumulator kernels of gemm_h2w.hip).
usage: CAPDEC_H2W=<n> CAPDEC_HOOK_PACKA=1 h2w_probe.py [M ...]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CAPDEC_HOOK_PACKA", "1")
import torch
from capdec_amd.engine import Engine


def main():
    eng = Engine(0, measure=os.environ.get("CAPDEC_MEASURE_LIB") == "1")   # CAPDEC_MEASURE_LIB=1: the -DCAPDEC_MEASURE build (ablation knobs)
    res = {"h2w": os.environ.get("CAPDEC_H2W", "default"), "mode": eng.gemm_mode()}
    g = torch.Generator().manual_seed(1)
    worst = 0.0
    for (m, n, k) in [(1000, 1531, 768), (257, 768, 3072), (3125, 2304, 768), (130, 130, 64), (513, 3072, 768)]:
        a = torch.randn(m, k, generator=g)
        bt = torch.randn(n, k, generator=g) * 0.1
        bias, resid = torch.randn(n, generator=g), torch.randn(m, n, generator=g)
        ref = a.double() @ bt.double().t()
        scale = a.abs().double() @ bt.abs().double().t()
        out = eng.gemm(a, bt).cpu().double()
        worst = max(worst, float(((out - ref).abs() / scale).max()))
        out2 = eng.gemm(a, bt, bias=bias, resid=resid, act=2).cpu().double()
        ref2 = torch.relu(ref + bias.double()) + resid.double()
        worst = max(worst, float(((out2 - ref2).abs() / (scale + 1)).max()))
    res["max_err_over_scale"] = worst
    # timing: both operands resident (packed once).  The knobs are read when a context is created (config.h), so the timed
    # loops get their own context
    eng.close()
    os.environ["CAPDEC_HOOK_CACHE"] = "1"
    eng = Engine(0, measure=os.environ.get("CAPDEC_MEASURE_LIB") == "1")
    Ms = [int(v) for v in sys.argv[1:]] or [25000]
    custom = [tuple(int(x) for x in t.split(",")) for t in os.environ.get("PROBE_SHAPES", "").split(";") if t]
    for M in ([0] if custom else Ms):
        for (m, n, k) in custom or ([(M, 2304, 768), (M, 768, 768), (M, 3072, 768), (M, 768, 3072)] + ([(M, 50257, 768)] if os.environ.get('PROBE_LMHEAD') else [])):
            a = (torch.rand(m, k, generator=g) * 2 - 1).cuda()
            bt = (torch.rand(n, k, generator=g) * 2 - 1).cuda()
            for _ in range(2):
                out = eng.gemm(a, bt)
            torch.cuda.synchronize()
            iters = 5 if n > 10000 else 20
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
            evs[0].record()
            for i_ in range(iters):                 # back-to-back launches, one event between each pair: the MEDIAN interval
                out = eng.gemm(a, bt)               # (a one-off stall of the box does not end up in the figure)
                evs[i_ + 1].record()
            torch.cuda.synchronize()
            ms = sorted(evs[i_].elapsed_time(evs[i_ + 1]) for i_ in range(iters))[iters // 2]
            res[f"{m}x{n}x{k}"] = dict(ms=round(ms, 4), tflops=round(2.0 * m * n * k / ms / 1e9, 1))
            del a, bt, out
    print(json.dumps(res))


if __name__ == "__main__":
    main()
