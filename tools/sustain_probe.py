#!/usr/bin/env python
"""GPU box: why do the GEMMs run at ~315 TFLOP/s inside the decode loop and ~360 in a short isolated loop?
(1) one shape, 25 s sustained, TFLOP/s / sclk / power / temperature every ~2 s (thermal drift?);
(2) the decode loop's GEMM sequence with 12 distinct weight sets (cold weights, activations from the previous GEMM)."""
import sys, os, json, subprocess, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CAPDEC_HOOK_PACKA"] = "1"
os.environ["CAPDEC_HOOK_CACHE"] = "1"
import torch
from capdec_amd.engine import Engine

eng = Engine(0, measure=os.environ.get("CAPDEC_MEASURE_LIB") == "1")   # CAPDEC_MEASURE_LIB=1: the -DCAPDEC_MEASURE build (ablation knobs)
g = torch.Generator().manual_seed(1)


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20).stdout
        c = json.loads(out).get("card0", {})
        keep = {}
        for kk, vv in c.items():
            lk = kk.lower()
            if "sclk clock speed" in lk or "power" in lk or "junction" in lk or "hotspot" in lk or "edge" in lk:
                keep[kk] = vv
        return keep
    except Exception as e:  # noqa
        return {"error": str(e)[:100]}


def timed(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


res = {"h2w": os.environ.get("CAPDEC_H2W", "default")}
m, n, k = 25000, 2304, 768
a = (torch.rand(m, k, generator=g) * 2 - 1).cuda()
bt = (torch.rand(n, k, generator=g) * 2 - 1).cuda()
eng.gemm(a, bt); torch.cuda.synchronize()
trace = []
t0 = time.time()
while time.time() - t0 < 25:
    ms = timed(lambda: eng.gemm(a, bt), 4000)
    trace.append(dict(t=round(time.time() - t0, 1), tflops=round(2.0 * m * n * k / ms / 1e9, 1), smi=smi()))
res["sustained_one_shape"] = trace
# decode-like sequence: 12 layers x (qkv, proj, fc, proj2) with distinct weights; A operands distinct per GEMM kind
M = 25000
acts = {kk: (torch.rand(M, kk, generator=g) * 2 - 1).cuda() for kk in (768, 3072)}
layers = []
for l in range(12):
    layers.append([(torch.rand(nn, kk, generator=g) * 2 - 1).mul_(0.05).cuda() for (nn, kk) in ((2304, 768), (768, 768), (3072, 768), (768, 3072))])
def sweep():
    for w in layers:
        for bt_ in w:
            eng.gemm(acts[bt_.shape[1]], bt_)
sweep(); torch.cuda.synchronize()
flops = 12 * 2.0 * M * (2304 * 768 + 768 * 768 + 3072 * 768 + 768 * 3072)
seq = []
for rep in range(5):
    ms = timed(sweep, 20)
    seq.append(dict(ms_per_sweep=round(ms, 3), tflops=round(flops / ms / 1e9, 1), smi=smi()))
res["decode_like_sequence"] = seq
print(json.dumps(res))
