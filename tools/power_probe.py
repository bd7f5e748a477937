#!/usr/bin/env python
"""GPU box: is the fp32-accurate GEMM power / clock bound?  Same kernel, same shape, operands of different bit entropy:
random fp32 (both planes full), fp16-exact values (low plane all zero), small integers (few mantissa bits), zeros.
Also samples sclk / power through rocm-smi while a sustained loop runs.  usage: CAPDEC_H2W=<n> power_probe.py"""
import sys, os, json, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CAPDEC_HOOK_PACKA"] = "1"
os.environ["CAPDEC_HOOK_CACHE"] = "1"
import torch
from capdec_amd.engine import Engine

eng = Engine(0, measure=os.environ.get("CAPDEC_MEASURE_LIB") == "1")   # CAPDEC_MEASURE_LIB=1: the -DCAPDEC_MEASURE build (ablation knobs)
g = torch.Generator().manual_seed(1)
m, n, k = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (25000, 2304, 768)))
kinds = sys.argv[4].split(",") if len(sys.argv) > 4 else ["random", "fp16_exact", "small_int", "zeros", "random"]
res = {"h2w": os.environ.get("CAPDEC_H2W", "default"), "shape": [m, n, k]}
keep = []


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        c = d.get("card0", {})
        return {kk: vv for kk, vv in c.items() if "sclk" in kk.lower() or "power" in kk.lower() or "mclk" in kk.lower()}
    except Exception as e:  # noqa
        return {"error": str(e)[:100]}


def make(kind):
    if kind == "random":
        return (torch.rand(m, k, generator=g) * 2 - 1), (torch.rand(n, k, generator=g) * 2 - 1)
    if kind == "fp16_exact":
        return (torch.rand(m, k, generator=g) * 2 - 1).half().float(), (torch.rand(n, k, generator=g) * 2 - 1).half().float()
    if kind == "small_int":
        return torch.randint(-3, 4, (m, k), generator=g).float(), torch.randint(-3, 4, (n, k), generator=g).float()
    return torch.zeros(m, k), torch.zeros(n, k)


for kind in kinds:
    a, bt = make(kind)
    a, bt = a.cuda(), bt.cuda()
    for _ in range(3):
        out = eng.gemm(a, bt)
    torch.cuda.synchronize()
    samples = []
    stop = False

    def sampler():
        while not stop:
            samples.append(smi())
            time.sleep(0.3)
    th = threading.Thread(target=sampler)
    th.start()
    iters = max(200, int(1.0e12 / (m * n * k)))      # ~1 s sustained
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        out = eng.gemm(a, bt)
    e.record(); torch.cuda.synchronize()
    stop = True
    th.join()
    ms = s.elapsed_time(e) / iters
    key = kind if kind not in res else kind + "_again"
    res[key] = dict(ms=round(ms, 4), tflops=round(2.0 * m * n * k / ms / 1e9, 1), smi=samples[1:4])
    keep.append((a, bt))          # (the engine's plane cache is keyed by address: never let torch recycle one)
    del out
print(json.dumps(res))
