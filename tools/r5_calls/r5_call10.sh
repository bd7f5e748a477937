#!/bin/bash
# round 5, call 10: the one-plane 256 x 128 kernel (gemm_x1w_kernel) -- isolated timing, parity (forced at every size), A/B in the loop
# (the variant it measures -- gemm_x1w_kernel behind CAPDEC_X1_WIDE -- was removed after this call: profiles/r5_x1_pingpong_ab.txt, "call 10";
#  the same holds for CAPDEC_PP_X1 in tools/r5_calls/r5_call3.sh / r5_call4.sh and CAPDEC_ATT_G16 in tools/r5_calls/r5_call5.sh)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
echo "== isolated: gemm_bench-style timing of the four decode shapes, CAPDEC_X1_WIDE 0 vs 2"
for w in 0 2; do for M in 5000 25000; do
  CAPDEC_X1_WIDE=$w CAPDEC_GEMM_MODE=bf16 CAPDEC_HOOK_PACKA=1 CAPDEC_HOOK_CACHE=1 python - $M <<'PY'
import sys, os, torch, time
sys.path.insert(0, os.getcwd())
from capdec_amd.engine import Engine
M = int(sys.argv[1]); e = Engine(0); g = torch.Generator().manual_seed(0)
for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    a = (torch.rand(M, K, generator=g) * 2 - 1).cuda(); bt = (torch.rand(N, K, generator=g) * 2 - 1).cuda()
    for _ in range(3): e.gemm(a, bt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): e.gemm(a, bt)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"X1_WIDE={os.environ['CAPDEC_X1_WIDE']} M={M} N={N} K={K}: {dt*1e6:.1f} us  {2*M*N*K/dt/1e12:.0f} TFLOP/s")
PY
done; done
echo "== parity: forced at every size (child) + default planner"
timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "(test_wide_single and bf16_w256x128) or bf16_mode" --durations=5 2>&1 | tail -10 | tee "$OUT/r5_x1w_tests.txt"
tail -3 "$OUT/parity_counts.txt" | cut -c1-400
summ() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernels',{})
print(d['value'], d['ms_per_step'], {n:(round(v.get('avg_ms',0)*1000,1), v.get('tflops')) for n,v in k.items() if 'gemm_x1' in n or 'attn_dec' in n})"; }
for w in 0 1 2; do
    echo "-- greedy_mlp bf16 CAPDEC_X1_WIDE=$w"
    CAPDEC_X1_WIDE=$w timeout 200 python bench.py --workload greedy_mlp --gemm-mode bf16 --steps 6 --warmup 2 --cpu-seconds 0 --no-checks 2>/dev/null | tee "$OUT/r5_greedy_bf16_x1w_$w.json" | summ
done
for w in 0 1; do
    echo "-- beam bf16 CAPDEC_X1_WIDE=$w"
    CAPDEC_X1_WIDE=$w timeout 300 python bench.py --gemm-mode bf16 --steps 3 --warmup 1 --cpu-seconds 0 --no-checks 2>/dev/null | tee "$OUT/r5_beam_bf16_x1w_$w.json" | summ
done
