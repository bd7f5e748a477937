#!/bin/bash
# Round 6, call 9: the fp32 decode attention with a three-deep LDS-DMA ring (CAPDEC_ATT_RING=3: two iterations in flight
# per wavefront, three blocks per CU) against the double buffer, same box: parity tests under the knob, then A/B at 5000 and
# 625 captions (value, ms per pass, attention ms per launch)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
export TMPDIR=/tmp
CAPDEC_ATT_RING=3 timeout 900 python -m pytest tests/ -x -q -m gpu -k "decode_small or decode_tiny or midsize or large_launch or compact or stop or p40 or long_context" 2>&1 | tail -3 | cut -c1-160
B="timeout 400 python bench.py --cpu-captions 0 --no-checks --cpu-seconds 0 --no-smi"
{
for i in 1 2; do for rg in 2 3; do
  CAPDEC_ATT_RING=$rg $B --steps 5 --warmup 2 > "$OUT/tmp.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/tmp.json'));k=r['kernels'];print('5000 captions ring $rg:',r['value'],r['ms_per_step'],'attn_decode avg ms',k['attn_decode']['avg_ms'],'gemm',k['gemm_f16x2p']['avg_ms'])"
done; done
for i in 1 2; do for rg in 2 3; do
  CAPDEC_ATT_RING=$rg $B --captions 625 --steps 10 --warmup 3 > "$OUT/tmp.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/tmp.json'));k=r['kernels'];print(' 625 captions ring $rg:',r['value'],r['ms_per_step'],'attn_decode avg ms',k['attn_decode']['avg_ms'])"
done; done
} | tee "$OUT/r6_att_ring_ab.txt"
