#!/bin/bash
# in-loop A/B of product knobs on one box:  gpurun --timeout 900 -- 'bash tools/knob_sweep.sh "CAPDEC_H2W=2 CAPDEC_H2W=8" [captions]'
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
CAPS=${2:-5000}
B="python bench.py --cpu-seconds 0 --no-checks --steps 3 --warmup 1 --captions $CAPS"
for kv in "X=0" $1; do
    echo "== $kv"
    env $kv $B 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=r['kernels']
print(r['value'], 'gemm', k['gemm_f16x2p']['avg_ms'], k['gemm_f16x2p'].get('tflops'), 'lmhead', k['gemm_f16x2p_lmhead_topk']['avg_ms'], 'attn', k['attn_decode']['avg_ms'], '2nd', (k.get('lmhead_second_pass') or {}).get('avg_ms'), 'select', k['select']['avg_ms'], 'W', r['power']['watts'], 'MHz', r['power']['sclk_mhz'])"
done
