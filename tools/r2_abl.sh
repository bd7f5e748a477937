#!/bin/bash
export TMPDIR=/tmp
for a in 0 5 6 2; do
echo "abl $a"
CAPDEC_H2_ABL=$a CAPDEC_HOOK_PACKA=1 CAPDEC_HOOK_CACHE=1 timeout 300 python tools/gemm_bench.py 25000 3125 2>/dev/null | python -c "
import json,sys; b=json.load(sys.stdin); print({k:(v['ms'],v['tflops']) for k,v in b.items() if k!='mode' and '50257' not in k})"
done
