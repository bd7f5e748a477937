#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests -m gpu -x -q -k "prompt or make_preds or capi_rccl or c_host or f16x2_split" > "$OUT/r2_pytest_new.txt" 2>&1
tail -15 "$OUT/r2_pytest_new.txt"
CAPDEC_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 \
   bench.py --gpus 1 --captions 625 --cpu-seconds 0 --steps 2 --warmup 1 > "$OUT/r2_dist1b.json" 2> "$OUT/r2_dist1b.err"
python -c "import json;r=json.load(open('$OUT/r2_dist1b.json'));print(r['value'], r['capi_collective'], r['scaling_check'])"; tail -3 "$OUT/r2_dist1b.err"
