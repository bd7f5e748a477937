#!/bin/bash
# Round 6, call 10: decode contexts up to GPT-2's 1024 positions (beam step state in dynamic LDS, per-launch LDS limits in the
# attention launchers): the whole GPU suite, then the metric workload and the 625-caption shard (no regression check)
set -u
TAG=r6
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
export TMPDIR=/tmp
rm -f "$OUT/parity_counts.txt"
SECONDS=0; timeout 1300 python -m pytest tests/ -x -q -m gpu --durations=25 > "$OUT/${TAG}_pytest_gpu.txt" 2>&1
echo "suite wall seconds: $SECONDS" | tee -a "$OUT/${TAG}_pytest_gpu.txt"; tail -5 "$OUT/${TAG}_pytest_gpu.txt" | cut -c1-160
cp "$OUT/parity_counts.txt" "$OUT/${TAG}_parity_counts.txt" 2>/dev/null
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a "$OUT/${TAG}_pytest_gpu.txt"
B="timeout 400 python bench.py --cpu-captions 0 --no-checks --cpu-seconds 0"
$B --steps 5 --warmup 2 > "$OUT/tmp5000.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/tmp5000.json'));k=r['kernels'];print('5000 (5 steps):',r['value'],r['ms_per_step'],'select',k['select']['avg_ms'],'attn',k['attn_decode']['avg_ms'])"
$B --captions 625 --steps 20 --warmup 5 > "$OUT/tmp625.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/tmp625.json'));k=r['kernels'];print('625:',r['value'],r['ms_per_step'],'select',k['select']['avg_ms'])"
