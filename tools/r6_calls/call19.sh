#!/bin/bash
# Round 6, call 19: FETCH_SIZE / WRITE_SIZE of the configs[3] line's kernels (separate --pmc passes) -> the `traffic` of its roofline
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT/pmct"; cd /tmp && export TMPDIR=/tmp && cd "$R"
CMD="python bench.py --workload text_embed --captions 20000 --gemm-mode f16 --steps 1 --warmup 0 --cpu-seconds 0 --cpu-captions 0 --no-checks --no-smi"
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmct/$c" -- $CMD > "$OUT/pmct/$c.log" 2>&1
done
python tools/pmc_summary.py "$OUT/pmct" "$OUT/r6_text_f16_pmc_traffic.json" "$CMD" f16 20000 r6
rm -rf "$OUT/pmct"
