#!/bin/bash
# round-2 probe 1 (one gpurun call): SQ counters of the dominant GEMM, mid-size (625-caption) baseline, two-lane
# probe at 625 captions, RCCL path with one rank.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$R"
export CAPDEC_HOOK_PACKA=1 CAPDEC_HOOK_CACHE=1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
   --output-format csv -d "$OUT/r2_pmc_sq" -- python tools/gemm_one.py 25000 2304 768 6 > "$OUT/r2_pmc_sq.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM \
   --output-format csv -d "$OUT/r2_pmc_sq2" -- python tools/gemm_one.py 25000 2304 768 6 > "$OUT/r2_pmc_sq2.log" 2>&1
unset CAPDEC_HOOK_PACKA CAPDEC_HOOK_CACHE
python - <<'PY' > "$OUT/r2_pmc_sq_summary.txt" 2>&1
import csv, glob, collections
for d in ("r2_pmc_sq", "r2_pmc_sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for p in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(p, newline="")):
            k = row["Kernel_Name"].split("(")[0][-60:]
            a = acc[k][row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
    for k, cs in acc.items():
        print(d, k)
        for c, (n, s) in sorted(cs.items()):
            print(f"   {c:36s} n={n:4d} avg={s/n:16.1f}")
PY
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
python bench.py --captions 625 --cpu-seconds 0 --steps 3 --warmup 1 > "$OUT/r2_base_625.json" 2> "$OUT/r2_base_625.err"
timeout 300 python tools/two_lane_probe.py 625 2 > "$OUT/r2_two_lane_625.txt" 2>&1
timeout 300 python tools/two_lane_probe.py 625 3 >> "$OUT/r2_two_lane_625.txt" 2>&1
CAPDEC_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
   bench.py --gpus 1 --captions 625 --cpu-seconds 0 --steps 2 --warmup 1 > "$OUT/r2_dist1_625.json" 2> "$OUT/r2_dist1_625.err"
cat "$OUT/r2_pmc_sq_summary.txt"; cut -c1-400 "$OUT/r2_base_625.json"; cat "$OUT/r2_two_lane_625.txt"; cut -c1-300 "$OUT/r2_dist1_625.json"; tail -3 "$OUT/r2_dist1_625.err"
