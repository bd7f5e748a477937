#!/bin/bash
# round-2 probe 3: SQ counters (+ kernel durations) of the f16x2 GEMM, ring-depth variants
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$R"
export CAPDEC_HOOK_PACKA=1 CAPDEC_HOOK_CACHE=1
for ns in 3 4 5; do
  CAPDEC_H2_NS=$ns timeout 200 python tools/gemm_bench.py 25000 3125 > "$OUT/r2_gemm_f16x2_ns$ns.json" 2>/dev/null
  cat "$OUT/r2_gemm_f16x2_ns$ns.json"
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
   --output-format csv -d "$OUT/r2_pmc_h2" -- python tools/gemm_one.py 25000 2304 768 6 > "$OUT/r2_pmc_h2.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM \
   --output-format csv -d "$OUT/r2_pmc_h2b" -- python tools/gemm_one.py 25000 2304 768 6 > "$OUT/r2_pmc_h2b.log" 2>&1
python - <<'PY' > "$OUT/r2_pmc_h2_summary.txt" 2>&1
import csv, glob, collections
for d in ("r2_pmc_h2", "r2_pmc_h2b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for p in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(p, newline="")):
            k = row["Kernel_Name"].split("(")[0][-60:]
            a = acc[k][row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
    for k, cs in acc.items():
        print(d, k)
        for c, (n, s) in sorted(cs.items()):
            print(f"   {c:36s} n={n:4d} avg={s/n:16.1f}")
    for p in glob.glob(f"gpurun_out/{d}/**/*kernel_trace.csv", recursive=True):
        dur = collections.defaultdict(list)
        for row in csv.DictReader(open(p, newline="")):
            dur[row["Kernel_Name"].split("(")[0][-60:]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        for k, v in dur.items():
            print(d, "duration_ns", k, "n=%d avg=%.0f min=%d" % (len(v), sum(v) / len(v), min(v)))
PY
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
cat "$OUT/r2_pmc_h2_summary.txt"
