#!/bin/bash
# Reproduce everything under profiles/ on an MI355X box (one gpurun call):
#   gpurun --timeout 1800 -- 'bash tools/collect_profiles.sh r2'
# then copy gpurun_out/<tag>_* into profiles/.  Steps: the default bench line (with the CPU baseline), the same command
# under rocprofv3 --kernel-trace --stats (kernel averages must agree with the hipEvent averages of the bench line),
# the FETCH_SIZE / WRITE_SIZE counters in their own passes (never combined with tracing), summarised per kernel, and
# the SQ counters (matrix-pipe busy cycles, wave wait states, LDS) of the dominant GEMM on the decode loop's qkv shape.
set -u
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$R"
python bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_kt" -- python bench.py --cpu-seconds 0 \
    > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_kt.err"
find "$OUT/${TAG}_kt" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_bench_kernel_stats.csv" \;
find "$OUT/${TAG}_kt" -name "*domain_stats.csv" -exec cp {} "$OUT/${TAG}_bench_domain_stats.csv" \;
find "$OUT/${TAG}_kt" -name "*kernel_trace.csv" -delete
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$OUT/${TAG}_pmc_$c" -- python bench.py --cpu-seconds 0 --steps 1 --warmup 0 \
        > "$OUT/${TAG}_pmc_$c.log" 2>&1
done
python tools/pmc_summary.py "$OUT" "$OUT/${TAG}_pmc_traffic.json" \
    "python bench.py --cpu-seconds 0 --steps 1 --warmup 0 (default workload: 5000 captions, beam 5, T=67)" f16x2 5000
find "$OUT" -name "*counter_collection.csv" -delete
export CAPDEC_HOOK_PACKA=1 CAPDEC_HOOK_CACHE=1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
   --output-format csv -d "$OUT/${TAG}_pmc_sq" -- python tools/gemm_one.py 25000 2304 768 6 > "$OUT/${TAG}_pmc_sq.log" 2>&1
unset CAPDEC_HOOK_PACKA CAPDEC_HOOK_CACHE
python - "$OUT" "$TAG" <<'PY' > "$OUT/${TAG}_pmc_sq_gemm.txt" 2>&1
import csv, glob, collections, sys
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for p in glob.glob(f"{out}/{tag}_pmc_sq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(p, newline="")):
        k = row["Kernel_Name"].split("(")[0][-60:]
        a = acc[k][row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
for k, cs in acc.items():
    print(k)
    for c, (n, s) in sorted(cs.items()):
        print(f"   {c:36s} n={n:4d} avg={s/n:16.1f}")
for p in glob.glob(f"{out}/{tag}_pmc_sq/**/*kernel_trace.csv", recursive=True):
    dur = collections.defaultdict(list)
    for row in csv.DictReader(open(p, newline="")):
        dur[row["Kernel_Name"].split("(")[0][-60:]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, v in dur.items():
        print("duration_ns", k, "n=%d avg=%.0f min=%d" % (len(v), sum(v) / len(v), min(v)))
PY
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*kernel_trace.csv" -delete
tail -c 700 "$OUT/${TAG}_bench.json"; echo
head -6 "$OUT/${TAG}_bench_kernel_stats.csv" | cut -c1-170
cat "$OUT/${TAG}_pmc_sq_gemm.txt"
