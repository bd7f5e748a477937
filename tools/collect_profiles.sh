#!/bin/bash
# Reproduce everything under profiles/ on an MI355X box (one gpurun call):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r4'
# (SHORT=1: only the two bench lines and the rocprofv3 --kernel-trace --stats pass of the bench command, ~8 minutes)
# then copy gpurun_out/<tag>_* into profiles/.  Steps: the driver-style bench line (with the CPU baseline), the same
# command under rocprofv3 --kernel-trace --stats (kernel averages must agree with the hipEvent averages of the bench
# line), the FETCH_SIZE / WRITE_SIZE counters in their own passes (never combined with tracing), summarised per kernel,
# the SQ counters (matrix-pipe busy cycles, wave wait states) of the round-2 GEMM and of the wide single-accumulator
# geometry on a shape whose grid is a whole number of rounds for both, with random and with zero-filled operands (the
# chip runs at its package power cap: zeros show the loop's structure at 2.4 GHz, random what the cap leaves), and the
# 625-caption line (the per-GPU shard of the metric at 8 GPUs).
set -u
TAG=${1:-r4}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$R"
timeout 400 python bench.py --steps 20 --warmup 5 > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
timeout 300 python bench.py --captions 625 --steps 20 --warmup 5 --cpu-seconds 0 > "$OUT/${TAG}_bench_625.json" 2> "$OUT/${TAG}_bench_625.err"
# what one decode step of the 625-caption shard is made of: per-(kernel, grid) table of a traced pass
[ "${SHORT:-0}" = 1 ] || timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/${TAG}_kt625" -- python bench.py --cpu-seconds 0 --no-checks --no-smi --captions 625 --steps 1 --warmup 1 \
    > "$OUT/${TAG}_kt625.json" 2> "$OUT/${TAG}_kt625.err"
[ "${SHORT:-0}" = 1 ] || python tools/trace_summary.py "$OUT/${TAG}_kt625" "$OUT/${TAG}_625_kernels.txt" --title "bench.py --captions 625 --steps 1 --warmup 1 under rocprofv3 --kernel-trace (two passes of 67 steps + mapper + prefill)"
find "$OUT/${TAG}_kt625" -name "*.csv" -delete
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_kt" -- python bench.py --cpu-seconds 0 --cpu-captions 0 --no-checks --no-smi \
    > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_kt.err"
find "$OUT/${TAG}_kt" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_bench_kernel_stats.csv" \;
find "$OUT/${TAG}_kt" -name "*kernel_trace.csv" -delete
if [ "${SHORT:-0}" = 1 ]; then
    tail -c 600 "$OUT/${TAG}_bench.json"; echo; tail -c 300 "$OUT/${TAG}_bench_625.json"; echo
    head -8 "$OUT/${TAG}_bench_kernel_stats.csv" | cut -c1-170
    exit 0
fi
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d "$OUT/${TAG}_pmc_$c" -- python bench.py --cpu-seconds 0 --cpu-captions 0 --no-checks --no-smi --steps 1 --warmup 0 \
        > "$OUT/${TAG}_pmc_$c.log" 2>&1
done
python tools/pmc_summary.py "$OUT" "$OUT/${TAG}_pmc_traffic.json" \
    "python bench.py --cpu-seconds 0 --no-checks --steps 1 --warmup 0 (default workload: 5000 captions, beam 5, T=67)" f16x2 5000 "${TAG}"
find "$OUT" -name "*counter_collection.csv" -delete
export CAPDEC_HOOK_PACKA=1 CAPDEC_HOOK_CACHE=1
for h in 0 2 10 12; do for data in random zeros; do      # (12 = the 256 x 256 ping-pong tile: measurement build only since round 5)
    CAPDEC_MEASURE_LIB=1 CAPDEC_H2W=$h timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
       --output-format csv -d "$OUT/${TAG}_pmc_sq_${h}_${data}" -- python tools/gemm_one.py 16384 2048 768 6 $data > "$OUT/${TAG}_pmc_sq_${h}_${data}.log" 2>&1
done; done
unset CAPDEC_HOOK_PACKA CAPDEC_HOOK_CACHE
python - "$OUT" "$TAG" <<'PY' > "$OUT/${TAG}_pmc_sq_gemm.txt" 2>&1
import csv, glob, collections, sys
out, tag = sys.argv[1], sys.argv[2]
print("SQ counters, GEMM 16384 x 2048 x 768 (a whole number of rounds for both tile shapes), 6 launches each;")
print("h2w 0 = round-2 128x128 two-accumulator kernel, h2w 2 = 256x128 single-accumulator kernel, 10 / 12 = round-4 ping-pong kernels")
print("(256x128 two accumulator sets / 256x256 one set: ONE 8-wavefront block per CU); operands random / zero-filled")
for h in ("0", "2", "10", "12"):
    for data in ("random", "zeros"):
        acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
        for p in glob.glob(f"{out}/{tag}_pmc_sq_{h}_{data}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(p, newline="")):
                k = row["Kernel_Name"].split("(")[0]
                if "gemm" not in k: continue                 # (rocprofv3 leaves template kernels mangled)
                k = k[:72]
                a = acc[k][row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
        for k, cs in acc.items():
            if "pack_planes" in k: continue
            print(f"h2w={h} data={data} {k}")
            for c, (n, s) in sorted(cs.items()):
                print(f"   {c:36s} n={n:4d} avg={s/n:16.1f}")
            if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "SQ_BUSY_CYCLES" in cs:
                print("   mfma_busy / sq_busy = %.3f" % (cs["SQ_VALU_MFMA_BUSY_CYCLES"][1] / cs["SQ_BUSY_CYCLES"][1]))
        for p in glob.glob(f"{out}/{tag}_pmc_sq_{h}_{data}/**/*kernel_trace.csv", recursive=True):
            dur = collections.defaultdict(list)
            for row in csv.DictReader(open(p, newline="")):
                dur[row["Kernel_Name"].split("(")[0][:72]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            for k, v in dur.items():
                if "gemm" in k: print("   duration_ns", k, "n=%d avg=%.0f min=%d" % (len(v), sum(v) / len(v), min(v)))
PY
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*kernel_trace.csv" -delete
tail -c 600 "$OUT/${TAG}_bench.json"; echo
tail -c 300 "$OUT/${TAG}_bench_625.json"; echo
head -8 "$OUT/${TAG}_bench_kernel_stats.csv" | cut -c1-170
cat "$OUT/${TAG}_pmc_sq_gemm.txt"
cat "$OUT/${TAG}_pmc_traffic.json" | head -c 1500
