#!/bin/bash
# SQ counters of the one-plane (bf16) GEMM kernel on the qkv shape of BASELINE configs[1] (5000 rows) and of the beam-5
# launches (25 000 rows), random and zero operands, next to the f16x2 kernel on the same shapes:
#   gpurun --timeout 900 -- 'bash tools/r5_x1_counters.sh'   ->  gpurun_out/r5_pmc_sq_x1.txt
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$R"
export CAPDEC_HOOK_PACKA=1 CAPDEC_HOOK_CACHE=1
for mode in bf16 f16x2; do for M in 5000 25000; do for data in random zeros; do
    CAPDEC_GEMM_MODE=$mode timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
        --output-format csv -d "$OUT/r5_sq_${mode}_${M}_${data}" -- python tools/gemm_one.py $M 2304 768 8 $data > "$OUT/r5_sq_${mode}_${M}_${data}.log" 2>&1
done; done; done
for mode in bf16; do for M in 5000 25000; do
    CAPDEC_GEMM_MODE=$mode timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM \
        --output-format csv -d "$OUT/r5_sq2_${mode}_${M}" -- python tools/gemm_one.py $M 2304 768 8 random > "$OUT/r5_sq2_${mode}_${M}.log" 2>&1
done; done
python - "$OUT" <<'PY' > "$OUT/r5_pmc_sq_x1.txt" 2>&1
import csv, glob, collections, sys, os
out = sys.argv[1]
print("SQ counters, GEMM M x 2304 x 768 (the qkv projection), 8 launches each, operands packed and resident (CAPDEC_HOOK_PACKA / _CACHE);")
print("bf16 = gemm_x1_kernel (one plane, 1 MFMA per product), f16x2 = the two-plane kernels (3 MFMAs per product); operands random / zero-filled")
for d in sorted(glob.glob(f"{out}/r5_sq_*") + glob.glob(f"{out}/r5_sq2_*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for p in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(p, newline="")):
            k = row["Kernel_Name"].split("(")[0]
            if "gemm" not in k: continue
            a = acc[k[:80]][row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
    for k, cs in acc.items():
        print(os.path.basename(d), k)
        for c, (n, s) in sorted(cs.items()):
            print(f"   {c:36s} n={n:4d} avg={s/n:16.1f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "SQ_BUSY_CYCLES" in cs:
            print("   mfma_busy / sq_busy = %.3f" % (cs["SQ_VALU_MFMA_BUSY_CYCLES"][1] / cs["SQ_BUSY_CYCLES"][1]))
    for p in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
        dur = collections.defaultdict(list)
        for row in csv.DictReader(open(p, newline="")):
            dur[row["Kernel_Name"].split("(")[0][:80]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        for k, v in dur.items():
            if "gemm" in k: print("   duration_ns", k, "n=%d avg=%.0f min=%d" % (len(v), sum(v) / len(v), min(v)))
PY
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete
cat "$OUT/r5_pmc_sq_x1.txt" | head -150
