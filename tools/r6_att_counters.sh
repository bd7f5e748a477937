#!/bin/bash
# SQ counters of the matrix-core prefill attention inside the CLIP text tower (BASELINE configs[3], fp16 towers): matrix-pipe
# busy cycles, wave wait states, VALU / MFMA / LDS instruction counts, LDS bank conflicts -- two --pmc passes (8 SQ slots each)
#   gpurun --timeout 900 -- 'bash tools/r6_att_counters.sh'   ->  gpurun_out/r6_pmc_sq_attn_prefill.txt
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$R"
CMD="python bench.py --workload text_embed --captions 8000 --gemm-mode f16 --steps 1 --warmup 0 --cpu-seconds 0 --cpu-captions 0 --no-checks --no-smi"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/r6_sq_att_a" -- $CMD > "$OUT/r6_sq_att_a.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU \
    --output-format csv -d "$OUT/r6_sq_att_b" -- $CMD > "$OUT/r6_sq_att_b.log" 2>&1
python - "$OUT" <<'PY' > "$OUT/r6_pmc_sq_attn_prefill.txt" 2>&1
import csv, glob, collections, sys, os
out = sys.argv[1]
print("SQ counters per launch, bench.py --workload text_embed --captions 8000 --gemm-mode f16 --steps 1 --warmup 0 (chunks of 4000 captions:")
print("a launch of attn_prefill_mfma_kernel<true, 3> = 4000 captions x 8 heads x 77 tokens = 32 000 blocks of 3 wavefronts); gemm_x1 next to it")
for d in ("r6_sq_att_a", "r6_sq_att_b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for p in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(p, newline="")):
            k = row["Kernel_Name"].split("(")[0]
            if "attn_prefill" not in k and "gemm_x1_kernel" not in k: continue
            key = k[:70] + " grid=" + row.get("Grid_Size", "?")
            a = acc[key][row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
    for k, cs in sorted(acc.items()):
        print(d, k)
        for c, (n, s) in sorted(cs.items()):
            print(f"   {c:32s} n={n:4d} avg={s/n:18.1f}")
        g = lambda c: cs[c][1] / cs[c][0] if c in cs else None
        if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_BUSY_CYCLES"):
            print("   mfma_busy / sq_busy = %.3f   wait_any / wave_cycles = %.3f   wait_inst_any / wave_cycles = %.3f   active / wave_cycles = %.3f"
                  % (g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES"), g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
                     g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")))
        if g("SQ_INSTS_VALU") and g("SQ_INSTS_MFMA"):
            print("   VALU / MFMA instructions = %.1f   LDS bank-conflict cycles / LDS active cycles = %.3f"
                  % (g("SQ_INSTS_VALU") / g("SQ_INSTS_MFMA"), (g("SQ_LDS_BANK_CONFLICT") or 0) / max(g("SQ_LDS_IDX_ACTIVE") or 1, 1)))
    for p in glob.glob(f"{out}/{d}/**/*kernel_trace.csv", recursive=True):
        dur = collections.defaultdict(list)
        for row in csv.DictReader(open(p, newline="")):
            dur[row["Kernel_Name"].split("(")[0][:70]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        for k, v in dur.items():
            if "attn_prefill" in k: print("   duration_ns (under counters)", k, "n=%d avg=%.0f min=%d" % (len(v), sum(v) / len(v), min(v)))
PY
rm -rf "$OUT/r6_sq_att_a" "$OUT/r6_sq_att_b"
cat "$OUT/r6_pmc_sq_attn_prefill.txt" | head -70
