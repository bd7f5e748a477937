#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd "$R"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/r2_kt625" -- python bench.py --cpu-seconds 0 --steps 1 --warmup 1 --captions 625 --profile-every 1000000 > "$OUT/r2_kt625.json" 2> "$OUT/r2_kt625.err"
python - <<'PY'
import csv, glob, collections
p = glob.glob("gpurun_out/r2_kt625/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(p, newline="")):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]))
rows.sort()
# last 40% of the trace = inside the timed step; take a window of 3000 kernels near the end
w = rows[-4000:-500]
busy = sum(e - s for s, e, _ in w)
span = w[-1][1] - w[0][0]
gaps = [w[i + 1][0] - w[i][1] for i in range(len(w) - 1)]
print("kernels", len(w), "span_ms %.2f busy_ms %.2f idle_ms %.2f" % (span / 1e6, busy / 1e6, (span - busy) / 1e6))
gs = sorted(gaps)
print("gap ns: median %d p90 %d p99 %d max %d, negative(overlap) %d" % (gs[len(gs) // 2], gs[int(len(gs) * .9)], gs[int(len(gs) * .99)], gs[-1], sum(g < 0 for g in gaps)))
per = collections.defaultdict(lambda: [0, 0])
for s, e, k in w:
    per[k][0] += 1; per[k][1] += e - s
for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:12]:
    print("  %-46s n=%5d avg_us %8.2f total_ms %7.2f" % (k, n, t / n / 1e3, t / 1e6))
# gap following each kernel type
gp = collections.defaultdict(list)
for i in range(len(w) - 1):
    gp[w[i][2]].append(w[i + 1][0] - w[i][1])
for k, v in sorted(gp.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print("  gap after %-40s n=%5d avg_us %7.2f" % (k, len(v), sum(v) / len(v) / 1e3))
PY
rm -rf "$OUT/r2_kt625"
