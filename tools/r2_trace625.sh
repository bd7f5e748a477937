#!/bin/bash
# per-(kernel, grid) durations inside the 625-caption decode loop
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd "$R"
CAP=${1:-625}
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/r2_ktb" -- python bench.py --cpu-seconds 0 --steps 1 --warmup 1 --captions $CAP --profile-every 1000000 > "$OUT/r2_ktb_$CAP.json" 2> "$OUT/r2_ktb.err"
python - $CAP <<'PY' > "$OUT/r2_ktb_$1.txt"
import csv, glob, collections, sys
p = glob.glob("gpurun_out/r2_ktb/**/*kernel_trace.csv", recursive=True)[0]
rows = []
rd = csv.DictReader(open(p, newline=""))
for r in rd:
    g = (r.get("Grid_Size_X") or r.get("Grid_Size") or "?", r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "?")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:], g))
rows.sort()
w = rows[-4500:-300]
span = w[-1][1] - w[0][0]
busy = sum(e - s for s, e, _, _ in w)
print("captions", sys.argv[1], "kernels", len(w), "span_ms %.2f busy_ms %.2f" % (span / 1e6, busy / 1e6))
per = collections.defaultdict(lambda: [0, 0, 10**12])
for s, e, k, g in w:
    a = per[(k, g)]
    a[0] += 1; a[1] += e - s; a[2] = min(a[2], e - s)
for (k, g), (n, t, mn) in sorted(per.items(), key=lambda kv: -kv[1][1])[:30]:
    print("  %-60s grid=%-8s wg=%-4s n=%5d avg_us %8.2f min_us %8.2f total_ms %7.2f  %4.1f%%" % (k, g[0], g[1], n, t / n / 1e3, mn / 1e3, t / 1e6, 100.0 * t / busy))
PY
cat "$OUT/r2_ktb_$CAP.txt"
rm -rf "$OUT/r2_ktb"
