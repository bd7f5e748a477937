#!/bin/bash
# per-(kernel, grid) time of the RN50x4 tower (128 images)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cat > /tmp/rn_once.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from capdec_amd import synth, clip as cclip
sd = synth.hot_clip_resnet_state_dict(44, synth.CLIP_RN50X4)
model, _ = cclip.load(sd, device=0, precision=os.environ.get("PREC", "fp32"))
imgs = synth.synthetic_images(128, seed=1, size=288).cuda()
model.encode_image(imgs); torch.cuda.synchronize()
model.encode_image(imgs); torch.cuda.synchronize()
PY
cd "$R"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/r2_rnkt" -- python /tmp/rn_once.py > /dev/null 2> "$OUT/r2_rnkt.err"
python - <<'PY' > "$OUT/r2_rn50_kernels.txt"
import csv, glob, collections
p = glob.glob("gpurun_out/r2_rnkt/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(p, newline="")):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-56:], r.get("Grid_Size_X") or r.get("Grid_Size") or "?"))
rows.sort()
# second encode_image call = last half of the non-init kernels: take kernels after the last attnpool_attend of call 1
idx = [i for i, r in enumerate(rows) if "attnpool_attend" in r[2]]
w = rows[idx[-2] + 3: idx[-1] + 3] if len(idx) >= 2 else rows
busy = sum(e - s for s, e, _, _ in w); span = w[-1][1] - w[0][0]
print("kernels", len(w), "span_ms %.2f busy_ms %.2f" % (span / 1e6, busy / 1e6))
per = collections.defaultdict(lambda: [0, 0])
for s, e, k, g in w:
    per[(k, g)][0] += 1; per[(k, g)][1] += e - s
for (k, g), (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:40]:
    print("  %-56s grid=%-9s n=%3d avg_us %8.1f total_ms %6.2f %4.1f%%" % (k, g, n, t / n / 1e3, t / 1e6, 100.0 * t / busy))
PY
cat "$OUT/r2_rn50_kernels.txt"
rm -rf "$OUT/r2_rnkt"
