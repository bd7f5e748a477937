#!/bin/bash
# L2 <-> fabric traffic of the RN50x4 tower's kernels (128 images): FETCH_SIZE and WRITE_SIZE in separate passes
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cat > /tmp/rn_once.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from capdec_amd import synth, clip as cclip
sd = synth.hot_clip_resnet_state_dict(44, synth.CLIP_RN50X4)
model, _ = cclip.load(sd, device=0)
imgs = synth.synthetic_images(128, seed=1, size=288).cuda()
model.encode_image(imgs); torch.cuda.synchronize()
PY
cd "$R"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d "$OUT/r2_rnpmc_$c" -- python /tmp/rn_once.py > "$OUT/r2_rnpmc_$c.log" 2>&1
done
python - <<'PY' > "$OUT/r2_rn50x4_traffic.txt"
import csv, glob, collections, re
acc = collections.defaultdict(lambda: {"FETCH_SIZE": [0, 0.0], "WRITE_SIZE": [0, 0.0]})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for p in glob.glob(f"gpurun_out/r2_rnpmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p, newline="")):
            if r["Counter_Name"] != c: continue
            k = r["Kernel_Name"].split("(")[0][-44:]
            g = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
            a = acc[(k, g)][c]; a[0] += 1; a[1] += float(r["Counter_Value"])
print("traffic at the L2 <-> fabric boundary, RN50x4 tower, 128 images (bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024; gfx950 FETCH correction)")
tot_f = tot_w = 0.0
rows = []
for (k, g), v in acc.items():
    n = max(v["FETCH_SIZE"][0], v["WRITE_SIZE"][0])
    f = 2 * v["FETCH_SIZE"][1] * 1024; w = v["WRITE_SIZE"][1] * 1024
    rows.append((f + w, k, g, n, f, w))
for t, k, g, n, f, w in sorted(rows, reverse=True)[:30]:
    print("  %-44s grid=%-9s n=%3d  fetch %8.1f MB/launch  write %8.1f MB/launch" % (k, g, n, f / n / 1e6, w / n / 1e6))
conv = [r for r in rows if "ConvGeo" in r[1] or "conv3x3" in r[1]]
print("implicit-GEMM convolutions: total fetch %.2f GB, total write %.2f GB over %d launches" % (sum(r[4] for r in conv) / 1e9, sum(r[5] for r in conv) / 1e9, sum(r[3] for r in conv)))
print("all kernels: fetch %.2f GB write %.2f GB" % (sum(r[4] for r in rows) / 1e9, sum(r[5] for r in rows) / 1e9))
PY
cat "$OUT/r2_rn50x4_traffic.txt"
find "$OUT" -name "*counter_collection.csv" -delete; rm -rf "$OUT"/r2_rnpmc_FETCH_SIZE "$OUT"/r2_rnpmc_WRITE_SIZE
