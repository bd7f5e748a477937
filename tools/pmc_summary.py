#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, one pass each) into profiles/<name>.json.

usage: pmc_summary.py <dir with *counter_collection.csv (searched recursively)> <out.json> "<command that was profiled>"
                      [gemm_mode [captions_per_gpu]]     (recorded so that bench.py only quotes matching traffic)

Per kernel: launches, average FETCH_SIZE / WRITE_SIZE (KB, as rocprofv3 reports them) and
traffic_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: gfx950's FETCH_SIZE counts a 128-byte request as
64 bytes (MI355X_MICROARCH.md, HBM section), WRITE_SIZE needs no correction.  The counters sit at the L2 <-> fabric
boundary, so Infinity-Cache hits are included."""
import csv, glob, json, os, re, sys
from collections import defaultdict


def short(name: str) -> str:
    m = re.match(r"_ZN6capdec(\d+)", name)       # rocprofv3 leaves some template kernels mangled: _ZN6capdec<len><name>I...
    if m:
        n = int(m.group(1))
        return name[m.end():m.end() + n]
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)          # drop the argument list
    return name.replace("capdec::", "")


def main():
    root, out, cmd = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))      # counter -> kernel -> [launches, sum]
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                c = row.get("Counter_Name")
                if c not in ("FETCH_SIZE", "WRITE_SIZE"):
                    continue
                a = acc[c][short(row["Kernel_Name"])]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    kernels = {}
    for k in sorted(set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"])):
        f, w = acc["FETCH_SIZE"].get(k, [0, 0.0]), acc["WRITE_SIZE"].get(k, [0, 0.0])
        fk = f[1] / f[0] if f[0] else 0.0
        wk = w[1] / w[0] if w[0] else 0.0
        kernels[k] = {"launches": max(f[0], w[0]), "fetch_KB_raw": round(fk, 1), "write_KB": round(wk, 1),
                      "traffic_bytes_per_launch": int((2 * fk + wk) * 1024)}
    rec = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes): " + cmd,
           "gemm_mode": sys.argv[4] if len(sys.argv) > 4 else "bf16x3",
           "captions_per_gpu": int(sys.argv[5]) if len(sys.argv) > 5 else 5000,
           "correction": "traffic = (2 x FETCH_SIZE + WRITE_SIZE) KB: gfx950 FETCH_SIZE counts 128-B requests as 64 B; "
                         "counters sit at the L2<->fabric boundary (Infinity-Cache hits included)",
           "kernels": kernels}
    for k, v in kernels.items():       # flat aliases bench.py looks up (kernel name without template arguments)
        rec.setdefault(re.sub(r"<.*$", "", k), v)
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print(f"{k:48s} launches {v['launches']:6d}  traffic/launch {v['traffic_bytes_per_launch']/1e6:10.1f} MB")


if __name__ == "__main__":
    main()
