"""Per-(kernel, grid) table of a `rocprofv3 --kernel-trace --output-format csv` run.

    python tools/trace_summary.py <dir with *kernel_trace.csv> <out.txt> [--skip-first-ms X] [--title "..."]

Rows: kernel name (template arguments kept, argument list dropped), grid, workgroup, launches, average / minimum
duration, total and share of the traced window; header: launches, window span, busy time (sum of durations) and the
idle time between kernels.  Used for profiles/r4_625_kernels*.txt (what one decode step of the 625-caption shard is
made of)."""
import collections
import csv
import glob
import sys


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("out")
    ap.add_argument("--title", default="")
    ap.add_argument("--skip-first-ms", type=float, default=0.0)
    a = ap.parse_args()
    src, out, title, skip_ms = a.src, a.out, a.title, a.skip_first_ms
    rows = []
    for p in glob.glob(f"{src}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(p, newline="")):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            name = r["Kernel_Name"].split("(")[0].strip()
            grid = r.get("Grid_Size") or "x".join(r.get(k, "1") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
            wg = r.get("Workgroup_Size") or "x".join(r.get(k, "1") for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
            rows.append((s, e, name, grid, wg))
    if not rows:
        open(out, "w").write("no kernel_trace.csv rows found\n")
        return
    rows.sort()
    t0 = rows[0][0] + int(skip_ms * 1e6)
    rows = [r for r in rows if r[0] >= t0]
    span = (rows[-1][1] - rows[0][0]) / 1e6
    busy = sum(e - s for s, e, *_ in rows) / 1e6
    acc = collections.OrderedDict()
    for s, e, name, grid, wg in rows:
        k = (name, grid, wg)
        a = acc.setdefault(k, [0, 0, 1 << 62])
        a[0] += 1
        a[1] += e - s
        a[2] = min(a[2], e - s)
    with open(out, "w") as f:
        if title:
            f.write(title + "\n")
        f.write(f"kernels {len(rows)} span_ms {span:.2f} busy_ms {busy:.2f} idle_ms {span - busy:.2f}\n")
        for (name, grid, wg), (n, tot, mn) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            f.write(f"  {name[-92:]:92s} grid={grid:>9s} wg={wg:>5s} n={n:6d} avg_us {tot / n / 1e3:9.2f} min_us {mn / 1e3:9.2f} "
                    f"total_ms {tot / 1e6:9.2f} {100.0 * tot / 1e6 / busy:5.1f}%\n")


if __name__ == "__main__":
    main()
