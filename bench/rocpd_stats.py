#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / mean / share -> CSV.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o run -- python3 bench.py ...
    python bench/rocpd_stats.py gpurun_out/prof/run_results.db profiles/NAME_kernel_stats.csv
"""
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main(db_path: str, out_csv: str | None = None, top: int = 40) -> None:
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0])
        a[0] += 1
        a[1] += e - s
    total = sum(v[1] for v in agg.values()) or 1
    table = sorted(((k, v[0], v[1] / 1e6, v[1] / v[0] / 1e3, 100.0 * v[1] / total) for k, v in agg.items()),
                   key=lambda r: -r[2])
    if out_csv:
        with open(out_csv, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_ms", "mean_us", "pct"])
            for r in table:
                w.writerow([r[0], r[1], f"{r[2]:.3f}", f"{r[3]:.2f}", f"{r[4]:.2f}"])
    print(f"total GPU kernel time {total / 1e6:.2f} ms over {len(rows)} dispatches")
    for r in table[:top]:
        print(f"{r[4]:6.2f}%  {r[2]:10.3f} ms  {r[1]:7d} x {r[3]:10.2f} us  {r[0]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
