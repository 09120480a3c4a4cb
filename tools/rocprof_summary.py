#!/usr/bin/env python3
"""Summarise a rocprofv3 `--kernel-trace --stats` results .db into a small CSV (one row per kernel).

    python tools/rocprof_summary.py gpurun_out/prof_xx/yy_results.db profiles/r01_xx_kernel_stats.csv [--ours]
"""
import csv
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    ours = "--ours" in sys.argv
    c = sqlite3.connect(db)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "pct_of_all_gpu_kernel_time"])
        for name, calls, total, avg, pct in rows:
            if ours and "anonymous namespace)::k_" not in name:
                continue
            short = name if len(name) < 160 else name[:157] + "..."
            w.writerow([short, calls, round(total, 3), round(avg, 3), round(pct, 3)])
    print(open(out).read())


if __name__ == "__main__":
    main()
