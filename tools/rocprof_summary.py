#!/usr/bin/env python
"""Turns a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into the per-kernel stats table committed under
profiles/:   python tools/rocprof_summary.py <results.db> <out.csv> [steps_in_run]"""
import csv
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "CallsPerStep", "MsPerStep"])
        for name, calls, total, avg, pct in rows:
            w.writerow([name, calls, int(total * 1e3), int(avg * 1e3), f"{pct:.3f}", f"{calls / steps:.1f}", f"{total / 1e3 / steps:.3f}"])
    tot = sum(r[2] for r in rows)
    print(f"{len(rows)} kernels, {tot / 1e3 / steps:.2f} ms of kernel time per step")


if __name__ == "__main__":
    main()
