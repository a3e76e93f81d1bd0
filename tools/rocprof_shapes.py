#!/usr/bin/env python
"""Per (kernel, grid) launch-shape table from a rocprofv3 --kernel-trace --output-format csv run:
   python tools/rocprof_shapes.py <*_kernel_trace.csv> <out.csv> [steps_in_run] [name filter]"""
import csv
import re
import sys
from collections import defaultdict


def main():
    src, out = sys.argv[1], sys.argv[2]
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    filt = sys.argv[4] if len(sys.argv) > 4 else ""
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(src)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"])
        if filt and filt not in name:
            continue
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        grid = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"]),
                int(r["Grid_Size_Z"]) // int(r["Workgroup_Size_Z"]))
        k = (name, grid, wg)
        agg[k][0] += 1
        agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Blocks", "Threads", "CallsPerStep", "AvgUs", "MsPerStep"])
        for (name, grid, wg), (n, us) in rows:
            w.writerow([name, "x".join(map(str, grid)), wg, f"{n / steps:.1f}", f"{us / n:.1f}", f"{us / 1e3 / steps:.3f}"])
    print(f"{len(rows)} launch shapes")


if __name__ == "__main__":
    main()
