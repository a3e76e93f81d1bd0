#!/usr/bin/env python
"""One steady-state train step of a rocprofv3 kernel trace, launch by launch in time order: start (us from the previous optimizer launch's end), duration, gap to
the previous kernel's end, workgroups, kernel name.  Shows which launches are dependent neighbours, which are latency-sized, where the host falls behind
(gaps; note that the profiler itself slows the host).   python tools/step_timeline.py gpurun_out/r06_prof/p_results.db [step_from_the_end=2]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = list(db.execute("select name,start,end,grid_x,grid_y,grid_z,workgroup_x,workgroup_y,workgroup_z from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "adam_step" in r[0]]
a, b = idx[-back - 1], idx[-back]
t0 = prev_end = rows[a][2]
for r in rows[a + 1:b + 1]:
    nm = re.sub(r"\(.*", "", r[0]).replace("void ", "")[:64]
    wgs = (r[3] // max(r[6], 1)) * (r[4] // max(r[7], 1)) * (r[5] // max(r[8], 1))
    print(f"{(r[1] - t0) / 1e3:9.1f} us  dur {(r[2] - r[1]) / 1e3:8.1f}  gap {(r[1] - prev_end) / 1e3:6.1f}  wgs {wgs:6d}  {nm}")
    prev_end = max(prev_end, r[2])
print(f"{b - a} launches, {(rows[b][2] - t0) / 1e6:.3f} ms")
