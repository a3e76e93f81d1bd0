#!/usr/bin/env python
"""Launches and kernel time of ONE steady-state train step, per kernel, from a rocprofv3 --kernel-trace --output-format csv run of
`bench.py --profile-run`: the steps between consecutive adam_step_kernel launches, first one (model set-up) skipped.
   python tools/step_count.py <*_kernel_trace.csv> [out.txt]"""
import csv
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")))
rows.sort()
marks = [i for i, r in enumerate(rows) if "adam_step" in r[2]]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
if len(marks) < 3:
    sys.exit("need at least three optimizer steps in the trace")
steps = len(marks) - 2
span = rows[marks[1] + 1:marks[-1] + 1]
cnt, tim = defaultdict(int), defaultdict(float)
for s, e, n in span:
    cnt[n] += 1; tim[n] += (e - s) * 1e-6
n_l = sum(cnt.values()) / steps
small = sum(1 for s, e, n in span if e - s < 30000) / steps
small_ms = sum((e - s) for s, e, n in span if e - s < 30000) * 1e-6 / steps
print(f"{steps} steady-state steps: {n_l:.1f} launches/step, {sum(tim.values()) / steps:.2f} ms of kernels/step; "
      f"launches under 30 us: {small:.1f}/step, {small_ms:.2f} ms/step", file=out)
for n in sorted(cnt, key=lambda k: -cnt[k]):
    print(f"{cnt[n] / steps:7.1f}  {tim[n] / steps:8.3f} ms  {n[:120]}", file=out)
