#!/usr/bin/env python
"""GPU idle time inside the train step, from a rocprofv3 --kernel-trace --output-format csv run of `bench.py --profile-run`:
   python tools/gap_probe.py <*_kernel_trace.csv> [steps_in_run] [out.txt]
Kernels are sorted by start time; a GAP is the time between the end of the latest-ending kernel so far and the start of the next one
(> 0: the GPU had nothing to run -- the host had not enqueued the next launch yet, or a launch gap).  Reports busy / idle per step, the
idle time attributed to the kernel that FOLLOWS each gap, and the histogram of gap sizes."""
import csv
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def main():
    src = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
    rows = []
    for r in csv.DictReader(open(src)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"])))
    rows.sort()
    # steady-state steps only: from the SECOND adam_step_kernel (model set-up, the first warm-up step and its one-off allocations are out) to the last
    marks = [i for i, r in enumerate(rows) if "adam_step" in r[2]]
    if len(marks) >= 3:
        rows = rows[marks[1] + 1:marks[-1] + 1]
    else:
        rows = rows[(marks[0] if marks else 0) + 1:]
    busy = 0
    end = rows[0][0]
    idle_by = defaultdict(lambda: [0, 0.0])
    hist = defaultdict(lambda: [0, 0.0])
    big = []
    for s, e, name in rows:
        if s > end and s - end >= 20_000_000:      # >= 20 ms: a phase boundary of the driving script (warm-up -> timed region), not a gap of the step
            end = s
        if s > end:
            g = (s - end) / 1e3
            idle_by[name][0] += 1; idle_by[name][1] += g
            b = "<2us" if g < 2 else "2-5us" if g < 5 else "5-10us" if g < 10 else "10-30us" if g < 30 else "30-100us" if g < 100 else ">=100us"
            hist[b][0] += 1; hist[b][1] += g
            if g >= 100:
                big.append((g, name))
            busy += (e - s)
            end = e
        else:
            busy += max(0, e - max(s, end))
            end = max(end, e)
    span = (rows[-1][1] - rows[0][0]) / 1e6
    nsteps = sum(1 for r in rows if "adam_step" in r[2])
    steps = nsteps or steps
    tot_idle = sum(v[1] for v in idle_by.values()) / 1e3
    print(f"{len(rows)} kernels over {span:.1f} ms = {nsteps} steps: busy {busy / 1e6 / steps:.2f} ms/step, idle {tot_idle / steps:.2f} ms/step "
          f"({len(rows) / steps:.0f} launches/step)", file=out)
    print("gap size histogram (count/step, ms/step):", file=out)
    for b in ("<2us", "2-5us", "5-10us", "10-30us", "30-100us", ">=100us"):
        print(f"   {b:9s} {hist[b][0] / steps:8.1f} {hist[b][1] / 1e3 / steps:8.3f}", file=out)
    print("idle time by the kernel that follows the gap (ms/step, gaps/step):", file=out)
    for name, (n, us) in sorted(idle_by.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"   {us / 1e3 / steps:7.3f} {n / steps:6.1f}  {name[:110]}", file=out)
    print("gaps >= 100 us:", [(round(g), n[:50]) for g, n in sorted(big, reverse=True)[:20]], file=out)


if __name__ == "__main__":
    main()
