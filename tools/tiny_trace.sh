#!/bin/bash
# per-dispatch durations of conv_tiny in the batch-1 inference forward (grid = pixel tiles x cout tiles x K splits)
export TMPDIR=/tmp; R=$PWD; S=${1:-512}; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tt -o p -- python $R/tools/eval_fwd_probe.py $S 3 > /dev/null 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("/tmp/tt/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "conv_tiny_kernel" in r["Kernel_Name"] or "conv_halo_kernel<3" in r["Kernel_Name"] or "conv_gather" in r["Kernel_Name"]]
rows=rows[-len(rows)//6:]   # the last forward
for r in rows:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    print(f'{r["Kernel_Name"][:28]:28s} grid {int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]):5d} x {r["Grid_Size_Y"]:>3s} x {r["Grid_Size_Z"]:>3s}  {d:7.1f} us')
PY
