#!/bin/bash
# per-dispatch durations of the post-processing kernels in bench.py --mode eval (run on a GPU box through gpurun)
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/evshp -o p -- python $R/bench.py --mode eval --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/evshp.log 2>&1
cd $R
python - <<PY
import csv,glob,re
f=glob.glob("gpurun_out/evshp/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
keep=("group_kernel","nms_kernel","scan_kernel","hough","gauss","peaks","kp_rank","boxes")
t0=None
sel=[r for r in rows if any(k in r["Kernel_Name"] for k in keep)]
for r in sel[-66:]:
    n=r["Kernel_Name"]
    if any(k in n for k in keep):
        s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
        if t0 is None: t0=s
        print(f"{(s-t0)/1e3:10.1f} {(e-s)/1e3:9.1f}  q{r.get('Queue_Id','')} {re.sub(r'[(<].*','',n)}")
PY
rm -rf gpurun_out/evshp
