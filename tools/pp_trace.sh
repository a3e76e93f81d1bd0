#!/bin/bash
# per-dispatch timeline of the post-processing kernels of ONE 512 x 512 image (kpp.detect on the eval bench's head maps): start offset,
# duration, queue, kernel -- run on a GPU box through gpurun:   bash tools/pp_trace.sh [size]
S=${1:-512}
export TMPDIR=/tmp; R=$PWD; cd /tmp
cat > /tmp/pp_one.py <<PY
import sys; sys.path.insert(0, "$R")
import torch, bench
from kg_instance_segmentation_amd import postprocessing as kpp
dec_np, _ = bench.eval_inputs($S, 300, 5)
dec = [[torch.from_numpy(a).cuda() for a in d] for d in dec_np]
for _ in range(4):
    kpp.detect(dec)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pptr -o p -- python /tmp/pp_one.py > $R/gpurun_out/pptr.log 2>&1
cd $R
python - <<PY
import csv,glob,re
f=glob.glob("gpurun_out/pptr/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
# the last detect() call = everything after the third-from-last nms_kernel ... simply: rows after the 3rd nms_kernel
idx=[i for i,r in enumerate(rows) if "nms_kernel" in r["Kernel_Name"]]
sel=rows[idx[-2]+1: idx[-1]+1]
t0=int(sel[0]["Start_Timestamp"])
tot={}
for r in sel:
    n=re.sub(r"[(<].*","",r["Kernel_Name"]).replace("void ","")
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    g=int(r["Grid_Size_X"])*int(r["Grid_Size_Y"])
    print(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f}  q{r.get('Queue_Id','')}  grid {g:9d}  {n}")
    tot[n]=tot.get(n,0)+(e-s)/1e3
print("per kernel (us, all 4 scales):", {k: round(v,1) for k,v in sorted(tot.items(), key=lambda kv:-kv[1])})
print("wall (us):", (int(sel[-1]["End_Timestamp"])-t0)/1e3)
PY
rm -rf gpurun_out/pptr
