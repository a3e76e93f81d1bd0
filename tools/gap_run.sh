#!/bin/bash
# GPU idle time inside a steady-state train step: kernel trace of a short bench run -> tools/gap_probe.py
#   bash tools/gap_run.sh r06        (GPU box)  -> gpurun_out/<tag>_gap_probe.txt
tag=${1:-r06}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/${tag}_gap -o p -- python $R/bench.py --steps 5 --warmup 3 --profile-run > $O/${tag}_gap.log 2>&1
cd $R
python tools/gap_probe.py $(find $O/${tag}_gap -name "*kernel_trace.csv" | head -1) 3 $O/${tag}_gap_probe.txt
rm -rf $O/${tag}_gap
head -16 $O/${tag}_gap_probe.txt
