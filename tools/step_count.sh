#!/bin/bash
# kernel trace of 5 bench steps -> gpurun_out/<tag>_step_count.txt (tools/step_count.py):   bash tools/step_count.sh r04
tag=${1:-r04}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/${tag}_sc -o p -- python $R/bench.py --steps 4 --warmup 1 --profile-run > $O/${tag}_sc.log 2>&1
python $R/tools/step_count.py $(find $O/${tag}_sc -name "*kernel_trace.csv" | head -1) $O/${tag}_step_count.txt
python $R/tools/gap_probe.py $(find $O/${tag}_sc -name "*kernel_trace.csv" | head -1) 1 $O/${tag}_gap_probe.txt
rm -rf $O/${tag}_sc
head -50 $O/${tag}_step_count.txt
