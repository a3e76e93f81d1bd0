#!/bin/bash
# per-kernel table of the batch-1 inference forward (tools/eval_fwd_probe.py) -> gpurun_out/<tag>_evalfwd_kernel_stats.csv
tag=${1:-r03}; S=${2:-512}
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O; cd /tmp
python $R/tools/eval_fwd_probe.py $S 20 | tee $O/${tag}_evalfwd_$S.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${tag}_evalfwd_prof -o p -- python $R/tools/eval_fwd_probe.py $S 20 > $O/${tag}_evalfwd_prof.log 2>&1
db=$(find $O/${tag}_evalfwd_prof -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $db $O/${tag}_evalfwd_${S}_kernel_stats.csv 23
rm -rf $O/${tag}_evalfwd_prof
head -32 $O/${tag}_evalfwd_${S}_kernel_stats.csv | cut -d, -f1,6,7 | cut -c1-150
