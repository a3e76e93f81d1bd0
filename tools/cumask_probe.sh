#!/bin/bash
# Which kernels of the train step are latency-bound?  The same step on HALF the CUs (HSA_CU_MASK): a kernel whose time does not move has
# no use for the other half -- the head room for running independent work next to it.   gpurun -- 'bash tools/cumask_probe.sh'
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
MASK=${MASK:-0:0-127}
for m in "" "$MASK"; do
  tag=full; [ -n "$m" ] && tag=half
  ( [ -n "$m" ] && export HSA_CU_MASK=$m
    timeout 300 python bench.py --steps 6 --warmup 3 --no-companion --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], 'ms/step')"
    cd /tmp
    timeout 600 rocprofv3 --kernel-trace --stats -d $O/cum_$tag -o p -- python $R/bench.py --steps 6 --warmup 3 --profile-run > $O/cum_$tag.log 2>&1
    python $R/tools/rocprof_summary.py $(find $O/cum_$tag -name "*.db" | head -1) $O/cumask_${tag}_kernel_stats.csv 9
    rm -rf $O/cum_$tag )
done
python - <<PY
import csv
f={r['Name']:r for r in csv.DictReader(open('$O/cumask_full_kernel_stats.csv'))}
h={r['Name']:r for r in csv.DictReader(open('$O/cumask_half_kernel_stats.csv'))}
print(f"{'kernel':62s} {'full ms/step':>12s} {'half ms/step':>12s} ratio")
tf=th=0
for k,r in f.items():
    if k in h:
        a,b=float(r['MsPerStep']),float(h[k]['MsPerStep']); tf+=a; th+=b
        if a>0.15: print(f"{k[:62]:62s} {a:12.3f} {b:12.3f} {b/max(a,1e-9):5.2f}")
print('sum', tf, th)
PY
