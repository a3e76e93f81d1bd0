#!/bin/bash
# Same-box A/B of environment switches / library variants on the train step: each line = one bench run (ms/step, dominant-kernel fraction, its launch time)
#   bash tools/ab_env.sh "KG_HALO7_W4=0" "KG_HALO7_W4=2" ...        (an argument is an env assignment list; repeated 3 times round-robin)
for i in 1 2 3; do
  for V in "$@"; do
    echo -n "$V  "
    env $V timeout 300 python bench.py --steps 20 --warmup 5 --no-companion --no-cpu-baseline 2>/dev/null | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],2), round(d['ms_per_step'],3), round(r['frac'],4), round(r['avg_launch_ms'],4), r['kernel'][:24])"
  done
done
