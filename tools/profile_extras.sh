#!/bin/bash
# The side measurements quoted in DESIGN.md / README.md (run after tools/profile_round.sh, same box):  bash tools/profile_extras.sh r02
tag=${1:-r03}
O=$PWD/gpurun_out
mkdir -p $O
python tools/pareto.py --out $O/${tag}_pareto.json --steps 8 ${PARETO_POLICIES:-fp32 fp32b2 halfmix half fp32bf fp32bf_full fp32bf_w1d1 trunk2 mixed bf16} > $O/${tag}_pareto.log 2>&1
python bench.py --steps 6 --warmup 2 --batch 16 --no-companion --no-cpu-baseline > $O/${tag}_bench_bs16.json 2>/dev/null
KG_BENCH_SYNC=0 python bench.py --steps 10 --warmup 3 --no-companion --no-cpu-baseline > $O/${tag}_bench_nosync.json 2>/dev/null
python bench.py --mode eval --steps 10 > $O/${tag}_eval.json 2>/dev/null
python bench.py --mode gt --steps 5 > $O/${tag}_gt.json 2>/dev/null
for m in mfma_peak wgrad_skel c3_parts; do
  [ -x tools/micro/$m ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value tools/micro/$m.hip -o tools/micro/$m > /dev/null 2>&1
  [ -x tools/micro/$m ] && timeout 120 tools/micro/$m > $O/${tag}_micro_$m.txt 2>&1
done
for f in bs16 nosync; do python -c "import json,sys; d=json.load(open('$O/${tag}_bench_$f.json')); print('$f', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')"; done
