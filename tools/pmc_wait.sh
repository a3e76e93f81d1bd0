#!/bin/bash
# Where do the wave cycles of a kernel go?  One rocprofv3 --pmc pass (kernel-trace only) over one train step with the SQ wait buckets of
# MI355X_MICROARCH.md: SQ_WAIT_ANY (wave parked: s_waitcnt / barrier), SQ_WAIT_INST_ANY (issue stall: MFMA RAW / busy pipe), SQ_WAIT_INST_LDS
# (LDS issue stall, a sub-bucket of WAIT_INST_ANY), SQ_ACTIVE_INST_ANY; the three add up to ~SQ_WAVE_CYCLES.
#   bash tools/pmc_wait.sh r06        (GPU box; ~1 min)  -> gpurun_out/<tag>_pmc_wait.json + a table
tag=${1:-r06}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $O/${tag}_pmc_w -o p -- python $R/bench.py --steps 1 --warmup 1 --profile-run > $O/${tag}_pmc_w.log 2>&1
cd $R
python tools/pmc_summary.py $O/${tag}_pmc_wait.json $(find $O/${tag}_pmc_w -name "*counter_collection.csv") > /dev/null 2>&1
rm -rf $O/${tag}_pmc_w
python - <<PY
import json
d=json.load(open("$O/${tag}_pmc_wait.json"))
rows=[]
for k,v in d.items():
    if k=="_build" or not v.get("SQ_WAVE_CYCLES"): continue
    wc=v["SQ_WAVE_CYCLES"]
    rows.append((wc*v["launches"],k,v["launches"],v.get("SQ_WAIT_ANY",0)/wc,v.get("SQ_WAIT_INST_ANY",0)/wc,v.get("SQ_WAIT_INST_LDS",0)/wc,v.get("SQ_ACTIVE_INST_ANY",0)/wc))
rows.sort(reverse=True)
print(f"{'kernel':56s} {'n':>4s} {'parked':>7s} {'issue':>7s} {'(lds)':>7s} {'active':>7s}   (fractions of SQ_WAVE_CYCLES)")
for _,k,n,a,b,c,e in rows[:32]:
    print(f"{k[:56]:56s} {n:4d} {a:7.3f} {b:7.3f} {c:7.3f} {e:7.3f}")
PY
