#!/bin/bash
# MFMA-busy / LDS-activity counters per kernel for one train step (two rocprofv3 --pmc passes, kernel-trace only), summarised into
# gpurun_out/<tag>_pmc_mfma_lds.json:   bash tools/pmc_quick.sh r04x        (GPU box; ~1 min)
tag=${1:-r04x}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/${tag}_pmc_a -o p -- python $R/bench.py --steps 1 --warmup 1 --profile-run > $O/${tag}_pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS --output-format csv -d $O/${tag}_pmc_b -o p -- python $R/bench.py --steps 1 --warmup 1 --profile-run > $O/${tag}_pmc_b.log 2>&1
cd $R
python tools/pmc_summary.py $O/${tag}_pmc_mfma_lds.json $(find $O/${tag}_pmc_a $O/${tag}_pmc_b -name "*counter_collection.csv") > /dev/null 2>&1
rm -rf $O/${tag}_pmc_a $O/${tag}_pmc_b
python - <<PY
import json
d=json.load(open("$O/${tag}_pmc_mfma_lds.json"))
print(f"{'kernel':52s} {'n':>4s} {'mfma_busy':>9s} {'lds_active':>10s} {'lds_conf':>8s}")
for k,v in d.items():
    if k=="_build" or not v.get("GRBM_GUI_ACTIVE"): continue
    act=v["GRBM_GUI_ACTIVE"]/8.0
    if act*v["launches"] < 2e5: continue
    mf=(v.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/1024.0)/act
    lds=(v.get("SQ_LDS_IDX_ACTIVE",0)/256.0)/act if v.get("SQ_LDS_IDX_ACTIVE") else 0
    cf=(v.get("SQ_LDS_BANK_CONFLICT",0)/256.0)/act if v.get("SQ_LDS_BANK_CONFLICT") else 0
    print(f"{k[:52]:52s} {v['launches']:4d} {mf:9.3f} {lds:10.3f} {cf:8.3f}")
PY
