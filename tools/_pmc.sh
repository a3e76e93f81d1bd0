cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/kbench.py igemm3_deep 10
python $R/tools/kbench.py igemm3_deep2 10
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_ig1 -o p -- python $R/tools/kbench.py igemm3_deep 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_ig2 -o p -- python $R/tools/kbench.py igemm3_deep 3 > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
csv.field_size_limit(1<<30)
R=os.environ["GRAFT_REPO_ROOT"]
for d in ("pmc_ig1","pmc_ig2"):
    for f in glob.glob(f"{R}/gpurun_out/{d}/*counter_collection.csv"):
        acc={}
        for r in csv.DictReader(open(f)):
            if "conv_igemm" not in r["Kernel_Name"]: continue
            a=acc.setdefault(r["Counter_Name"],[0.0,0]); a[0]+=float(r["Counter_Value"]); a[1]+=1
        for k,v in sorted(acc.items()): print(d,k,v[0]/max(v[1],1))
PY
