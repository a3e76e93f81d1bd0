#!/bin/bash
# Regenerates the measurement artefacts of a round on a GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r01
# Writes gpurun_out/<tag>_*: bench JSON (with cpu_baseline), rocprofv3 kernel-trace summary of the same command,
# per-launch dump, and two separate PMC passes (FETCH_SIZE, WRITE_SIZE) as MI355X_MICROARCH.md prescribes.
tag=${1:-r03}
R=$PWD
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
python bench.py --steps 10 --warmup 3 --details $O/${tag}_bench_details.json > $O/${tag}_bench_n1.json 2> $O/${tag}_bench.err      # (re-taken at the end, once the PMC files of this build exist)
KG_BENCH_DUMP=$O/${tag}_bench_launches.txt python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-companion > /dev/null 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/${tag}_prof -o p -- python $R/bench.py --steps 10 --warmup 3 --profile-run > $O/${tag}_prof.log 2>&1
python $R/tools/rocprof_summary.py $(find $O/${tag}_prof -name "*.db" | head -1) $O/${tag}_bench_kernel_stats.csv 13 >> $O/${tag}_prof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${tag}_pmc_rd -o p -- python $R/bench.py --steps 1 --warmup 1 --profile-run > $O/${tag}_pmc_rd.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${tag}_pmc_wr -o p -- python $R/bench.py --steps 1 --warmup 1 --profile-run > $O/${tag}_pmc_wr.log 2>&1
python $R/tools/pmc_summary.py $O/${tag}_pmc_hbm.json $(find $O/${tag}_pmc_rd $O/${tag}_pmc_wr -name "*counter_collection.csv") >> $O/${tag}_prof.log 2>&1
rm -rf $O/${tag}_pmc_rd $O/${tag}_pmc_wr   # raw per-dispatch CSVs are large
tail -3 $O/${tag}_prof.log; cat $O/${tag}_bench_n1.json | cut -c1-600
cd /tmp
# MFMA-pipe utilisation / LDS conflicts: two more counter passes (SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/${tag}_pmc_a -o p -- python $R/bench.py --steps 1 --warmup 1 --profile-run > $O/${tag}_pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/${tag}_pmc_b -o p -- python $R/bench.py --steps 1 --warmup 1 --profile-run > $O/${tag}_pmc_b.log 2>&1
python $R/tools/pmc_summary.py $O/${tag}_pmc_mfma_lds.json $(find $O/${tag}_pmc_a $O/${tag}_pmc_b -name "*counter_collection.csv") > /dev/null 2>&1
rm -rf $O/${tag}_pmc_a $O/${tag}_pmc_b
rocprofv3 --kernel-trace --output-format csv -d $O/${tag}_shp -o p -- python $R/bench.py --steps 4 --warmup 1 --profile-run > $O/${tag}_shp.log 2>&1
python $R/tools/rocprof_shapes.py $(find $O/${tag}_shp -name "*kernel_trace.csv" | head -1) $O/${tag}_bench_launch_shapes.csv 5 >> $O/${tag}_prof.log 2>&1
rm -rf $O/${tag}_shp
# the bench line again, now that the counter files of THIS build exist: bench.py fills roofline.traffic / roofline.pmc from profiles/*_pmc_*.json
# only when their build hash matches the loaded libraries
cd $R
cp $O/${tag}_pmc_hbm.json $O/${tag}_pmc_mfma_lds.json $R/profiles/
python bench.py --steps 20 --warmup 5 --details $O/${tag}_bench_details.json > $O/${tag}_bench_n1.json 2> $O/${tag}_bench.err
cut -c1-400 $O/${tag}_bench_n1.json
