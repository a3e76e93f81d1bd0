#!/bin/bash
# Same-box A/B of two builds (box-to-box variance of the train step is ~1.5 %, larger than most kernel changes).  The default policy runs on
# libkgnet_hip_f16.so (IEEE-half rows), the bf16 policies on libkgnet_hip.so:
#   cp kg_instance_segmentation_amd/libkgnet_hip_f16.so kg_instance_segmentation_amd/libkgnet_hip_f16_old.so   # baseline build
#   ... edit, rebuild ...;  gpurun -- 'bash tools/ab_bench.sh'            (bf16 policies: LIBVAR=KG_LIB_PATH BASE=libkgnet_hip bash tools/ab_bench.sh --precision mixed)
LIBVAR=${LIBVAR:-KG_LIB_F16_PATH}
BASE=${BASE:-libkgnet_hip_f16}
for i in 1 2 3; do
  for L in ${BASE}_old.so ${BASE}.so; do
    echo -n "$L "
    env $LIBVAR=$PWD/kg_instance_segmentation_amd/$L python bench.py --steps 10 --warmup 3 --no-companion --no-cpu-baseline "$@" 2>&1 | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms/step; dominant kernel', d['roofline']['achieved'], d['roofline']['unit'])"
  done
done
