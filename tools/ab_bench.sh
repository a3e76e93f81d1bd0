#!/bin/bash
# Same-box A/B of two builds of libkgnet_hip.so (box-to-box variance of the train step is ~1.5 %, larger than most kernel changes):
#   cp kg_instance_segmentation_amd/libkgnet_hip.so kg_instance_segmentation_amd/libkgnet_hip_old.so   # baseline build
#   ... edit, rebuild ...;  gpurun -- 'bash tools/ab_bench.sh'
for i in 1 2 3; do
  for L in libkgnet_hip_old.so libkgnet_hip.so; do
    echo -n "$L "
    KG_LIB_PATH=$PWD/kg_instance_segmentation_amd/$L python bench.py --steps 10 --warmup 3 --no-companion --no-cpu-baseline "$@" 2>&1 | tail -1 |
      python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
