python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k wgrad 2>&1 | tail -1
for L in libkgnet_hip_old.so libkgnet_hip.so; do for c in wg7_c0 wg7_c3 wg3_c0; do echo -n "$L "; KG_LIB_PATH=$PWD/kg_instance_segmentation_amd/$L python tools/kbench.py $c 10; done; done
bash tools/ab_bench.sh
