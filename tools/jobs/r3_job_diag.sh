mkdir -p gpurun_out/r3d && cd $GRAFT_REPO_ROOT
export CED_NERF_LIB=$GRAFT_REPO_ROOT/build/variants/libcednerf_hip.diag.so
for sc in dnerf; do timeout -k 10 200 python tools/march_diag.py $sc 2>&1 | tee -a gpurun_out/r3d/march_diag.txt; done
