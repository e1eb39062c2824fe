mkdir -p gpurun_out/r3t && cd $GRAFT_REPO_ROOT
for cap in 0; do for ov in 1 0; do echo "hash_grad_blocks=$cap overlap=$ov: $(N_RAYS=65536,262144 CED_HASH_GRAD_BLOCKS=$cap OVERLAP=$ov timeout -k 10 300 python tools/bench_train.py 2>&1 | grep train_step | cut -c1-45 | tr '\n' ' ')"; done; done | tee -a gpurun_out/r3t/bench_train_caps.txt
