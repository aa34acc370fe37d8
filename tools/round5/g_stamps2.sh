R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5g; mkdir -p $O
cd $R
MPU_STAMPS=1 python tools/round5/deep_layers.py stamps 2>&1 | grep -v "^$" | tee $O/stamps.txt
