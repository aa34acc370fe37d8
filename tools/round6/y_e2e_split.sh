R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6y; mkdir -p $O; cd $R
timeout 600 python tools/round6/y_e2e_split.py 2>&1 | grep -v amdgpu.ids | tee $O/split.txt
