R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6ag; mkdir -p $O; cd $R
timeout 900 python tools/round6/ag_lossdbg.py bf16x3 2>&1 | grep -v amdgpu.ids | tee $O/lossdbg.txt | tail -14
