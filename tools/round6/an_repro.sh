R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6an; mkdir -p $O; cd $R
timeout 600 python tools/round6/an_repro.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/predict_repro.txt

