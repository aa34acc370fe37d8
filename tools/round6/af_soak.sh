R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6af; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_unet.py -q -k "ready_events" 2>&1 | tail -2
timeout 900 python tools/round6/af_soak.py bf16 2>&1 | grep -v amdgpu.ids | tee $O/soak_bf16.txt | tail -12
timeout 900 python tools/round6/af_soak.py bf16x3 2>&1 | grep -v amdgpu.ids | tee $O/soak_bf16x3.txt | tail -12
