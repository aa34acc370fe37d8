# R6ay: the 600-step soak (graphed, overlapped pipeline against the serial eager loop, configs[1] network) on the build with the
# head_bn_* passes and the pool-backward recompute: bf16 and bf16x3
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6ay; mkdir -p $O; cd $R
timeout 900 python tools/round6/af_soak.py bf16 2>&1 | grep -v amdgpu.ids | tee $O/soak_bf16.txt | tail -12
timeout 900 python tools/round6/af_soak.py bf16x3 2>&1 | grep -v amdgpu.ids | tee $O/soak_bf16x3.txt | tail -12
