R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6ah; mkdir -p $O; cd $R
for v in "graphed=1,overlap=0" "graphed=0,overlap=1"; do timeout 600 python tools/round6/af_soak.py bf16x3 $v 2>&1 | grep -v amdgpu.ids | grep -E "DIFF|SOAK" | tail -4; done
MPU_BN_ATOMIC=0 timeout 600 python tools/round6/af_soak.py bf16x3 2>&1 | grep -v amdgpu.ids | grep -E "DIFF|SOAK" | tail -4
timeout 600 python tools/round6/af_soak.py f32 2>&1 | grep -v amdgpu.ids | grep -E "DIFF|SOAK" | tail -4
