R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { echo "== $*"; env "$@" timeout 600 python tools/round6/af_soak.py bf16x3 graphed=1,overlap=0 2>&1 | grep -v amdgpu.ids | grep -E "DIFF|SOAK" | tail -3; }
run MPU_FUSED_ADAM=0
run MPU_WGRAD_GROUP=0
run MPU_WGRAD_BATCHED_REDUCE=0
run MPU_WGRAD_TAPS=0
run MPU_FUSED_BN_STATS=0
run MPU_CONV_HALO=0
