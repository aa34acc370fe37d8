R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python tools/round6/al_which.py bf16x3 2>&1 | grep -v amdgpu.ids | tail -10
