R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python tools/round6/aj_guard.py bf16x3 2>&1 | grep -v amdgpu.ids | tail -12
