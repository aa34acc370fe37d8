R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for md in sync async async_eager; do timeout 600 python tools/round6/ai_graphdbg.py bf16x3 $md 2>&1 | grep -v amdgpu.ids | tail -3; done
timeout 600 python tools/round6/ai_graphdbg.py bf16 async 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python tools/round6/ai_graphdbg.py f32 async 2>&1 | grep -v amdgpu.ids | tail -2
