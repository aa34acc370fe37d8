R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6k; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_pipeline.py -q -x -s 2>&1 | grep -E "bare step|passed|failed|producer"
