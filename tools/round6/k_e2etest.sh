R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6k; mkdir -p $O; cd $R
echo "--- in suite, overlap on"; timeout 900 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_pipeline.py -q -s 2>&1 | grep -E "bare step|passed|failed|producer"
echo "--- in suite, overlap off"; MPU_TAIL_OVERLAP=0 timeout 900 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_pipeline.py -q -s 2>&1 | grep -E "bare step|passed|failed|producer"
echo "--- pipeline file alone, overlap on"; timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -s 2>&1 | grep -E "bare step|passed|failed|producer"
