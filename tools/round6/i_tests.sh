# R6i: new GPU tests (per-view evaluation, e2e guard, CLI) 
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6i; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_pipeline.py tests/test_gpu_cli.py tests/test_gpu_fusion_train.py -q -x -s -k "per_view or pipeline or side_stream or mp_train_then or fusion" > $O/pytest.log 2>&1; grep -E "passed|failed|bare step|producer stream|Error" $O/pytest.log | tail -12
