R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6ap; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_cli.py -q -x -s 2>&1 | grep -E "passed|failed|probe latency|bare step|Error" | tail -6
