R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6p; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_baseline_shapes.py tests/test_gpu_unet.py tests/test_gpu_replay.py -q -s -k "split_bf16" > $O/pytest.log 2>&1; grep -E "passed|failed|max .err|Error|assert|bf16x3 step|replay:" $O/pytest.log | tail -12
