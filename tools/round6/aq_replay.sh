R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6aq; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests/test_gpu_replay.py -q -x -s -k "split_bf16" > $O/pytest.log 2>&1; grep -E "passed|failed|Error|assert|worst|forward|gradient" $O/pytest.log | tail -12
