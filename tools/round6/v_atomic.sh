R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6v; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests/test_gpu_unet.py -q -s -k "accumulators or folded" > $O/pytest.log 2>&1; grep -E "passed|failed|accumulator form|Error|assert " $O/pytest.log | tail -8
