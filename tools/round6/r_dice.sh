# R6r: Dice delta on the REAL network (depth 4, 64 filters, dim 128, 128^3 volume)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6r; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_baseline_shapes.py -q -x -s -k "real_network" > $O/pytest.log 2>&1; grep -E "passed|failed|REAL network|Error|assert" $O/pytest.log | tail -8
