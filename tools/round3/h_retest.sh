#!/bin/bash
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3h; mkdir -p $O
cd $R
python tools/round3/g_debug_adam.py 2>&1 | grep -v "^      " | grep -v amdgpu | head -20
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_replay.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 1500 python -m pytest tests/test_gpu_baseline_shapes.py -x -q -s -k "first_layer or cfg1 or configs0 or cf2" > $O/pytest2.log 2>&1; tail -3 $O/pytest2.log; grep "cf=2" $O/pytest2.log | cut -c1-400
