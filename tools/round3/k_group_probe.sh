#!/bin/bash
# round 3 call K: does a multi-wave grid amortise the per-launch overhead of wgrad_taps? (B=16: 256 WGs; B=64: 1024 WGs of the same size)
R="$GRAFT_REPO_ROOT"; cd $R
L=enc0c2,enc1c2,enc2c2,up2c2
echo "B=16"; BENCH_ONLY=$L timeout 200 python tools/bench_conv.py wgrad 20 2>&1 | grep -v amdgpu | cut -c1-75
echo "B=64, 4x the workgroups"; BENCH_B=64 MPU_WGRAD_TAPS_WGS=2048 BENCH_ONLY=$L timeout 300 python tools/bench_conv.py wgrad 20 2>&1 | grep -v amdgpu | cut -c1-75
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py -x -q -k "cfg2 or cfg4" -s 2>&1 | grep -v amdgpu | grep "predict \|passed\|failed\|Error" | cut -c1-300
