#!/bin/bash
# round 4 calls E..: s_memtime stamps of conv_halo16 on predict-size layers (+ parity of the halo16 cases, per-layer time)
R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out/R4e; mkdir -p $O
MPU_HALO16_MIN=1 timeout 300 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "halo16_cases or halo16_at" 2>&1 | tail -2
MPU_STAMPS=1 timeout 300 python tools/round4/stamps16.py enc1c2,up2c2 2>&1 | grep -v amdgpu.ids | tee $O/stamps16.txt
BENCH_B=138 BENCH_SCALE=2 BENCH_ONLY=enc1c2,enc2c2,up2c2 timeout 200 python tools/bench_conv.py fwd 10 2>&1 | grep -v "amdgpu.ids" | tee -a $O/stamps16.txt
