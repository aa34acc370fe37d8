#!/bin/bash
# round 3 call B: conv_halo8 (8-wave, double-buffered patch) vs conv_halo per layer; parity of the conv suite with it on
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3b; mkdir -p $O
cd $R
L=enc1c1,enc1c2,enc2c1,enc2c2,up1c2,up2c2,up3c1,up3c2
MPU_HALO8=0 BENCH_ONLY=$L timeout 200 python tools/bench_conv.py fwd 30 > $O/conv_h8off.txt 2>&1
MPU_HALO8=1 BENCH_ONLY=$L timeout 200 python tools/bench_conv.py fwd 30 > $O/conv_h8on.txt 2>&1
MPU_HALO8=1 MPU_HALO8_MAX_WGS=4096 BENCH_ONLY=$L timeout 200 python tools/bench_conv.py fwd 30 > $O/conv_h8all.txt 2>&1
paste $O/conv_h8off.txt $O/conv_h8on.txt $O/conv_h8all.txt | grep -v amdgpu | awk '{print $1, $5, $(5+13), $(5+26)}' | column -t
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_unet.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python -m pytest tests/test_gpu_baseline_shapes.py -x -q -k "first_layer or cfg1" > $O/pytest2.log 2>&1; tail -3 $O/pytest2.log
timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-300
