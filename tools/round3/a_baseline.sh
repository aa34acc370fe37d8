#!/bin/bash
# round 3 call A: regression tests of the ADVICE fixes, baseline bench line, conv_halo fixed-cost split
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3a; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_baseline_shapes.py -x -q -k "first_layer or cfg1_bf16" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python -m pytest tests/test_gpu_geometry.py -x -q > $O/pytest_geo.log 2>&1; tail -2 $O/pytest_geo.log
timeout 600 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-600
L=enc1c1,enc1c2,enc2c1,enc2c2,up1c2,up2c2,up3c2,up3c1
BENCH_ONLY=$L timeout 200 python tools/bench_conv.py fwd 20 > $O/conv_full.txt 2>&1
for n in 1 2 9 10; do
  MPU_HALO_DEBUG=$n BENCH_ONLY=$L timeout 200 python tools/bench_conv.py fwd 20 > $O/conv_n$n.txt 2>&1
done
paste $O/conv_full.txt $O/conv_n1.txt $O/conv_n2.txt $O/conv_n9.txt $O/conv_n10.txt | awk '{print $1, $8, $(8+13), $(8+26), $(8+39), $(8+52)}' | column -t
timeout 200 python tools/bench_conv.py wgrad 20 > $O/wgrad_full.txt 2>&1; cat $O/wgrad_full.txt | cut -c1-90
