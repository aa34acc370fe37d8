#!/bin/bash
# round 3 call C: wgrad_taps 16-byte partial stores; full conv suite; halo8 A/B on the train step; 2-rank bench legs
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3c; mkdir -p $O
cd $R
timeout 300 python tools/bench_conv.py wgrad 20 > $O/wgrad.txt 2>&1; grep -v amdgpu $O/wgrad.txt | cut -c1-75
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_unet.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python -m pytest tests/test_gpu_baseline_shapes.py -x -q -k "first_layer or cfg1" > $O/pytest2.log 2>&1; tail -3 $O/pytest2.log
for h in 0 1; do
  MPU_HALO8=$h timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks > $O/bench_h$h.log 2>&1
  tail -1 $O/bench_h$h.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('halo8=$h', d['ms_per_step'], d['ms_per_step_median'], d['roofline']['kernel_ms_per_step'], d['wgrad']['kernel_ms_per_step'])"
done
timeout 1500 python -m pytest tests/test_gpu_bench_multi.py -x -q > $O/pytest3.log 2>&1; tail -15 $O/pytest3.log
