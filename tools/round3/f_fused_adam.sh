#!/bin/bash
# round 3 call F: fused Adam+pack, early epilogue constants; new tests (configs[0], cf=2, replay, fused == separate)
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3f; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_conv.py tests/test_gpu_replay.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 1500 python -m pytest tests/test_gpu_baseline_shapes.py -x -q -s -k "first_layer or cfg1 or configs0 or cf2" > $O/pytest2.log 2>&1; tail -3 $O/pytest2.log; grep "cf=2\|cfg1 dispatch" $O/pytest2.log | cut -c1-400
for f in 0 1; do
  MPU_FUSED_ADAM=$f timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks > $O/bench_f$f.log 2>&1
  tail -1 $O/bench_f$f.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused_adam=$f', d['ms_per_step'], d['ms_per_step_median'], d['roofline']['kernel_ms_per_step'], d['wgrad']['kernel_ms_per_step'])"
done
timeout 300 python bench.py --cf 2 --batch 8 --no-predict --no-cpu-baseline --no-peaks > $O/bench_cf2.log 2>&1; tail -1 $O/bench_cf2.log | cut -c1-700
timeout 1500 python -m pytest tests/test_gpu_bench_multi.py -x -q > $O/pytest3.log 2>&1; tail -5 $O/pytest3.log
