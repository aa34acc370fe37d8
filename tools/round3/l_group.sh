#!/bin/bash
# round 3 call L: grouped weight-gradient launches: parity suite + A/B on the train step; predict property tests
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3l; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_replay.py tests/test_gpu_distributed.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 1500 python -m pytest tests/test_gpu_baseline_shapes.py -x -q -s -k "first_layer or cfg1 or cfg3 or cf2 or configs0" > $O/pytest2.log 2>&1; tail -3 $O/pytest2.log
for g in 0 1 0 1; do
  MPU_WGRAD_GROUP=$g timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks > $O/bench_g$g.log 2>&1
  tail -1 $O/bench_g$g.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('group=$g', d['ms_per_step'], d['ms_per_step_median'], d['roofline']['kernel_ms_per_step'], d['wgrad']['kernel_ms_per_step'], d['wgrad']['frac'])"
done
timeout 1200 python -m pytest tests/test_gpu_baseline_shapes.py -x -q -s -k "cfg2 or cfg4" 2>&1 | grep -v amdgpu | grep "predict \|passed\|failed\|Error" | cut -c1-300
