#!/bin/bash
# round 3 call M: with grouped launches, fewer / longer wgrad_taps workgroups per layer (fewer partial copies)?
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3m; mkdir -p $O
cd $R
for w in 512 256 128 512 256; do
  MPU_WGRAD_TAPS_WGS=$w timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks > $O/bench_w$w.log 2>&1
  tail -1 $O/bench_w$w.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('taps_wgs=$w', d['ms_per_step'], d['ms_per_step_median'], d['roofline']['kernel_ms_per_step'], d['wgrad']['kernel_ms_per_step'], d['wgrad']['frac'])"
done
timeout 1200 python -m pytest tests/test_gpu_baseline_shapes.py -x -q -s -k "cfg2 or cfg4" 2>&1 | grep -v amdgpu | grep "predict \|passed\|failed\|Error" | cut -c1-300
