#!/bin/bash
# round 3 call AB: the lean instantiation of the staggered conv_halo8 (no stamps / run-time switches) against the instrumented one
R="$GRAFT_REPO_ROOT"; cd $R
L=enc1c1,enc1c2,enc2c1,enc2c2,up1c2,up2c2
for v in lean dev lean dev; do
  echo "== per layer $v"
  if [ $v = dev ]; then export MPU_HALO8_PRIO=2; else unset MPU_HALO8_PRIO; fi
  BENCH_ONLY=$L timeout 300 python tools/bench_conv.py fwd 30 2>&1 | grep -v amdgpu
done
unset MPU_HALO8_PRIO
timeout 300 python -m pytest tests/test_gpu_conv.py -x -q -k halo8 2>&1 | tail -2
for v in lean dev lean dev lean dev; do
  if [ $v = dev ]; then export MPU_HALO8_PRIO=2; else unset MPU_HALO8_PRIO; fi
  timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d.get('ms_per_step_median'), d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])"
done
