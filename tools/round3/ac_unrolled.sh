#!/bin/bash
# round 3 call AC: staggered conv_halo8 with the taps of a chunk unrolled: parity, per-layer and step A/B against the lockstep schedule
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3ac; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q > $O/pytest_conv.log 2>&1; tail -2 $O/pytest_conv.log
L=enc1c1,enc1c2,enc2c1,enc2c2,up1c2,up2c2
for s in 1 0 1 0; do
  echo "== per layer sched=$s"
  MPU_HALO8_SCHED=$s BENCH_ONLY=$L timeout 300 python tools/bench_conv.py fwd 30 2>&1 | grep -v amdgpu
done
echo "== stamps"
MPU_STAMPS=1 timeout 300 python tools/stamps.py fwd enc1c2,enc2c2 2>&1 | grep -v amdgpu
for s in 1 0 1 0 1 0; do
  MPU_HALO8_SCHED=$s timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sched=$s', d['ms_per_step'], d.get('ms_per_step_median'), d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])"
done
