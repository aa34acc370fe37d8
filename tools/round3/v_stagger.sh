#!/bin/bash
# round 3 call V: conv_halo8 with the two halves one phase apart (MPU_HALO8_SCHED=1): parity, per-layer A/B, stamps, step A/B
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3v; mkdir -p $O
cd $R
MPU_HALO8_SCHED=1 timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -k "halo8 or concat or CASES or conv" > $O/pytest_sched1.log 2>&1; tail -3 $O/pytest_sched1.log
L=enc1c1,enc1c2,enc2c1,enc2c2,up1c2,up2c2
for s in 0 1 0 1; do
  echo "== per layer sched=$s"
  MPU_HALO8_SCHED=$s BENCH_ONLY=$L timeout 300 python tools/bench_conv.py fwd 30 2>&1 | grep -v amdgpu
done
for s in 0 1; do
  echo "== stamps sched=$s"
  MPU_HALO8_SCHED=$s MPU_STAMPS=1 timeout 300 python tools/stamps.py fwd enc1c2,enc2c2,up2c2 2>&1 | grep -v amdgpu
done
for s in 0 1 0 1; do
  MPU_HALO8_SCHED=$s timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks > $O/bench_$s.log 2>&1
  tail -1 $O/bench_$s.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sched=$s', d['ms_per_step'], d.get('ms_per_step_median'), d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])"
done
