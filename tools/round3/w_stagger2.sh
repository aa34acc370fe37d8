#!/bin/bash
# round 3 call W: conv_halo8 SCHED 1 with the DMA requests in the compute phase (4 weight stages): parity, per-layer A/B over
# the s_setprio variants, stamps, step A/B
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3w; mkdir -p $O
cd $R
MPU_HALO8_SCHED=1 timeout 600 python -m pytest tests/test_gpu_conv.py -x -q > $O/pytest_sched1.log 2>&1; tail -3 $O/pytest_sched1.log
L=enc1c1,enc1c2,enc2c1,enc2c2,up1c2,up2c2
for cfg in "0 0" "1 0" "1 1" "1 2" "0 0" "1 0"; do
  set -- $cfg
  echo "== per layer sched=$1 prio=$2"
  MPU_HALO8_SCHED=$1 MPU_HALO8_PRIO=$2 BENCH_ONLY=$L timeout 300 python tools/bench_conv.py fwd 30 2>&1 | grep -v amdgpu
done
echo "== stamps sched=1 prio=0"
MPU_HALO8_SCHED=1 MPU_STAMPS=1 timeout 300 python tools/stamps.py fwd enc1c2,enc2c2,up2c2 2>&1 | grep -v amdgpu
for cfg in "0 0" "1 0" "0 0" "1 0"; do
  set -- $cfg
  MPU_HALO8_SCHED=$1 MPU_HALO8_PRIO=$2 timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks > $O/bench_$1_$2.log 2>&1
  tail -1 $O/bench_$1_$2.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sched=$1 prio=$2', d['ms_per_step'], d.get('ms_per_step_median'), d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])"
done
