#!/bin/bash
# round 3 calls AI, AN: conv_halo8 variants (one barrier per tap; weight requests in the load phase): parity, per-layer and step A/B against the previous build
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3ai; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q > $O/pytest_conv.log 2>&1; tail -2 $O/pytest_conv.log
P=$R/multiplanarunet_amd/lib/libmpunet_hip_prev.so
L=enc1c1,enc1c2,enc2c1,enc2c2,up1c2,up2c2
for v in new prev new prev; do
  if [ $v = prev ]; then export MPU_LIB_PATH=$P; else unset MPU_LIB_PATH; fi
  echo "== per layer $v"
  BENCH_ONLY=$L timeout 300 python tools/bench_conv.py fwd 30 2>&1 | grep -v amdgpu
done
for v in new prev new prev new prev; do
  if [ $v = prev ]; then export MPU_LIB_PATH=$P; else unset MPU_LIB_PATH; fi
  timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v train', d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])"
done
