#!/bin/bash
# round 3 call P: halo8 fragment reads two k-steps ahead (three register sets) vs one k-step ahead (previous build)
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3p; mkdir -p $O
cd $R
L=enc1c1,enc1c2,enc2c1,enc2c2,up1c2,up2c2
for lib in prev new prev new; do
  if [ $lib = prev ]; then export MPU_LIB_PATH=$R/multiplanarunet_amd/lib_ab/libmpunet_hip_prev.so; else unset MPU_LIB_PATH; fi
  echo "== $lib"; BENCH_ONLY=$L timeout 200 python tools/bench_conv.py fwd 30 2>&1 | grep -v amdgpu | cut -c1-72
done
unset MPU_LIB_PATH
MPU_STAMPS=1 timeout 300 python tools/stamps.py fwd enc1c2,up2c2 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -k "halo8" 2>&1 | tail -2
for lib in prev new prev new; do
  if [ $lib = prev ]; then export MPU_LIB_PATH=$R/multiplanarunet_amd/lib_ab/libmpunet_hip_prev.so; else unset MPU_LIB_PATH; fi
  timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks > $O/bench_$lib.log 2>&1
  tail -1 $O/bench_$lib.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['ms_per_step_median'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])"
done
