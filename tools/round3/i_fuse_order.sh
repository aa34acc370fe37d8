#!/bin/bash
# round 3 call I: retest (fused Adam, configs[0]); fused back-mapping brick orders (time + FETCH_SIZE); FETCH calibration
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3i; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_unet.py -x -q -k "fused_adam or graphed or f32_train" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py -x -q -k "configs0" > $O/pytest2.log 2>&1; tail -3 $O/pytest2.log
for m in 1 2 3; do
  echo "== MPU_FUSE_MORTON=$m"
  MPU_FUSE_MORTON=$m CHECK=$([ $m = 1 ] && echo 0 || echo 1) timeout 300 python tools/bench_geometry.py 2>&1 | grep -v amdgpu | grep "fast=1\|fast == exact"
done
cd /tmp; export TMPDIR=/tmp
for m in 1 2 3; do
  MPU_FUSE_MORTON=$m CHECK=0 REPS=2 timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_m$m -o f -- python $R/tools/bench_geometry.py > /dev/null 2>&1
  echo "== FETCH_SIZE MPU_FUSE_MORTON=$m"; python $R/tools/rocpd_pmc.py $(find $O/pmc_m$m -name "*.db" | head -1) all 2>&1 | grep -A1 "map_fuse_fast"
done
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_cal -o f -- python $R/tools/probe_fetch.py > $O/cal.log 2>&1; tail -1 $O/cal.log
python $R/tools/rocpd_pmc.py $(find $O/pmc_cal -name "*.db" | head -1) all 2>&1 | grep -A1 "probe_"
rm -rf $O/pmc_m* $O/pmc_cal
