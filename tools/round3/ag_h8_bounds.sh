#!/bin/bash
# round 3 call AG: which grids should the (now leaner) staggered conv_halo8 take? step A/B over MPU_HALO8_MAX_WGS / _MIN_WGS
R="$GRAFT_REPO_ROOT"; cd $R
for cfg in "192 400" "192 1100" "192 2100" "120 400" "192 400" "192 1100"; do
  set -- $cfg
  MPU_HALO8_MIN_WGS=$1 MPU_HALO8_MAX_WGS=$2 timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('min=$1 max=$2', d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])"
done
