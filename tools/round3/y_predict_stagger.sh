#!/bin/bash
# round 3 call Y: the staggered conv_halo8 on predict-size grids (A/B against the 4-wave kernel)
R="$GRAFT_REPO_ROOT"; cd $R
for mx in 640 100000000 640 100000000; do
  MPU_HALO8_MAX_WGS=$mx timeout 300 python bench.py --predict-only 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['predict_fuse']; print('halo8_max_wgs=$mx', d['seconds'], d['unet_ms'], d['unet_frac_of_mfma_peak'])"
done
