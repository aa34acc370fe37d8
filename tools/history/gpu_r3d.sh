#!/bin/bash
# round-2 call 3D: halo tile-shape knobs on the predict leg
R="$GRAFT_REPO_ROOT"; cd $R; export TMPDIR=/tmp
run() { env "$@" timeout 300 python bench.py --predict-only 2>/dev/null | python -c "
import json,sys; p=json.loads(sys.stdin.read())['predict_fuse']; print('$*', p['seconds'], p['unet_ms'], p['unet_tflops_algorithmic'])"; }
run X=base
run MPU_HALO_VARIANT=1
run MPU_HALO_VARIANT=2
run MPU_HALO_BN64_BELOW=100000
run MPU_BENCH_PREDICT_BATCH=92
run MPU_BENCH_PREDICT_BATCH=69
run X=base2
