#!/bin/bash
# round 3 call AL: BatchNorm finalize + apply as one launch (levels >= 1): network parity tests, step A/B (MPU_FUSED_BN_APPLY)
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3al; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_unet.py tests/test_gpu_baseline_shapes.py -x -q -k "not predict_properties and not cfg4 and not cfg3" > $O/pytest_net.log 2>&1; tail -3 $O/pytest_net.log
for s in 1 0 1 0 1 0; do
  MPU_FUSED_BN_APPLY=$s timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused_bn_apply=$s', d['ms_per_step'], d.get('ms_per_step_median'))"
done
