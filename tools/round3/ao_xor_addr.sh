#!/bin/bash
# round 3 call AO: k-step XOR form of the fragment addresses in conv_halo (4-wave): parity, predict A/B against the previous build
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3ao; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q > $O/pytest_conv.log 2>&1; tail -2 $O/pytest_conv.log
P=$R/multiplanarunet_amd/lib/libmpunet_hip_prev.so
for v in new prev new prev; do
  if [ $v = prev ]; then export MPU_LIB_PATH=$P; else unset MPU_LIB_PATH; fi
  timeout 300 python bench.py --predict-only 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['predict_fuse']; print('$v predict', d['seconds'], d['unet_ms'], d['unet_frac_of_mfma_peak'])"
done
