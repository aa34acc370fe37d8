#!/bin/bash
# round 4 call K: tap-combined up-conv (conv_halo UPQ): parity, predict A/B (MPU_UPQ=0 / 1), network-level inference tests
R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out/R4k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "tap_combined" 2>&1 | tail -15 | cut -c1-250 | tee $O/pytest_upq.log
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_baseline_shapes.py -x -q -m gpu -k "bf16_forward_and_step or fused_pool or six_view or cf2 or cfg2_predict" 2>&1 | tail -5 | cut -c1-250 | tee -a $O/pytest_upq.log
for u in 0 1 0 1; do
  MPU_UPQ=$u timeout 300 python bench.py --predict-only 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['predict_fuse']; print('upq=$u', d['seconds'], 'unet_ms', d['unet_ms'], d['unet_frac_of_mfma_peak'], 'clock', d.get('shader_clock_mhz_during_predict'), d['label_histogram'])" | tee -a $O/predict_ab.txt
done
