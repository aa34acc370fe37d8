#!/bin/bash
# R4s: conv_halo16p variants: stamps + per-layer A/B against the round-3 schedules (MPU_HALO16P=0) + conv parity
R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out/R4s; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
MPU_STAMPS=1 timeout 300 python tools/round4/stamps16p.py enc1c2,up2c2 2>&1 | grep -v amdgpu.ids | grep -v "tile [1-4]:" | tee $O/stamps.txt
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_unet.py -x -q -m gpu -k "halo16" 2>&1 | tail -3 | tee $O/pytest.log
for p in 0 1 0 1; do
  echo "MPU_HALO16P=$p" | tee -a $O/layers.txt
  MPU_HALO16P=$p BENCH_B=138 BENCH_SCALE=2 BENCH_ONLY=enc1c2,enc2c2,up2c2,up1c2 timeout 200 python tools/bench_conv.py fwd 10 2>&1 | grep -v "amdgpu.ids" | tee -a $O/layers.txt
done
for p in 0 1 0 1; do
  MPU_HALO16P=$p timeout 300 python bench.py --predict-only 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['predict_fuse']; print('p=$p', d['seconds'], 'unet_ms', d['unet_ms'], d['unet_frac_of_mfma_peak'], 'clk', d.get('shader_clock_mhz_during_predict'), d['label_histogram'])" | tee -a $O/predict_ab.txt
done
