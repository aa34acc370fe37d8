#!/bin/bash
# R4p: conv_halo16p (persistent 16-row kernel, direct-store epilogue): parity, per-layer and predict A/B
R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out/R4p; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "halo16" > $O/pytest_conv.log 2>&1; echo "conv rc=$?" | tee -a $O/summary.txt; tail -15 $O/pytest_conv.log
timeout 600 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -s -k "persistent_halo16" > $O/pytest_unet.log 2>&1; echo "unet rc=$?" | tee -a $O/summary.txt; tail -15 $O/pytest_unet.log
for h in "0 1" "1 0" "1 1" "0 1" "1 0" "1 1"; do
  set -- $h
  echo "MPU_HALO16=$1 MPU_HALO16P=$2" | tee -a $O/layers.txt
  MPU_HALO16=$1 MPU_HALO16P=$2 BENCH_B=138 BENCH_SCALE=2 BENCH_ONLY=enc1c2,enc2c2,up2c2,up1c2 timeout 200 python tools/bench_conv.py fwd 10 2>&1 | grep -v "amdgpu.ids" | tee -a $O/layers.txt
done
for h in "0 1" "1 1" "0 1" "1 1"; do
  set -- $h
  MPU_HALO16=$1 MPU_HALO16P=$2 timeout 300 python bench.py --predict-only 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['predict_fuse']; print('halo16=$1 p=$2', d['seconds'], 'unet_ms', d['unet_ms'], d['unet_frac_of_mfma_peak'], 'clk', d.get('shader_clock_mhz_during_predict'), d['label_histogram'])" | tee -a $O/predict_ab.txt
done
