#!/bin/bash
# round 4 call C: conv_halo16 (16-row tiles, staggered halves) parity + per-layer / predict A/B; fixed-point back-mapping A/B
R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out/R4c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "halo16" > $O/pytest_conv.log 2>&1; echo "conv rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_conv.log
timeout 600 python -m pytest tests/test_gpu_replay.py -x -q -m gpu > $O/pytest_replay.log 2>&1; echo "replay rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_replay.log
MPU_HALO16_MIN=1 timeout 600 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -k "bf16_forward_and_step or training_reduces or graphed" > $O/pytest_unet16.log 2>&1; echo "unet(halo16 min 1) rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_unet16.log
timeout 900 python -m pytest tests/test_gpu_geometry.py -x -q -m gpu > $O/pytest_geom.log 2>&1; echo "geometry rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_geom.log
for h in 0 1 0 1; do
  echo "MPU_HALO16=$h" | tee -a $O/layers.txt
  MPU_HALO16=$h BENCH_B=138 BENCH_SCALE=2 BENCH_ONLY=enc1c2,enc2c2,up2c2,up1c2 timeout 200 python tools/bench_conv.py fwd 10 2>&1 | grep -v "amdgpu.ids" | tee -a $O/layers.txt
done
for h in 0 1 0 1; do
  MPU_HALO16=$h timeout 300 python bench.py --predict-only 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['predict_fuse']; print('halo16=$h', d['seconds'], 'unet_ms', d['unet_ms'], d['unet_frac_of_mfma_peak'], 'map_fuse_ms', d['map_fuse_ms'], d['label_histogram'])" | tee -a $O/predict_ab.txt
done
for f in 0 1 0 1; do
  echo "MPU_FUSE_FX=$f" | tee -a $O/geom.txt
  MPU_FUSE_FX=$f timeout 300 python tools/bench_geometry.py 2>&1 | grep -v "amdgpu.ids" | tee -a $O/geom.txt
done
