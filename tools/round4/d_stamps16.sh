#!/bin/bash
# round 4 call D: where does a conv_halo16 workgroup spend its time? (s_memtime stamps); fix-up kernel with eight waves;
# replay of the non-convolution launches
R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out/R4d; mkdir -p $O
MPU_STAMPS=1 timeout 300 python tools/round4/stamps16.py enc1c2,up2c2,enc2c2 2>&1 | grep -v amdgpu.ids | tee $O/stamps16.txt
timeout 900 python -m pytest tests/test_gpu_replay.py -x -q -m gpu -s -k "non_conv or every_conv" > $O/pytest_replay.log 2>&1; echo "replay rc=$?" | tee -a $O/summary.txt; grep "replay" $O/pytest_replay.log | tail -5; tail -3 $O/pytest_replay.log
timeout 600 python -m pytest tests/test_gpu_geometry.py -x -q -m gpu > $O/pytest_geom.log 2>&1; echo "geometry rc=$?" | tee -a $O/summary.txt
for i in 1 2; do timeout 300 python bench.py --predict-only 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['predict_fuse']; print('predict', d['seconds'], 'unet_ms', d['unet_ms'], 'map_fuse_ms', d['map_fuse_ms'])" | tee -a $O/predict.txt; done
