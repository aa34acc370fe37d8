#!/bin/bash
# round-2 call R: fused 2x2 max-pool output of the inference conv epilogues: parity + predict bench A/B
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r2r; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_unet.py tests/test_gpu_conv.py tests/test_gpu_geometry.py -q 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_baseline_shapes.py -q -k "cfg1" 2>&1 | tail -3
for P in 1 0; do MPU_FUSED_POOL=$P timeout 600 python bench.py --predict-only 2> $O/b$P.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d.get('predict_fuse', d); print('fused_pool=$P', p['value'], p['seconds'], p['sample_ms'], p['unet_ms'], p['map_fuse_ms'], p['unet_tflops_algorithmic'], p['label_histogram'])"; done
