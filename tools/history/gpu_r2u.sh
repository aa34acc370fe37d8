#!/bin/bash
# round-2 call U: 1x1 head fused into the last conv_ws epilogue: parity + predict A/B
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r2u; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_unet.py tests/test_gpu_cli.py -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py -q -k "cfg1 or predict" 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for P in 1 0; do MPU_FUSED_HEAD=$P timeout 600 python bench.py --predict-only 2> $O/b$P.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d.get('predict_fuse', d); print('fused_head=$P', p['value'], p['seconds'], p['sample_ms'], p['unet_ms'], p['map_fuse_ms'], p['unet_tflops_algorithmic'], p['label_histogram'])"; done
