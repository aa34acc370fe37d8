#!/bin/bash
# round-2 call Q: selective many-tile conv_pipe dispatch: parity + predict bench
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r2q; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py -q -k "deep_level" 2>&1 | tail -3
for P in 1 0; do MPU_PIPE_BIG=$P timeout 600 python bench.py --predict-only 2> $O/b$P.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d.get('predict_fuse', d); print('pipe_big=$P', p['value'], p['seconds'], p['sample_ms'], p['unet_ms'], p['map_fuse_ms'], p['unet_tflops_algorithmic'], p['label_histogram'])"; done
