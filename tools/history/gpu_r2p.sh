#!/bin/bash
# round-2 call P: conv_pipe on the many-tile predict layers (MPU_PIPE_BIG) vs the halo / glds schedules; geometry re-check
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r2p; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_geometry.py -q 2>&1 | tail -2
L=enc2c1,enc2c2,enc3c1,enc3c2,botc1,botc2,up0c1,up0c2,up0c3,up1c2
for P in 0 256; do MPU_PIPE_BIG=$P BENCH_B=138 BENCH_SCALE=2 BENCH_ONLY=$L timeout 300 python tools/bench_conv.py fwd 5 2>/dev/null > $O/conv_big$P.txt; done
paste $O/conv_big0.txt $O/conv_big256.txt | awk -F'\t' '{print substr($1,1,62), "|", substr($2,38,26)}'
for P in 0 256; do MPU_PIPE_BIG=$P timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-peaks 2> $O/b$P.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['predict_fuse']; print('pipe_big=$P', p['value'], p['seconds'], p['sample_ms'], p['unet_ms'], p['map_fuse_ms'], p['unet_tflops_algorithmic'], p['label_histogram'])"; done
