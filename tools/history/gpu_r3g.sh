#!/bin/bash
# round-2 call 3G: 8-row tiles for the up-conv halo kernels on large grids
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r3g; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_conv.py -q -k "deep_level or forward_dgrad" 2>&1 | tail -2
L=up0c1,up3c1
for P in 1000000000 1; do MPU_HALO_UP8_MIN=$P BENCH_B=138 BENCH_SCALE=2 BENCH_ONLY=$L timeout 300 python tools/bench_conv.py fwd 5 2>/dev/null > $O/p$P.txt; done
echo "-- predict shapes: TH=4 | TH=8"; paste $O/p1000000000.txt $O/p1.txt | awk -F'\t' '{print substr($1,1,62), "|", substr($2,38,26)}'
for P in 1000000000 2048; do MPU_HALO_UP8_MIN=$P timeout 600 python bench.py --predict-only 2>/dev/null | python -c "
import json,sys; p=json.loads(sys.stdin.read())['predict_fuse']; print('up8_min=$P', p['seconds'], p['unet_ms'], p['unet_tflops_algorithmic'], p['label_histogram'])"; done
