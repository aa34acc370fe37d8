#!/bin/bash
# round-2 call S: software-pipelined conv_halo (MPU_HALO_SWP): parity + A/B on predict and train shapes
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r2s; mkdir -p $O
cd $R
export TMPDIR=/tmp
MPU_HALO_SWP=15 timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_unet.py -q -x -k "not subprocess" 2>&1 | tail -3
L=enc1c1,enc1c2,enc2c1,enc2c2,enc3c2,up0c2,up1c2,up2c2,up3c2
for P in 0 15; do MPU_HALO_SWP=$P BENCH_B=138 BENCH_SCALE=2 BENCH_ONLY=$L timeout 300 python tools/bench_conv.py fwd 5 2>/dev/null > $O/conv_p$P.txt; done
echo "-- predict batch shapes (138 x 256^2): SWP 0 | 15"; paste $O/conv_p0.txt $O/conv_p15.txt | awk -F'\t' '{print substr($1,1,62), "|", substr($2,38,26)}'
for P in 0 15; do MPU_HALO_SWP=$P BENCH_ONLY=$L timeout 300 python tools/bench_conv.py fwd 20 2>/dev/null > $O/conv_t$P.txt; done
echo "-- train shapes (16 x 128^2): SWP 0 | 15"; paste $O/conv_t0.txt $O/conv_t15.txt | awk -F'\t' '{print substr($1,1,62), "|", substr($2,38,26)}'
for P in 0 15; do MPU_HALO_SWP=$P timeout 600 python bench.py --no-cpu-baseline --no-peaks 2> $O/b$P.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['predict_fuse']; print('swp=$P train', d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac'], '| predict', p['value'], p['seconds'], p['unet_ms'], p['unet_tflops_algorithmic'])"; done
