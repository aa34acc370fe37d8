#!/bin/bash
# round-2 call K: wgrad_taps tap-split (half the partial copies): parity + A/B
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r2k; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_unet.py -q -x 2>&1 | tail -2
for S in 1 0; do MPU_WGRAD_TAPS_SPLIT=$S timeout 200 python tools/bench_conv.py wgrad 20 2>/dev/null | grep -E "enc0c2|enc1c2|enc2c2|up3c1|up3c2|up2c2|up1c2|total" > $O/wg$S.txt; done
paste $O/wg1.txt $O/wg0.txt | awk -F'\t' '{print substr($1,1,66), "|", substr($2,40,26)}'
for S in 1 0; do MPU_WGRAD_TAPS_SPLIT=$S timeout 300 python bench.py --steps 40 --warmup 8 --no-predict --no-cpu-baseline 2> $O/b$S.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('taps_split=$S', d['ms_per_step'], d['ms_per_step_median'], d['wgrad']['frac'], d['wgrad']['avg_launch_us'])"; done
