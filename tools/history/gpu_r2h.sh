#!/bin/bash
# round-2 call H: two-pass map+fuse, Dice-delta test, wgrad-overlap A/B (conv split-K now has its own scratch)
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r2h; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_baseline_shapes.py tests/test_gpu_unet.py tests/test_gpu_cli.py -q -rA 2>&1 | grep -E "passed|failed|FAILED|ERROR|bf16 6-view|predict " | tail -12
timeout 300 python bench.py --predict-only > $O/predict.json 2> $O/predict.err; python -c "
import json
d=json.load(open('gpurun_out/r2h/predict.json'))['predict_fuse']; print({k: d[k] for k in ('value','seconds','sample_ms','unet_ms','map_fuse_ms','map_fuse_frac_of_hbm_peak')})"
for OV in 0 1; do MPU_WGRAD_OVERLAP=$OV timeout 300 python bench.py --steps 40 --warmup 8 --no-predict --no-cpu-baseline 2> $O/b$OV.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wgrad_overlap=$OV', d['ms_per_step'], d['ms_per_step_median'], d['guard'])"; done
