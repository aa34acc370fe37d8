#!/bin/bash
# round-2 call I: batched weight-gradient reduction (parity + A/B), Dice-delta test
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r2i; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_distributed.py tests/test_gpu_baseline_shapes.py -q -rA -k "not cfg4 and not cfg2" 2>&1 | grep -E "passed|failed|FAILED|ERROR|bf16 6-view|cfg1 bf16" | tail -12
for D in 1 0; do MPU_WGRAD_BATCHED_REDUCE=$D timeout 300 python bench.py --steps 40 --warmup 8 --no-predict --no-cpu-baseline 2> $O/b$D.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batched_reduce=$D', d['ms_per_step'], d['ms_per_step_median'], d['wgrad']['frac'], d['wgrad']['avg_launch_us'])"; done
