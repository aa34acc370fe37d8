#!/bin/bash
# round-2 call 3A: optimizer in line with the backward pass (mpu_unet_backward_apply): graph == eager, A/B on the step
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r3a; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_unet.py -q -k "graphed or reduces_loss" 2>&1 | tail -3
for rep in 1 2; do for P in 1 0; do MPU_INLINE_ADAM=$P timeout 600 python bench.py --no-cpu-baseline --no-peaks --no-predict 2> $O/b$P.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('inline_adam=$P', d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min'], d['value'], d['guard']['loss_after_timed_steps'])"; done; done
