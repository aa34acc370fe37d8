#!/bin/bash
# round-2 call 3H: 8-row up-conv tiles at configs[1] sizes (threshold sweep on the train step)
R="$GRAFT_REPO_ROOT"; cd $R; export TMPDIR=/tmp
run() { env "$@" timeout 300 python bench.py --steps 40 --warmup 8 --no-predict --no-cpu-baseline --no-peaks --no-kernel-events 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min'])"; }
run MPU_HALO_UP8_MIN=2048
run MPU_HALO_UP8_MIN=1024
run MPU_HALO_UP8_MIN=512
run MPU_HALO_UP8_MIN=1
run MPU_HALO_UP8_MIN=2048
