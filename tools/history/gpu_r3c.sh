#!/bin/bash
# round-2 call 3C: sweep of the split / grid knobs on the train step after conv_pipe (each line: knob, ms/step mean, median, min)
R="$GRAFT_REPO_ROOT"; cd $R; export TMPDIR=/tmp
run() { env "$@" timeout 300 python bench.py --steps 40 --warmup 8 --no-predict --no-cpu-baseline --no-peaks --no-kernel-events 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min'])"; }
run X=base
run MPU_PIPE_WGS=192
run MPU_PIPE_WGS=384
run MPU_PIPE_WGS=512
run MPU_PIPE_MIN_STEPS=9
run MPU_WGRAD_SPLIT_TARGET=384
run MPU_WGRAD_SPLIT_TARGET=768
run MPU_WGRAD_NOSPLIT_TILES=256
run MPU_WGRAD_TAPS_WGS=512
run MPU_WGRAD_TAPS_WGS=128
run MPU_HALO_BN64_BELOW=1024
run MPU_HALO_BN64_BELOW=512
run X=base2
