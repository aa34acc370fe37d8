#!/bin/bash
# round-2 call 3B: two-channel sampler with 16-byte buffer loads: equality + timing; configs[4]-size predict
R="$GRAFT_REPO_ROOT"; cd $R; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_geometry.py -q 2>&1 | tail -2
C=2 K=5 D=192 timeout 300 python tools/bench_geometry.py 2>&1 | grep -v amdgpu.ids | tail -5
BIG_ONLY=cfg5 timeout 900 python tools/run_big_configs.py 2>&1 | grep -v amdgpu.ids | tail -2
