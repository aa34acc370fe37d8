#!/bin/bash
# round-2 call Z: straight-line single-view map / accumulate kernels: equality + sharded-predict tests + timing
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r2z; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_distributed.py tests/test_gpu_cli.py -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py -q -k "predict or cfg4 or cfg2" 2>&1 | tail -3
timeout 300 python tools/bench_geometry.py 2>&1 | grep -v amdgpu.ids | tail -5
