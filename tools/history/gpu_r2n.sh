#!/bin/bash
# round-2 call N: geometry kernels with work-list fix-ups for both; timing + kernel stats
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r2n; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_baseline_shapes.py -q -k "not dice_delta and not cfg1" 2>&1 | tail -4
echo "== 256^3 C=1 K=3"; timeout 300 python tools/bench_geometry.py 2>&1 | grep -v amdgpu.ids | tail -4
echo "== C=2 K=5 D=192"; C=2 K=5 D=192 timeout 300 python tools/bench_geometry.py 2>&1 | grep -v amdgpu.ids | tail -3
cd /tmp
CHECK=0 REPS=3 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o g -- python $R/tools/bench_geometry.py > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof/*/*.db $O/prof/*.db 2>/dev/null | head -1) 2>/dev/null | head -8 | cut -c1-150
