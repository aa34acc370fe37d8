#!/bin/bash
# round-2 call L: straight-line geometry kernels (sample_fast, map_fuse_fast + fix-up): bit-exactness + timing
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r2l; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_geometry.py -q 2>&1 | tail -4
echo "== 256^3 C=1 K=3 brick 4x4x64"; timeout 300 python tools/bench_geometry.py 2>&1 | tail -5
echo "== brick 8x8x16"; MPU_FUSE_BRICK=1 CHECK=0 timeout 300 python tools/bench_geometry.py 2>&1 | tail -2
echo "== C=2 K=5 D=192"; C=2 K=5 D=192 timeout 300 python tools/bench_geometry.py 2>&1 | tail -4
cd /tmp && CHECK=0 REPS=3 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o g -- python $R/tools/bench_geometry.py > /dev/null 2>&1
cd $R && python tools/rocpd_stats.py $(ls $O/prof/*/*.db $O/prof/*.db 2>/dev/null | head -1) 2>/dev/null | head -8 | cut -c1-150
