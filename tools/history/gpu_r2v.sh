#!/bin/bash
# round-2 call V: Morton order of the brick columns in the fused back-mapping: equality + timing + FETCH_SIZE
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r2v; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_geometry.py -q 2>&1 | tail -2
for M in 1 0; do echo "== MPU_FUSE_MORTON=$M"; MPU_FUSE_MORTON=$M timeout 300 python tools/bench_geometry.py 2>&1 | grep -v amdgpu.ids | tail -3; done
echo "== D=200 (ragged grid)"; D=200 timeout 300 python tools/bench_geometry.py 2>&1 | grep -v amdgpu.ids | tail -3
cd /tmp
for M in 1 0; do MPU_FUSE_MORTON=$M CHECK=0 REPS=2 timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/p$M -o p -- python $R/tools/bench_geometry.py > /dev/null 2>&1
python $R/tools/rocpd_pmc.py $(ls $O/p$M/*/*.db $O/p$M/*.db 2>/dev/null | head -1) all 2>&1 | grep -A1 "map_fuse_fast" | cut -c1-200; done
