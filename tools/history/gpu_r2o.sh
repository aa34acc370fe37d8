#!/bin/bash
# round-2 call O: are the geometry kernels bound by the ALU or by the gather path? (debug: coalesced dummy addresses)
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r2o; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for D in 0 1; do
MPU_GEOM_DEBUG=$D CHECK=0 REPS=3 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof$D -o g -- python $R/tools/bench_geometry.py > /dev/null 2>&1
echo "== MPU_GEOM_DEBUG=$D"; python $R/tools/rocpd_stats.py $(ls $O/prof$D/*/*.db $O/prof$D/*.db 2>/dev/null | head -1) 2>/dev/null | grep "sample_fast\|map_fuse_fast" | cut -c1-150
done
MPU_GEOM_DEBUG=0 CHECK=0 REPS=2 timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE -d $O/p5 -o p -- python $R/tools/bench_geometry.py > /dev/null 2>&1
python $R/tools/rocpd_pmc.py $(ls $O/p5/*/*.db $O/p5/*.db 2>/dev/null | head -1) all 2>&1 | grep -A1 "sample_fast\|map_fuse_fast" | cut -c1-200
