#!/bin/bash
# round-2 call M: straight-line geometry kernels, kind-1 axes handled; PMC counters of the two kernels
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r2m; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_geometry.py -q 2>&1 | tail -4
echo "== 256^3 C=1 K=3 brick cfg 2 (default)"; timeout 300 python tools/bench_geometry.py 2>&1 | grep -v amdgpu.ids | tail -4
for B in 0 1; do echo "== brick cfg $B"; MPU_FUSE_BRICK=$B CHECK=0 timeout 300 python tools/bench_geometry.py 2>&1 | grep -v amdgpu.ids | tail -1; done
cd /tmp
CHECK=0 REPS=3 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o g -- python $R/tools/bench_geometry.py > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof/*/*.db $O/prof/*.db 2>/dev/null | head -1) 2>/dev/null | head -7 | cut -c1-150
CHECK=0 REPS=2 timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/p1 -o p -- python $R/tools/bench_geometry.py > /dev/null 2>&1
CHECK=0 REPS=2 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU -d $O/p2 -o p -- python $R/tools/bench_geometry.py > /dev/null 2>&1
CHECK=0 REPS=2 timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $O/p3 -o p -- python $R/tools/bench_geometry.py > /dev/null 2>&1
CHECK=0 REPS=2 timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/p4 -o p -- python $R/tools/bench_geometry.py > /dev/null 2>&1
for d in p1 p2 p3 p4; do f=$(ls $O/$d/*/*.db $O/$d/*.db 2>/dev/null | head -1); echo "-- $d"; python $R/tools/rocpd_pmc.py $f all 2>&1 | grep -A1 "sample_fast\|map_fuse" | cut -c1-260; done
