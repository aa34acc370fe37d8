#!/bin/bash
# round-2 call 3K: kernel trace of BASELINE configs[3] (train B=32 256^2) and configs[4] (predict 512^3 x 2, K=5) at full size
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r3k; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o b -- python $R/tools/run_big_configs.py > $O/run.log 2>&1
grep -v amdgpu.ids $O/run.log | tail -3
python $R/tools/rocpd_stats.py $(ls $O/prof/*/*.db $O/prof/*.db 2>/dev/null | head -1) 36 > $O/big_configs_kernel_stats.txt 2>&1
head -20 $O/big_configs_kernel_stats.txt | cut -c1-150
