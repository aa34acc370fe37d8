#!/bin/bash
# round 3 call Z: kernel sequence of the default-YAML network's train step (complexity_factor 2, 8 slices of 128x128)
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3z; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-predict --no-cpu-baseline --no-graph --no-kernel-events --no-peaks --cf 2"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o s -- $B --steps 10 --warmup 3 > /dev/null 2>&1
S=$(find $O/stats -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $S 48 > $O/cf2_train_step_kernel_stats.txt
python $R/tools/rocpd_sequence.py $S > $O/cf2_train_step_sequence.txt 2>&1
rm -rf $O/stats
cd $R
cut -c1-110 $O/cf2_train_step_sequence.txt | head -140; tail -3 $O/cf2_train_step_sequence.txt
