# usage (on the GPU box): bash tools/profile_round.sh TAG   -> gpurun_out/prof_TAG/{stats,fetch,write,predict} + text summaries
TAG=$1; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-predict --no-cpu-baseline --no-graph --no-kernel-events --no-peaks --no-e2e"
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- $B --steps 10 --warmup 3 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o f -- $B --steps 3 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o w -- $B --steps 3 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/predict -o p -- python $R/bench.py --predict-only > /dev/null 2>&1
S=$(find $O/stats -name "*.db" | head -1); F=$(find $O/fetch -name "*.db" | head -1); W=$(find $O/write -name "*.db" | head -1); P=$(find $O/predict -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $S 48 > $O/train_step_kernel_stats.txt
python $R/tools/rocpd_sequence.py $S > $O/train_step_sequence.txt 2>&1
python $R/tools/rocpd_stats.py $P 30 > $O/predict_kernel_stats.txt
python $R/tools/rocpd_traffic.py $F $W $O/hbm_traffic_pmc.json > /dev/null
head -14 $O/train_step_kernel_stats.txt; tail -2 $O/train_step_sequence.txt
