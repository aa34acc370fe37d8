# usage (on the GPU box): TAG=r06a bash tools/round6/u_profile.sh -> gpurun_out/prof_TAG/*: kernel table, step sequence, PMC traffic, predict table, bench lines
TAG=${TAG:-r06a}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-predict --no-cpu-baseline --no-graph --no-kernel-events --no-peaks --no-e2e"
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- $B --steps 24 --warmup 3 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o f -- $B --steps 3 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o w -- $B --steps 3 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/predict -o p -- python $R/bench.py --predict-only > /dev/null 2>&1
S=$(find $O/stats -name "*.db" | head -1); F=$(find $O/fetch -name "*.db" | head -1); W=$(find $O/write -name "*.db" | head -1); P=$(find $O/predict -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $S 52 > $O/train_step_kernel_stats.txt
python $R/tools/rocpd_sequence.py $S > $O/train_step_sequence.txt 2>&1
python $R/tools/round6/tail_trace.py $S > $O/train_step_tail_timeline.txt 2>&1
python $R/tools/rocpd_stats.py $P 30 > $O/predict_kernel_stats.txt
python $R/tools/rocpd_traffic.py $F $W $O/hbm_traffic_pmc.json > /dev/null
CHECK=0 REPS=2 timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/gf -o f -- python $R/tools/bench_geometry.py > /dev/null 2>&1
CHECK=0 REPS=2 timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/gw -o w -- python $R/tools/bench_geometry.py > /dev/null 2>&1
python $R/tools/geometry_pmc.py $(find $O/gf -name "*.db" | head -1) $(find $O/gw -name "*.db" | head -1) $O/geometry_pmc.json | head -6
rm -rf $O/stats $O/fetch $O/write $O/predict $O/gf $O/gw
# the bench lines below quote the HBM bytes of THIS build's counter passes (bench.py takes the newest profiles/*_pmc.json whose source
# hash matches the tree)
cp $O/hbm_traffic_pmc.json $R/profiles/${TAG}_hbm_traffic_pmc.json; cp $O/geometry_pmc.json $R/profiles/${TAG}_geometry_pmc.json
head -14 $O/train_step_kernel_stats.txt | cut -c1-160; tail -2 $O/train_step_sequence.txt; tail -14 $O/train_step_tail_timeline.txt
cd $R
timeout 900 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench_line.json; cut -c1-400 $O/bench_line.json
timeout 600 python bench.py --config 3 --no-predict --no-cpu-baseline > $O/bench3.log 2>&1; tail -1 $O/bench3.log > $O/bench_line_configs3.json; cut -c1-200 $O/bench_line_configs3.json
timeout 600 python bench.py --config 4 > $O/bench4.log 2>&1; tail -1 $O/bench4.log > $O/bench_line_configs4.json; cut -c1-200 $O/bench_line_configs4.json
