#!/bin/bash
# round 5 call P: profiles of the round's kernels: train step kernel table + sequence + HBM counters (tools/profile_round.sh),
# geometry PMC, bench line, and BASELINE configs[3] / configs[4] at full size (bench lines + kernel tables)
TAG=${TAG:-r05a}
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
cd $R
bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1; tail -6 $O/profile_round.log | cut -c1-150
cd /tmp; export TMPDIR=/tmp
CHECK=0 REPS=2 timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/gf -o f -- python $R/tools/bench_geometry.py > /dev/null 2>&1
CHECK=0 REPS=2 timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/gw -o w -- python $R/tools/bench_geometry.py > /dev/null 2>&1
python $R/tools/geometry_pmc.py $(find $O/gf -name "*.db" | head -1) $(find $O/gw -name "*.db" | head -1) $O/geometry_pmc.json | head -12
timeout 600 rocprofv3 --kernel-trace --stats -d $O/c3 -o c -- python $R/bench.py --config 3 --no-predict --no-cpu-baseline --no-peaks --no-graph --no-kernel-events --no-e2e --steps 5 --warmup 2 > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find $O/c3 -name "*.db" | head -1) 30 > $O/configs3_train_step_kernel_stats.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $O/c4 -o c -- python $R/bench.py --config 4 > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find $O/c4 -name "*.db" | head -1) 24 > $O/configs4_predict_kernel_stats.txt
rm -rf $O/gf $O/gw $O/c3 $O/c4 $O/stats $O/fetch $O/write $O/predict
cd $R
# the bench lines below quote the HBM bytes of THIS build's counter passes (bench.py takes the newest profiles/*_pmc.json whose source
# hash matches the tree)
cp $O/hbm_traffic_pmc.json profiles/${TAG}_hbm_traffic_pmc.json; cp $O/geometry_pmc.json profiles/${TAG}_geometry_pmc.json
timeout 600 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench_line.json; cut -c1-300 $O/bench_line.json
timeout 600 python bench.py --config 3 --no-predict --no-cpu-baseline > $O/bench3.log 2>&1; tail -1 $O/bench3.log > $O/bench_line_configs3.json; cut -c1-300 $O/bench_line_configs3.json
timeout 900 python bench.py --config 4 > $O/bench4.log 2>&1; tail -1 $O/bench4.log > $O/bench_line_configs4.json; cut -c1-300 $O/bench_line_configs4.json
ls $O
