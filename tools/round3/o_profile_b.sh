#!/bin/bash
# round 3 call J: whole GPU suite, profile round r03a (kernel stats, sequence, PMC traffic), geometry PMC, bench line
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3o; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; tail -4 $O/pytest_all.log
bash tools/profile_round.sh r03b > $O/profile_round.log 2>&1; tail -18 $O/profile_round.log | cut -c1-150
cd /tmp; export TMPDIR=/tmp
CHECK=0 REPS=2 timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/gf -o f -- python $R/tools/bench_geometry.py > /dev/null 2>&1
CHECK=0 REPS=2 timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/gw -o w -- python $R/tools/bench_geometry.py > /dev/null 2>&1
python $R/tools/geometry_pmc.py $(find $O/gf -name "*.db" | head -1) $(find $O/gw -name "*.db" | head -1) $R/gpurun_out/prof_r03b/geometry_pmc.json | head -30
rm -rf $O/gf $O/gw
cd $R
timeout 600 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $R/gpurun_out/prof_r03b/bench_line.json; cut -c1-400 $R/gpurun_out/prof_r03b/bench_line.json
find $R/gpurun_out/prof_r03b -name "*.db" -size +20M -delete
