#!/bin/bash
# round 3 call AH: profiles of the round's final kernels (r03c): kernel tables, step sequence, HBM counters, predict, bench line
R="$GRAFT_REPO_ROOT"
bash $R/tools/profile_round.sh r03c > $R/gpurun_out/prof_r03c.log 2>&1
cd $R
timeout 600 python bench.py > gpurun_out/prof_r03c/bench_line.json 2> gpurun_out/prof_r03c/bench.err
tail -1 gpurun_out/prof_r03c/bench_line.json | cut -c1-400
rm -rf gpurun_out/prof_r03c/stats gpurun_out/prof_r03c/fetch gpurun_out/prof_r03c/write gpurun_out/prof_r03c/predict
ls gpurun_out/prof_r03c
