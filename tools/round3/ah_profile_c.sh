#!/bin/bash
# round 3 calls AH (r03c), AM (r03d): profiles of the round's final kernels (r03d): kernel tables, step sequence, HBM counters, predict, bench line
R="$GRAFT_REPO_ROOT"
bash $R/tools/profile_round.sh r03d > $R/gpurun_out/prof_r03d.log 2>&1
cd $R
timeout 600 python bench.py > gpurun_out/prof_r03d/bench_line.json 2> gpurun_out/prof_r03d/bench.err
tail -1 gpurun_out/prof_r03d/bench_line.json | cut -c1-400
rm -rf gpurun_out/prof_r03d/stats gpurun_out/prof_r03d/fetch gpurun_out/prof_r03d/write gpurun_out/prof_r03d/predict
ls gpurun_out/prof_r03d
