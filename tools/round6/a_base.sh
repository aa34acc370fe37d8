# R6a: baseline of the round-5 tree on this round's box (train step only)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6a; mkdir -p $O
cd $R
timeout 600 python bench.py --no-predict --no-cpu-baseline --no-e2e > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-600
