# R5y: the whole GPU suite + smoke on the final tree, and the configs[3] bench line again (bench.py no longer attaches the configs[1]
# counter bytes to other workloads)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5y; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; tail -8 $O/pytest_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --config 3 --no-predict --no-cpu-baseline > $O/bench3.log 2>&1; tail -1 $O/bench3.log > $O/bench_line_configs3.json; cut -c1-200 $O/bench_line_configs3.json
