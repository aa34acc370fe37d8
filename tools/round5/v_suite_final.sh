# R5v: the whole GPU suite + smoke on the final tree of the round
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5v; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; tail -15 $O/pytest_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
