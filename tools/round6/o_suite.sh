# R6o: the whole GPU suite + smoke
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6o; mkdir -p $O; cd $R
timeout 2700 python -m pytest tests -q -m gpu -x > $O/pytest_full.log 2>&1; tail -8 $O/pytest_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
