R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6g; mkdir -p $O; cd $R
timeout 300 python tools/round6/tail_events.py 2>&1 | tee $O/events.txt | tail -20
