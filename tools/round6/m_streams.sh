R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for i in 1 2 3; do echo "--- process $i"; timeout 300 python tools/round6/m_streams.py 2>&1 | grep -E "ms per step"; done
