# R6aw: the whole GPU suite + smoke on the build with XCD-contiguous tiles, the unsplit up-conv data gradient, the head_bn_* passes and
# the pool-backward recompute; then the round's profile call (TAG=r06e)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6aw; mkdir -p $O; cd $R
timeout 2700 python -m pytest tests -q -m gpu -x > $O/pytest_full.log 2>&1; tail -6 $O/pytest_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
TAG=${TAG:-r06e} bash tools/round6/u_profile.sh
