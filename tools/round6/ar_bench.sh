R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6ar; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_bench_multi.py -q -x -k single_gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
