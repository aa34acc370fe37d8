R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6am; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_unet.py tests/test_gpu_cli.py tests/test_gpu_replay.py -q -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log



B="python $R/bench.py --no-predict --no-cpu-baseline --no-e2e --no-peaks --no-kernel-events --no-graph"

