# R6s: two-fork tail (c8 beside glds; lean reduce + adam beside taps): tests, A/B (eager and graph), event timeline
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6s; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_conv.py tests/test_gpu_distributed.py -q -x -k "backward_adam or fused_adam or graphed or wgrad or staggered or grouped" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="python bench.py --no-predict --no-cpu-baseline --no-e2e --no-peaks --no-kernel-events"
J='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("ms_per_step_median"), d["config"].get("launch"), d["config"].get("launch_calibration_ms"))'
for i in 1 2; do
  for v in 0 1; do
    MPU_TAIL_OVERLAP=$v timeout 300 $B > $O/c_${v}_$i.log 2>&1; echo "overlap=$v $(tail -1 $O/c_${v}_$i.log | python -c "$J")"
  done
done
timeout 300 python tools/round6/tail_events.py 2>&1 | grep eager | tail -2
