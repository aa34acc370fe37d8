R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6n; mkdir -p $O; cd $R
B="python bench.py --no-predict --no-cpu-baseline --no-e2e --no-peaks --no-kernel-events"
J='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("ms_per_step_median"), d["config"].get("launch"), d["config"].get("launch_calibration_ms"))'
MPU_SIDE_NORMAL=1 timeout 300 $B --graph > $O/g_n.log 2>&1; echo "forced graph, eager side stream normal priority: $(tail -1 $O/g_n.log | python -c "$J")"
timeout 300 $B --graph > $O/g_h.log 2>&1; echo "forced graph, eager side stream high priority: $(tail -1 $O/g_h.log | python -c "$J")"
MPU_SIDE_NORMAL=1 timeout 300 $B --no-graph > $O/e_n.log 2>&1; echo "eager, side stream normal priority: $(tail -1 $O/e_n.log | python -c "$J")"
timeout 300 $B --no-graph > $O/e_h.log 2>&1; echo "eager, side stream high priority: $(tail -1 $O/e_h.log | python -c "$J")"
