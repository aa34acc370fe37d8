# R6h: the overlapped tail, graph replay vs eager launches (same box, alternating)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6h; mkdir -p $O; cd $R
B="python bench.py --no-predict --no-cpu-baseline --no-e2e --no-peaks --no-kernel-events"
J='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("ms_per_step_median"), d["config"].get("launch"))'
for i in 1 2; do
  for v in 0 1; do
    MPU_TAIL_OVERLAP=$v timeout 300 $B > $O/g_$v_$i.log 2>&1; echo "graph overlap=$v $(tail -1 $O/g_$v_$i.log | python -c "$J")"
    MPU_TAIL_OVERLAP=$v timeout 300 $B --no-graph > $O/e_$v_$i.log 2>&1; echo "eager overlap=$v $(tail -1 $O/e_$v_$i.log | python -c "$J")"
  done
done
