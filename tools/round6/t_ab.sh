# R6t: same-box A/B of the tail forms: 0 serial, 2 optimizer only on the side stream, 1 + first-layer wgrad and deep reductions
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6t; mkdir -p $O; cd $R
B="python bench.py --no-predict --no-cpu-baseline --no-e2e --no-peaks --no-kernel-events --no-graph"
J='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("ms_per_step_median"))'
for i in 1 2 3; do for v in 0 2 1; do
  MPU_TAIL_OVERLAP=$v timeout 300 $B > $O/e_${v}_$i.log 2>&1; echo "eager overlap=$v $(tail -1 $O/e_${v}_$i.log | python -c "$J")"
done; done
