# R6b: mpu_unet_backward_adam -- equality test, then A/B of the tail overlap on the bench line (same box, alternating)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6b; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_unet.py -q -x -k "backward_adam or fused_adam or graphed" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
B="python bench.py --no-predict --no-cpu-baseline --no-e2e --no-peaks"
for i in 1 2 3; do
  MPU_TAIL_OVERLAP=0 timeout 300 $B > $O/bench_off_$i.log 2>&1; echo "off $i $(tail -1 $O/bench_off_$i.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("ms_per_step_median"))')"
  MPU_TAIL_OVERLAP=1 timeout 300 $B > $O/bench_on_$i.log 2>&1; echo "on  $i $(tail -1 $O/bench_on_$i.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("ms_per_step_median"))')"
done
