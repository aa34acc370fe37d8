# R6f (as R6d, after the restructured tail): tail overlap with the optimizer capped at one workgroup per CU + wgrad_taps at 74 KB of LDS: tests, A/B, timeline
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6f; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_conv.py -q -x -k "backward_adam or fused_adam or graphed or wgrad or staggered" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
B="python bench.py --no-predict --no-cpu-baseline --no-e2e --no-peaks"
for i in 1 2 3; do
  MPU_TAIL_OVERLAP=0 timeout 300 $B > $O/bench_off_$i.log 2>&1; echo "off $i $(tail -1 $O/bench_off_$i.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("ms_per_step_median"))')"
  MPU_TAIL_OVERLAP=1 timeout 300 $B > $O/bench_on_$i.log 2>&1; echo "on  $i $(tail -1 $O/bench_on_$i.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("ms_per_step_median"))')"
done
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-predict --no-cpu-baseline --no-e2e --no-peaks --no-kernel-events --steps 10 --warmup 3"
for v in 1 0; do
  MPU_TAIL_OVERLAP=$v rocprofv3 --kernel-trace -d $O/t$v -o t -- $B > $O/bench_$v.log 2>&1
  DB=$(find $O/t$v -name "*.db" | head -1)
  python $R/tools/round6/tail_trace.py $DB > $O/tail_$v.txt 2>&1; echo "== overlap $v"; cat $O/tail_$v.txt
  rm -rf $O/t$v
done
