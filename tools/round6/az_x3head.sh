# R6az: the head_bn_* passes in f32 storage (dtype bf16x3): A/B tests of both dtypes, the x3 tests, x3 step time with / without
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6az; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_unet.py -q -x -m gpu -s -k "training_head_without or pool_backward_without" > $O/pytest_head.log 2>&1; grep -E "fused head|pool backward|passed|failed|Error|assert" $O/pytest_head.log | head -20
timeout 2400 python -m pytest tests/test_gpu_unet.py tests/test_gpu_replay.py tests/test_gpu_baseline_shapes.py tests/test_gpu_pipeline.py -q -x -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="python $R/bench.py --no-predict --no-cpu-baseline --no-e2e --no-peaks --no-kernel-events --no-graph --dtype bf16x3 --steps 20 --warmup 5"
for i in 1 2; do for X in 0 1; do
  MPU_HEAD_TRAIN_FUSED=$X $B 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("x3 fused='$X'", d["ms_per_step"], d["ms_per_step_median"], d["ms_per_step_min"])'
done; done
B="python $R/bench.py --no-predict --no-cpu-baseline --no-e2e --no-peaks --no-kernel-events --no-graph"
for i in 1 2; do $B 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("bf16", d["ms_per_step"], d["ms_per_step_median"], d["ms_per_step_min"])'; done
