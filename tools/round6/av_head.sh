# R6av: training head without the last post-BatchNorm tensor (head_bn_*) and pool backward without the summed-gradient tensor: the A/B
# tests, the unet / replay / baseline suites, step A/B (both off / head only / both on)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6av; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_unet.py -q -x -m gpu -s -k "training_head_without or pool_backward_without" > $O/pytest_head.log 2>&1; grep -E "fused head|pool backward|passed|failed|Error|assert" $O/pytest_head.log | head -20
timeout 2400 python -m pytest tests/test_gpu_unet.py tests/test_gpu_replay.py tests/test_gpu_baseline_shapes.py tests/test_gpu_pipeline.py -q -x -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="python $R/bench.py --no-predict --no-cpu-baseline --no-e2e --no-peaks --no-kernel-events --no-graph"
for i in 1 2 3; do for X in 00 10 11; do
  MPU_HEAD_TRAIN_FUSED=${X:0:1} MPU_POOL_BWD_RECOMPUTE=${X:1:1} $B 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("fused='$X'", d["ms_per_step"], d["ms_per_step_median"], d["ms_per_step_min"], d["guard"]["loss_after_timed_steps"])'
done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- $B --steps 24 --warmup 3 > /dev/null 2>&1
S=$(find $O/stats -name "*.db" | head -1)
python $R/tools/rocpd_sequence.py $S > $O/seq.txt 2>&1; sed -n 36,48p $O/seq.txt; sed -n 70,90p $O/seq.txt; tail -1 $O/seq.txt
rm -rf $O/stats
