# R6au: up-conv data gradient with two K splits on unsplit conv_glds tiles: conv tests, step time, sequence rows; the projection
# coefficients of the bf16 step test
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6au; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_replay.py tests/test_gpu_unet.py -q -x -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py -q -x -m gpu -s -k "cfg1_bf16_train_step" > $O/pytest_bf16.log 2>&1; grep -E "projection|tightest|passed|failed" $O/pytest_bf16.log
B="python $R/bench.py --no-predict --no-cpu-baseline --no-e2e --no-peaks --no-kernel-events --no-graph"
for i in 1 2 3; do $B 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["ms_per_step_median"], d["ms_per_step_min"])'; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- $B --steps 24 --warmup 3 > /dev/null 2>&1
S=$(find $O/stats -name "*.db" | head -1)
python $R/tools/rocpd_sequence.py $S > $O/seq.txt 2>&1; sed -n 54,60p $O/seq.txt; tail -1 $O/seq.txt
rm -rf $O/stats
