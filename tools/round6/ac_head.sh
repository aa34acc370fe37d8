# R6ac: head_backward with the last BatchNorm's backward sums (one more colreduce launch gone): tests, step time, sequence
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6ac; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests/test_gpu_replay.py tests/test_gpu_unet.py tests/test_gpu_baseline_shapes.py -q -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
B="python $R/bench.py --no-predict --no-cpu-baseline --no-e2e --no-peaks --no-kernel-events --no-graph"
for i in 1 2; do $B 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["ms_per_step_median"])'; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- $B --steps 24 --warmup 3 > /dev/null 2>&1
S=$(find $O/stats -name "*.db" | head -1)
python $R/tools/rocpd_sequence.py $S > $O/train_step_sequence.txt 2>&1; sed -n 40,52p $O/train_step_sequence.txt; tail -1 $O/train_step_sequence.txt
rm -rf $O/stats
