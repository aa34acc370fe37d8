# R6ae: conv_glds epilogue with the BatchNorm sums (accumulator mode): tests (bf16 + bf16x3 + f32), x3 and bf16 step times, bf16 sequence
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6ae; mkdir -p $O; cd $R
timeout 2700 python -m pytest tests/test_gpu_replay.py tests/test_gpu_unet.py tests/test_gpu_baseline_shapes.py tests/test_gpu_conv.py -q -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python /tmp/x3step.py time 2>/dev/null || true
B="python $R/bench.py --no-predict --no-cpu-baseline --no-e2e --no-peaks --no-kernel-events --no-graph"
for i in 1 2; do $B 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["ms_per_step_median"])'; done
$B --dtype bf16x3 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("x3", d["ms_per_step"], d["ms_per_step_median"])'
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- $B --steps 24 --warmup 3 > /dev/null 2>&1
S=$(find $O/stats -name "*.db" | head -1)
python $R/tools/rocpd_sequence.py $S > $O/train_step_sequence.txt 2>&1; sed -n 44,54p $O/train_step_sequence.txt; tail -1 $O/train_step_sequence.txt
rm -rf $O/stats
