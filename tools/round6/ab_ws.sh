# R6ab: conv_ws data gradient with the BatchNorm-backward sums in its epilogue (one colreduce launch less): sequence, step time
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6ab; mkdir -p $O; cd $R
B="python $R/bench.py --no-predict --no-cpu-baseline --no-e2e --no-peaks --no-kernel-events --no-graph"
for i in 1 2; do $B 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["ms_per_step_median"])'; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- $B --steps 24 --warmup 3 > /dev/null 2>&1
S=$(find $O/stats -name "*.db" | head -1)
python $R/tools/rocpd_sequence.py $S > $O/train_step_sequence.txt 2>&1; sed -n 42,56p $O/train_step_sequence.txt; tail -1 $O/train_step_sequence.txt
rm -rf $O/stats
