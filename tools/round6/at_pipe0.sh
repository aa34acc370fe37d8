# R6at: per-layer times of the ten conv_pipe layers on the conv_glds schedules instead (MPU_CONV_PIPE=0), for the up-conv data gradients
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6at; mkdir -p $O; cd $R
B="python $R/bench.py --no-predict --no-cpu-baseline --no-e2e --no-peaks --no-kernel-events --no-graph"
cd /tmp && export TMPDIR=/tmp
for X in 1 0; do
  MPU_CONV_PIPE=$X rocprofv3 --kernel-trace --stats -d $O/stats$X -o s -- $B --steps 24 --warmup 3 > /dev/null 2>&1
  S=$(find $O/stats$X -name "*.db" | head -1)
  python $R/tools/rocpd_sequence.py $S > $O/seq_pipe$X.txt 2>&1; tail -1 $O/seq_pipe$X.txt
  rm -rf $O/stats$X
done
