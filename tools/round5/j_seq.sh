# R5j: per-launch sequence of the train step (rocprofv3 kernel trace, eager launches) with conv_deepk on / off
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-predict --no-cpu-baseline --no-graph --no-kernel-events --no-peaks --no-e2e"
for v in 1 0; do
  MPU_CONV_DEEPK=$v MPU_CONV_DEEP=0 rocprofv3 --kernel-trace --stats -d $O/s$v -o s -- $B --steps 10 --warmup 3 > /dev/null 2>&1
  S=$(find $O/s$v -name "*.db" | head -1)
  python $R/tools/rocpd_sequence.py $S > $O/seq$v.txt 2>&1
  echo "== deepk=$v"; grep -E "conv_deepk|conv_pipe|splitk_finish" $O/seq$v.txt | awk '{print}' | head -40; tail -1 $O/seq$v.txt
  rm -rf $O/s$v
done
