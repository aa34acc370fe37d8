# R5f: conv_deep with its epilogue knocked out (MPU_PIPE_DEBUG 1 = no partial stores, 2 = no epilogue): where do the ~13 us go
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for dbg in 0 1 2; do
  MPU_PIPE_DEBUG=$dbg rocprofv3 --kernel-trace --stats -d $O/t$dbg -o t -- python $R/tools/round5/deep_layers.py > $O/l$dbg.log 2>&1
  DB=$(find $O/t$dbg -name "*.db" | head -1)
  echo "== conv_deep MPU_PIPE_DEBUG=$dbg"; python $R/tools/rocpd_stats.py $DB 6 | grep -E "conv_deep|splitk|conv_pipe"
done
for dbg in 0 1; do
  MPU_CONV_DEEP=0 MPU_PIPE_DEBUG=$dbg rocprofv3 --kernel-trace --stats -d $O/p$dbg -o t -- python $R/tools/round5/deep_layers.py > $O/lp$dbg.log 2>&1
  DB=$(find $O/p$dbg -name "*.db" | head -1)
  echo "== conv_pipe MPU_PIPE_DEBUG=$dbg"; python $R/tools/rocpd_stats.py $DB 6 | grep -E "conv_deep|splitk|conv_pipe"
done
rm -rf $O/t0 $O/t1 $O/t2 $O/p0 $O/p1
