# R6bd: workgroup counts of the three head_bn_* passes (one pass of 256 pixels per workgroup at configs[1]: all prologue / epilogue)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6bd; mkdir -p $O; cd $R
B="python $R/bench.py --no-predict --no-cpu-baseline --no-e2e --no-peaks --no-kernel-events --no-graph"
cd /tmp && export TMPDIR=/tmp
for X in 0 51205120512 25602560256 76807680768 102402560512; do
  MPU_HEAD_BN_BLOCKS=$X rocprofv3 --kernel-trace --stats -d $O/stats$X -o s -- $B --steps 24 --warmup 3 > /dev/null 2>&1
  S=$(find $O/stats$X -name "*.db" | head -1)
  python $R/tools/rocpd_sequence.py $S > $O/seq_$X.txt 2>&1
  echo "caps $X: $(grep -E 'head_bn|head_bwd' $O/seq_$X.txt | awk '{printf "%s(g%s) ", $NF, $(NF-1)}')"
  rm -rf $O/stats$X
done
