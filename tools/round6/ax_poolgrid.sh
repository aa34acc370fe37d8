# R6ax: workgroup cap of the two pool-backward recompute passes (147 / 160 registers: three workgroups per CU resident, so 1024 is 1.33 rounds)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6ax; mkdir -p $O; cd $R
B="python $R/bench.py --no-predict --no-cpu-baseline --no-e2e --no-peaks --no-kernel-events --no-graph"
cd /tmp && export TMPDIR=/tmp
for X in 1024 768 512 2048; do
  MPU_POOL_BWD_BLOCKS=$X rocprofv3 --kernel-trace --stats -d $O/stats$X -o s -- $B --steps 24 --warmup 3 > /dev/null 2>&1
  S=$(find $O/stats$X -name "*.db" | head -1)
  python $R/tools/rocpd_sequence.py $S > $O/seq_$X.txt 2>&1
  echo "cap $X: $(grep -E 'maxpool_bwd' $O/seq_$X.txt | awk '{printf "%s ", $NF}') | head: $(grep -E 'head_bn|head_bwd' $O/seq_$X.txt | awk '{printf "%s ", $NF}')"
  rm -rf $O/stats$X
done
cd $R
for i in 1 2; do for X in 1024 512; do
  MPU_POOL_BWD_BLOCKS=$X $B 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("cap='$X'", d["ms_per_step"], d["ms_per_step_median"], d["ms_per_step_min"])'
done; done
