# R6c: timeline of the step's tail with the overlap on / off (graph replay, rocprofv3 kernel trace)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-predict --no-cpu-baseline --no-e2e --no-peaks --no-kernel-events --steps 10 --warmup 3"
for v in 1 0; do
  MPU_TAIL_OVERLAP=$v rocprofv3 --kernel-trace -d $O/t$v -o t -- $B > $O/bench_$v.log 2>&1
  DB=$(find $O/t$v -name "*.db" | head -1)
  python $R/tools/round6/tail_trace.py $DB > $O/tail_$v.txt 2>&1; echo "== overlap $v"; cat $O/tail_$v.txt
  rm -rf $O/t$v
done
