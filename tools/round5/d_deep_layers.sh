# R5d: where conv_deep spends its time: per-layer kernel durations (rocprofv3 kernel trace) of conv_deep vs conv_pipe on the six
# deep 3x3 forward layers of configs[1], and conv_deep's in-kernel stamps
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  MPU_CONV_DEEP=$v rocprofv3 --kernel-trace -d $O/trace$v -o t -- python $R/tools/round5/deep_layers.py > $O/layers$v.log 2>&1
  DB=$(find $O/trace$v -name "*.db" | head -1)
  echo "== MPU_CONV_DEEP=$v"; python $R/tools/round5/deep_trace.py $DB | tee $O/trace$v.txt
  grep "us per launch" $O/layers$v.log
done
MPU_STAMPS=1 python $R/tools/round5/deep_layers.py stamps 2>&1 | tee $O/stamps.txt | grep -v "^$"
rm -rf $O/trace1 $O/trace0
