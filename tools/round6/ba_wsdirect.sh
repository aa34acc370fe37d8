# R6ba: conv_ws weights straight into registers (MPU_WS_DIRECT): conv / replay / unet tests, step A/B (three alternations), per-launch
# times of both settings, predict A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6ba; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_replay.py tests/test_gpu_unet.py -q -x -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="python $R/bench.py --no-predict --no-cpu-baseline --no-e2e --no-peaks --no-kernel-events --no-graph"
for i in 1 2 3; do for X in 0 1; do
  MPU_WS_DIRECT=$X $B 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("direct='$X'", d["ms_per_step"], d["ms_per_step_median"], d["ms_per_step_min"])'
done; done
for X in 0 1 0 1; do
  MPU_WS_DIRECT=$X python $R/bench.py --predict-only 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); p=d.get("predict_fuse",d); print("predict direct='$X'", p.get("seconds"), p.get("unet_ms"))'
done
cd /tmp && export TMPDIR=/tmp
for X in 0 1; do
  MPU_WS_DIRECT=$X rocprofv3 --kernel-trace --stats -d $O/stats$X -o s -- $B --steps 24 --warmup 3 > /dev/null 2>&1
  S=$(find $O/stats$X -name "*.db" | head -1)
  python $R/tools/rocpd_sequence.py $S > $O/seq_$X.txt 2>&1
  echo "direct $X: conv_ws $(grep conv_ws $O/seq_$X.txt | awk '{printf "%s ", $NF}') | $(tail -1 $O/seq_$X.txt | cut -c1-40)"
  rm -rf $O/stats$X
done
