# R6as: XCD-contiguous tile ranges in conv_halo / conv_halo8 / conv_ws (MPU_XCD_TILES): conv tests, A/B of the step (three alternations),
# predict A/B, kernel table + FETCH_SIZE of both settings
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6as; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_replay.py tests/test_gpu_unet.py -q -x -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="python $R/bench.py --no-predict --no-cpu-baseline --no-e2e --no-peaks --no-kernel-events --no-graph"
for i in 1 2 3; do for X in 0 1; do
  MPU_XCD_TILES=$X $B 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("xcd='$X'", d["ms_per_step"], d["ms_per_step_median"], d["ms_per_step_min"])'
done; done
for X in 0 1 0 1; do
  MPU_XCD_TILES=$X python $R/bench.py --predict-only 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); p=d.get("predict_fuse",d); print("predict xcd='$X'", p.get("seconds"), p.get("unet_ms"))'
done
cd /tmp && export TMPDIR=/tmp
for X in 0 1; do
  MPU_XCD_TILES=$X rocprofv3 --kernel-trace --stats -d $O/stats$X -o s -- $B --steps 24 --warmup 3 > /dev/null 2>&1
  S=$(find $O/stats$X -name "*.db" | head -1)
  python $R/tools/rocpd_sequence.py $S > $O/seq_xcd$X.txt 2>&1; tail -1 $O/seq_xcd$X.txt
  MPU_XCD_TILES=$X rocprofv3 --pmc FETCH_SIZE -d $O/fetch$X -o f -- $B --steps 3 --warmup 2 > /dev/null 2>&1
  MPU_XCD_TILES=$X rocprofv3 --pmc WRITE_SIZE -d $O/write$X -o w -- $B --steps 3 --warmup 2 > /dev/null 2>&1
  python $R/tools/rocpd_traffic.py $(find $O/fetch$X -name "*.db" | head -1) $(find $O/write$X -name "*.db" | head -1) $O/traffic_xcd$X.json > /dev/null
  rm -rf $O/stats$X $O/fetch$X $O/write$X
done
