# R6bb: XCD-contiguous (pixel-tile sequence, n-tile) ranges in the persistent predict kernel conv_halo16p: equality test, predict A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6bb; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_conv.py -q -x -m gpu -k "halo16 or persistent or predict" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for X in 0 1 0 1 0 1; do
  MPU_XCD_TILES=$X python $R/bench.py --predict-only 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); p=d.get("predict_fuse",d); print("predict xcd='$X'", p.get("seconds"), p.get("unet_ms"))'
done
for X in 0 1; do
  MPU_XCD_TILES=$X python $R/bench.py --config 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("cfg4 xcd='$X'", d["ms_per_step"])'
done
