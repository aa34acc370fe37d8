# R5t: fused back-mapping, lane layout 3 (gather instruction = compact 4x4x4 cube; MPU_FUSE_CUBE=1) against layout 2 (four z positions
# 4 apart): kernel time at 256^3 x 6 views with equality against the exact search, and the geometry parity tests under the new layout
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5t; mkdir -p $O
cd $R
for rep in 1 2; do
  for v in 0 1; do
    echo "== cube=$v"; MPU_FUSE_CUBE=$v REPS=10 python tools/bench_geometry.py 2>&1 | grep -E "fast=1|fast == exact|fuse"
  done
done | tee $O/cube_ab.txt
MPU_FUSE_CUBE=1 python -m pytest tests/test_gpu_geometry.py -q -x 2>&1 | tail -3
MPU_FUSE_CUBE=1 python -m pytest tests/test_gpu_baseline_shapes.py -q -x -k "predict or fuse or 256" 2>&1 | tail -3
