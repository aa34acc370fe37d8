# R5l: fused back-mapping with non-temporal gathers / label stores and the view loop unrolled (variant libraries under libab/,
# built from csrc/geometry.hip with -DMPU_FUSE_NT / -DMPU_FUSE_UNROLL; MPU_LIB_PATH selects): same-box A/B, equality with the
# exact search checked by tools/bench_geometry.py in every run
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5l; mkdir -p $O
cd $R
for rep in 1 2; do
  echo "== base"; REPS=8 python tools/bench_geometry.py 2>&1 | grep -E "fast=1|fast == exact"
  for v in nt u2 ntu2 ntu3; do
    echo "== $v"; MPU_LIB_PATH=$R/libab/geom_$v.so REPS=8 python tools/bench_geometry.py 2>&1 | grep -E "fast=1|fast == exact"
  done
done | tee $O/fuse_ab.txt
