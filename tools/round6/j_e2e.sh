R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6j; mkdir -p $O; cd $R
for v in 1 0 1 0; do
  MPU_TAIL_OVERLAP=$v timeout 300 python bench.py --e2e-only --steps 60 > $O/e2e_$v.log 2>&1; echo "overlap=$v $(tail -1 $O/e2e_$v.log | cut -c1-700)"
done
