R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6l; mkdir -p $O; cd $R
for p in 1 2 3 4 5 6 7 8; do
  echo "run $p $(timeout 300 python -m pytest tests/test_gpu_pipeline.py -q -s -k delivers 2>&1 | grep -E 'bare step' | cut -c1-150)"
done
echo "whole file: $(timeout 300 python -m pytest tests/test_gpu_pipeline.py -q -s 2>&1 | grep -E 'bare step|passed|failed' | cut -c1-150)"
