# R6bc: the GPU suite with the last session's switches OFF (launch-order tiles, the five-launch head chain, the two-tensor pool backward):
# the fallback paths (also taken under a launch tap and by networks without a 64-channel last block) stay green
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6bc; mkdir -p $O; cd $R
MPU_XCD_TILES=0 MPU_HEAD_TRAIN_FUSED=0 MPU_POOL_BWD_RECOMPUTE=0 timeout 2700 python -m pytest tests -q -m gpu -x \
  -k "not training_head_without and not pool_backward_without" > $O/pytest_off.log 2>&1; tail -4 $O/pytest_off.log
