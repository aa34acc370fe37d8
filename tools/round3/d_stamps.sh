#!/bin/bash
# round 3 call D: where the time of a one-wave conv / wgrad launch goes (s_memtime stamps)
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3d; mkdir -p $O
cd $R
MPU_STAMPS=1 timeout 300 python tools/stamps.py fwd enc1c1,enc1c2,enc2c2,up2c2,up2c1 > $O/stamps_fwd.txt 2>&1
MPU_STAMPS=1 timeout 300 python tools/stamps.py wgrad enc0c2,enc1c2,enc2c2,up2c2 > $O/stamps_wgrad.txt 2>&1
grep -v amdgpu $O/stamps_fwd.txt; grep -v amdgpu $O/stamps_wgrad.txt
timeout 900 python -m pytest tests/test_gpu_replay.py -x -q -s > $O/replay.log 2>&1; tail -8 $O/replay.log
