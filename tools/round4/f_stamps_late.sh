#!/bin/bash
# round 4 call F: conv_halo16 stamps of a LATER round of workgroups (is the first round representative?)
R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out/R4f; mkdir -p $O
for first in 0 2048 4096; do
  echo "MPU_STAMPS_FIRST=$first" | tee -a $O/stamps16.txt
  MPU_STAMPS_FIRST=$first MPU_STAMPS=1 timeout 300 python tools/round4/stamps16.py enc1c2 2>&1 | grep -v amdgpu.ids | head -9 | tee -a $O/stamps16.txt
done
