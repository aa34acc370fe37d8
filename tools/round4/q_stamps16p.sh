#!/bin/bash
# R4q: in-kernel stamps of conv_halo16p
R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out/R4q; mkdir -p $O
MPU_STAMPS=1 MPU_HALO16=1 timeout 300 python tools/round4/stamps16p.py enc1c2,up2c2,enc2c2 2>&1 | grep -v amdgpu.ids | tee $O/stamps.txt
