#!/bin/bash
# R4r: knock-outs of the conv_halo16p epilogue (compile-time variants in libab/): where do its ~4.3 k cycles go?
R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out/R4r; mkdir -p $O
for ko in 0 1 2 4 8 15; do
  lib=$R/multiplanarunet_amd/lib/libmpunet_hip.so
  [ $ko != 0 ] && lib=$R/multiplanarunet_amd/libab/h16p_ko$ko.so
  echo "== knock-out $ko" | tee -a $O/ko.txt
  MPU_LIB_PATH=$lib MPU_STAMPS=1 MPU_HALO16=1 timeout 300 python tools/round4/stamps16p.py up2c2 2>&1 | grep -v amdgpu.ids | grep -v "tile [1-4]:" | tee -a $O/ko.txt
done
