#!/bin/bash
# round 4 call H: shader clock under the knocked-out variants of conv_halo<128,8,2> and under conv_halo16: what costs the power budget?
R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out/R4h; mkdir -p $O
export MPU_LIB_PATH=$R/multiplanarunet_amd/libab/knock.so
for ko in 0 2 8 10 4 20 32 30 62; do
  echo -n "knockout=$ko  " | tee -a $O/clock.txt
  MPU_HALO_KNOCKOUT=$ko timeout 120 python tools/round4/clock_layers.py enc1c2 2>&1 | grep -v amdgpu.ids | tee -a $O/clock.txt
done
echo -n "knockout=0 zero data  " | tee -a $O/clock.txt
ZERO_DATA=1 MPU_HALO_KNOCKOUT=0 timeout 120 python tools/round4/clock_layers.py enc1c2 2>&1 | grep -v amdgpu.ids | tee -a $O/clock.txt
unset MPU_LIB_PATH
echo -n "halo16  " | tee -a $O/clock.txt
MPU_HALO16=1 timeout 120 python tools/round4/clock_layers.py enc1c2,enc2c2,up2c2 2>&1 | grep -v amdgpu.ids | tee -a $O/clock.txt
echo -n "halo16 zero data  " | tee -a $O/clock.txt
ZERO_DATA=1 MPU_HALO16=1 timeout 120 python tools/round4/clock_layers.py enc1c2 2>&1 | grep -v amdgpu.ids | tee -a $O/clock.txt
echo -n "4-wave  " | tee -a $O/clock.txt
timeout 120 python tools/round4/clock_layers.py enc1c2,enc2c2,up2c2 2>&1 | grep -v amdgpu.ids | tee -a $O/clock.txt
