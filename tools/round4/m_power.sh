#!/bin/bash
# round 4 call M: socket power + clock of the probes, the predict kernel and its knock-outs, conv_halo16, the train step
R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out/R4m; mkdir -p $O
export MPU_LIB_PATH=$R/multiplanarunet_amd/libab/knock.so
for ko in 0 2 8 10 30 62; do
  echo "# knockout=$ko" | tee -a $O/power_knock.txt
  MPU_HALO_KNOCKOUT=$ko timeout 200 python tools/round4/power_layers.py enc1c2 2>&1 | grep -v amdgpu.ids | tee -a $O/power_knock.txt
done
unset MPU_LIB_PATH
echo "# predict layers of other kinds (production build): level-2 256->256, wgrad-free" | tee -a $O/power_knock.txt
timeout 200 python tools/round4/power_layers.py enc2c2 2>&1 | grep -v amdgpu.ids | tee -a $O/power_knock.txt
