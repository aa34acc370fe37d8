#!/bin/bash
# R4v: socket power + shader clock of the predict layers on conv_halo16p vs the round-3 schedules (MPU_HALO16P=0)
R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out/R4v; mkdir -p $O
for p in 1 0 1 0; do
  echo "# MPU_HALO16P=$p" | tee -a $O/power.txt
  MPU_HALO16P=$p timeout 200 python tools/round4/power_layers.py enc1c2,up2c2 2>&1 | grep -v amdgpu.ids | tee -a $O/power.txt
done
