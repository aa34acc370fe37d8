#!/bin/bash
# round 4 call B: knock-out timing of conv_halo<128,8,2> (compile-time masks, libab/knock.so) on predict-size layers;
# SQ LDS counters of the train step after the wgrad_taps swizzle change
R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out/R4b; mkdir -p $O
export MPU_LIB_PATH=$R/multiplanarunet_amd/libab/knock.so
for ko in 0 1 32 33 2 8 10 4 16 20 64 30 62 0; do
  echo "knockout=$ko" | tee -a $O/knock.txt
  MPU_HALO_KNOCKOUT=$ko BENCH_B=138 BENCH_SCALE=2 BENCH_ONLY=enc1c2,up2c2 timeout 200 python tools/bench_conv.py fwd 10 2>&1 | grep -v "^total\|amdgpu.ids" | tee -a $O/knock.txt
done
unset MPU_LIB_PATH
cd /tmp; export TMPDIR=/tmp
C2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
C3="SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_VALU"
B="python $R/bench.py --no-predict --no-cpu-baseline --no-graph --no-kernel-events --no-peaks --steps 3 --warmup 2"
timeout 300 rocprofv3 --pmc $C2 -d $O/t2 -o p -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc $C3 -d $O/t3 -o p -- $B > /dev/null 2>&1
for d in t2 t3; do f=$(find $O/$d -name "*.db" | head -1); echo "-- $d"; python $R/tools/rocpd_pmc.py $f 2>&1 | cut -c1-400; done > $O/step_pmc.txt
rm -rf $O/t2 $O/t3
grep -A1 "wgrad_taps_group\|wgrad_glds_group\|conv_halo8\|conv_pipe" $O/step_pmc.txt | cut -c1-300 | head -40
