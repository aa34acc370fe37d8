#!/bin/bash
# round-2 call T: PMC view of the conv kernels on predict shapes: clock (GRBM_GUI_ACTIVE), MFMA busy, wave stall split
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r2t; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
L=enc1c2,enc2c2,enc3c2,botc2,up1c2,enc0c2
export BENCH_B=138 BENCH_SCALE=2 BENCH_ONLY=$L
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $O/p1 -o p -- python $R/tools/bench_conv.py fwd 3 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $O/p2 -o p -- python $R/tools/bench_conv.py fwd 3 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_WAVES -d $O/p3 -o p -- python $R/tools/bench_conv.py fwd 3 > /dev/null 2>&1
for d in p1 p2 p3; do f=$(ls $O/$d/*/*.db $O/$d/*.db 2>/dev/null | head -1); echo "-- $d"; python $R/tools/rocpd_pmc.py $f 2>&1 | cut -c1-330; done
