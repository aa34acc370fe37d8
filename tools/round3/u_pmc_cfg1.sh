#!/bin/bash
# round 3 call U: SQ counters of the configs[1]-size conv / wgrad kernels (halo8, pipe, ws; grouped wgrads through the train step)
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3u; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
L=enc0c2,enc1c2,enc2c2,enc3c2,botc2,up2c2
C2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
C3="SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_VALU"
BENCH_ONLY=$L timeout 300 rocprofv3 --pmc $C2 -d $O/p2 -o p -- python $R/tools/bench_conv.py fwd 3 > /dev/null 2>&1
BENCH_ONLY=$L timeout 300 rocprofv3 --pmc $C3 -d $O/p3 -o p -- python $R/tools/bench_conv.py fwd 3 > /dev/null 2>&1
B="python $R/bench.py --no-predict --no-cpu-baseline --no-graph --no-kernel-events --no-peaks --steps 3 --warmup 2"
timeout 300 rocprofv3 --pmc $C2 -d $O/t2 -o p -- $B > /dev/null 2>&1
{
echo "# SQ counters at configs[1] sizes (gpurun R3u). p2/p3: tools/bench_conv.py fwd on $L (B=16); t2: one eager train step."
echo "# Per-dispatch averages; SQ_* cycle counters in quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES (cycles = 32 x #MFMA 32x32x16, 16 x #MFMA 16x16x32)."
echo "# MFMA busy share of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES / waves per SIMD)."
for d in p2 p3 t2; do f=$(find $O/$d -name "*.db" | head -1); echo "-- $d"; python $R/tools/rocpd_pmc.py $f 2>&1 | cut -c1-400; done
} > $O/conv_pmc_cfg1_shapes.txt
rm -rf $O/p2 $O/p3 $O/t2
head -60 $O/conv_pmc_cfg1_shapes.txt | cut -c1-330
