#!/bin/bash
# R4w: SQ counters of the predict-size layers on conv_halo16p vs the round-3 schedules (MPU_HALO16P=0)
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R4w; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
L=enc1c2,enc2c2,up2c2,up1c2
C2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
for p in 1 0; do
  MPU_HALO16P=$p BENCH_B=138 BENCH_SCALE=2 BENCH_ONLY=$L timeout 300 rocprofv3 --pmc $C2 -d $O/p$p -o p -- python $R/tools/bench_conv.py fwd 3 > /dev/null 2>&1
done
{
echo "# SQ counters of four predict-size layers (B = 138 planes; tools/bench_conv.py fwd), conv_halo16p (p1) vs the round-3 schedules (p0, MPU_HALO16P=0). gpurun R4w."
echo "# Per-dispatch averages; SQ_* cycle counters in quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES (cycles = 32 x #MFMA 32x32x16)."
echo "# MFMA busy share of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES / waves per SIMD)."
for d in p1 p0; do f=$(find $O/$d -name "*.db" | head -1); echo "-- $d"; python $R/tools/rocpd_pmc.py $f 2>&1 | cut -c1-400; done
} > $O/predict_layers_pmc.txt
rm -rf $O/p1 $O/p0
cat $O/predict_layers_pmc.txt | cut -c1-330
