#!/bin/bash
# R5w: SQ counters of the level-3 3x3 layers (16 x 16 maps, B = 16) on conv_deepk against conv_pipe + splitk_finish (MPU_CONV_DEEPK=0):
# matrix-pipe busy share, wait share, LDS instructions / bank conflicts (tools/bench_conv.py fwd, back to back)
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r5w; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
L=enc3c1,enc3c2,up0c3
C2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
C3="SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_VALU"
for v in 1 0; do
  MPU_CONV_DEEPK=$v BENCH_ONLY=$L timeout 300 rocprofv3 --pmc $C2 -d $O/p2_$v -o p -- python $R/tools/bench_conv.py fwd 3 > /dev/null 2>&1
  MPU_CONV_DEEPK=$v BENCH_ONLY=$L timeout 300 rocprofv3 --pmc $C3 -d $O/p3_$v -o p -- python $R/tools/bench_conv.py fwd 3 > /dev/null 2>&1
done
{
echo "# SQ counters of the level-3 layers enc3c1 (256->512), enc3c2 (512->512), up0c3 (512->512) at B = 16, 16 x 16 maps (gpurun R5w): tools/bench_conv.py fwd"
echo "# deepk=1: conv_deepk_kernel (one launch per layer); deepk=0: conv_pipe_kernel + splitk_finish_kernel. Per-dispatch averages."
echo "# SQ_* cycle counters in quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES (cycles = 32 x #MFMA 32x32x16)."
echo "# MFMA busy share of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES / waves per SIMD); both kernels run 2 waves per SIMD."
for v in 1 0; do for d in p2 p3; do f=$(find $O/${d}_$v -name "*.db" | head -1); echo "-- deepk=$v $d"; python $R/tools/rocpd_pmc.py $f all 2>&1 | grep -A1 -E "conv_deepk|conv_pipe|splitk_finish" | grep -v "^--" | cut -c1-400; done; done
} > $O/conv_deepk_sq_counters.txt
rm -rf $O/p2_* $O/p3_*
cat $O/conv_deepk_sq_counters.txt | cut -c1-330
