#!/bin/bash
# round-2 call Y: how much of conv_halo is the exposed patch reload at chunk boundaries? (ablation: skip it; results wrong)
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r2y; mkdir -p $O
cd $R
export TMPDIR=/tmp
L=enc1c2,enc2c2,enc3c2,up0c2,up1c2,up2c2
for P in 0 64; do MPU_HALO_DEBUG=$P BENCH_B=138 BENCH_SCALE=2 BENCH_ONLY=$L timeout 300 python tools/bench_conv.py fwd 5 2>/dev/null > $O/conv_p$P.txt; done
echo "-- predict batch shapes: normal | no patch reload"; paste $O/conv_p0.txt $O/conv_p64.txt | awk -F'\t' '{print substr($1,1,62), "|", substr($2,38,26)}'
for P in 0 64; do MPU_HALO_DEBUG=$P BENCH_ONLY=$L timeout 300 python tools/bench_conv.py fwd 20 2>/dev/null > $O/conv_t$P.txt; done
echo "-- train shapes: normal | no patch reload"; paste $O/conv_t0.txt $O/conv_t64.txt | awk -F'\t' '{print substr($1,1,62), "|", substr($2,38,26)}'
