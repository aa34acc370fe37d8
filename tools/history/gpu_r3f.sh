#!/bin/bash
# round-2 call 3F: s_setprio(1) around the MFMA clusters of conv_halo (two co-resident workgroups per CU)
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r3f; mkdir -p $O; cd $R; export TMPDIR=/tmp
L=enc1c2,enc2c2,enc3c2,up0c2,up1c2,up2c2
for P in 0 128; do MPU_HALO_DEBUG=$P BENCH_B=138 BENCH_SCALE=2 BENCH_ONLY=$L timeout 300 python tools/bench_conv.py fwd 5 2>/dev/null > $O/p$P.txt; done
echo "-- predict shapes: normal | setprio"; paste $O/p0.txt $O/p128.txt | awk -F'\t' '{print substr($1,1,62), "|", substr($2,38,26)}'
for P in 0 128; do MPU_HALO_DEBUG=$P BENCH_ONLY=$L timeout 300 python tools/bench_conv.py fwd 20 2>/dev/null > $O/t$P.txt; done
echo "-- train shapes: normal | setprio"; paste $O/t0.txt $O/t128.txt | awk -F'\t' '{print substr($1,1,62), "|", substr($2,38,26)}'
