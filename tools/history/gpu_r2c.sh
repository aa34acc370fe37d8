#!/bin/bash
# round-2 call C: where does conv_pipe_kernel's time go? kernel-trace of three deep layers under the profiling switches
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r2c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # tag, env...
  tag=$1; shift
  env "$@" BENCH_ONLY=enc3c2,botc2,up0c2,dg_botc1,enc3c1 rocprofv3 --kernel-trace --stats -d $O/$tag -o t -- python $R/tools/bench_conv.py fwd 20 > $O/$tag.log 2>&1
  db=$(find $O/$tag -name "*.db" | head -1)
  echo "== $tag"; python $R/tools/rocpd_stats.py $db 8 | grep -E "conv_pipe|splitk|conv_glds" | awk '{printf "%-60s %6s %10s %8s\n", substr($1,1,60), $(NF-3), $(NF-2), $(NF-1)}'
}
run base MPU_PIPE_DEBUG=0
run nostore MPU_PIPE_DEBUG=1
run nomma MPU_PIPE_DEBUG=2
run nodma MPU_PIPE_DEBUG=4
run nomma_nostore MPU_PIPE_DEBUG=3
run nodma_nostore MPU_PIPE_DEBUG=5
run old MPU_CONV_PIPE=0
cd $R && timeout 600 python -m pytest tests/test_gpu_baseline_shapes.py -q -k "cfg1_bf16" -rA 2>&1 | grep -E "cfg1|passed|failed" | head -20
