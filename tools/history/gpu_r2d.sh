#!/bin/bash
# round-2 call D: conv_pipe stamps + per-step floor switches; geometry fast paths (tests + predict bench)
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r2d; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_validation.py tests/test_gpu_distributed.py -q -rA 2>&1 | grep -E "PASSED|FAILED|passed|failed|Error" | tail -50
timeout 300 python bench.py --predict-only > $O/predict_fast.json 2> $O/predict_fast.err; cat $O/predict_fast.json | head -c 900; echo
MPU_GEOM_FAST=0 timeout 300 python bench.py --predict-only > $O/predict_exact.json 2>/dev/null; python -c "
import json
for t in ('fast','exact'):
    d=json.load(open('gpurun_out/r2d/predict_%s.json'%t))['predict_fuse']; print(t, d['sample_ms'], d['map_fuse_ms'], d['unet_ms'], d['seconds'])
"
MPU_PIPE_DEBUG=32 BENCH_ONLY=enc3c2,botc2,up0c2 timeout 200 python tools/bench_conv.py fwd 1 2> $O/stamps.txt > /dev/null; grep -c stamps $O/stamps.txt
cd /tmp
run() { tag=$1; shift
  env "$@" BENCH_ONLY=enc3c1,enc3c2,botc2,up0c2,dg_botc1 rocprofv3 --kernel-trace --stats -d $O/$tag -o t -- python $R/tools/bench_conv.py fwd 20 > $O/$tag.log 2>&1
}
run nolds_nostore MPU_PIPE_DEBUG=9
run nobar_nostore MPU_PIPE_DEBUG=17
run nolds_nodma_nostore MPU_PIPE_DEBUG=13
run nolds_nomma_nodma_nostore MPU_PIPE_DEBUG=15
run nomma_nolds_nostore MPU_PIPE_DEBUG=11
