#!/bin/bash
# R4t: BatchNorm finalize kernels fused into the apply kernels: parity + same-box step A/B
R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out/R4t; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_unet.py tests/test_gpu_replay.py tests/test_gpu_baseline_shapes.py -x -q -m gpu -s -k "not predict_fuse and not cfg4 and not six_view" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/summary.txt; tail -4 $O/pytest.log; grep -n "max rel diff" $O/pytest.log
for f in 1 0 1 0; do
  MPU_FUSED_BN_FINALIZE=$f timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks --steps 100 --warmup 20 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused=$f', d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min'])" | tee -a $O/step_ab.txt
done
