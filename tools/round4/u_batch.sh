#!/bin/bash
# R4u: round-aware auto batch of predict: parity tests that depend on the batching + predict bench
R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out/R4u; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_baseline_shapes.py tests/test_gpu_unet.py tests/test_gpu_distributed.py tests/test_gpu_cli.py tests/test_gpu_geometry.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/summary.txt; tail -3 $O/pytest.log
for i in 1 2; do timeout 300 python bench.py --predict-only 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['predict_fuse']; print(d['seconds'], 'unet_ms', d['unet_ms'], d['unet_frac_of_mfma_peak'], 'clk', d.get('shader_clock_mhz_during_predict'), d['label_histogram'])" | tee -a $O/predict.txt; done
timeout 900 python bench.py --config 4 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['predict_fuse']; print('cfg4', d['seconds'], 'unet_ms', d['unet_ms'], d['unet_frac_of_mfma_peak'])" | tee -a $O/predict.txt
