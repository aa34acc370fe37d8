#!/bin/bash
# round 3 call Q: concat weight gradients as one job per source (cf=2 network); full parity on the touched paths; cf=2 bench
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3q; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_unet.py tests/test_gpu_replay.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 1500 python -m pytest tests/test_gpu_baseline_shapes.py -x -q -s -k "first_layer or cfg1 or cf2" > $O/pytest2.log 2>&1; tail -3 $O/pytest2.log; grep "cf=2" $O/pytest2.log | cut -c1-300
timeout 300 python bench.py --cf 2 --batch 8 --no-predict --no-cpu-baseline --no-peaks > $O/bench_cf2.log 2>&1
tail -1 $O/bench_cf2.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cf2', d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['wgrad']['kernel_ms_per_step'], d['wgrad']['frac'], d['schedules'])"
timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks > $O/bench.log 2>&1
tail -1 $O/bench.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg1', d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['wgrad']['kernel_ms_per_step'])"
timeout 1500 python -m pytest tests/test_gpu_bench_multi.py tests/test_gpu_distributed.py -x -q > $O/pytest3.log 2>&1; tail -3 $O/pytest3.log
