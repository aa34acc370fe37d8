#!/bin/bash
# round 3 call AA: k-step skipping of tail chunks (96-channel layers) -- parity of the conv suite, cf=2 step, halo8 grid bound A/B
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3aa; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q > $O/pytest_conv.log 2>&1; tail -3 $O/pytest_conv.log
timeout 600 python -m pytest tests/test_gpu_baseline_shapes.py -x -q -k "cf2 or first_layer" > $O/pytest_cf2.log 2>&1; tail -3 $O/pytest_cf2.log
for mx in 640 400 640 400; do
  MPU_HALO8_MAX_WGS=$mx timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks --cf 2 --batch 16 > $O/bench_cf2_$mx.log 2>&1
  tail -1 $O/bench_cf2_$mx.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cf2 B16 halo8_max=$mx', d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'], d['wgrad']['kernel_ms_per_step'])"
done
timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks --cf 2 > $O/bench_cf2_b8.log 2>&1
tail -1 $O/bench_cf2_b8.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cf2 default batch', d['ms_per_step'], d['config']['slices_per_gpu'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])"
timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks > $O/bench_cf1.log 2>&1
tail -1 $O/bench_cf1.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cf1', d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])"
