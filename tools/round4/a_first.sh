#!/bin/bash
# round 4 call A: parity of the wgrad_taps swizzle + CLI changes; step A/B against the round-3 kernels (libab/base.so);
# knock-out timing of conv_halo<128,8,2> on predict-size layers (MPU_HALO_KNOCKOUT)
R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out/R4a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_cli.py tests/test_gpu_replay.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for i in 1 2; do
  for lib in base new; do
    if [ $lib = base ]; then export MPU_LIB_PATH=$R/multiplanarunet_amd/libab/base.so; else unset MPU_LIB_PATH; fi
    timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks --steps 50 --warmup 10 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['ms_per_step_median'], 'conv', d['roofline']['kernel_ms_per_step'], 'wgrad', d['wgrad']['kernel_ms_per_step'], d['wgrad']['frac'])" | tee -a $O/step_ab.txt
  done
done
unset MPU_LIB_PATH
for ko in 0 1 32 2 8 10 4 16 20 64 0; do
  echo "knockout=$ko" | tee -a $O/knock.txt
  MPU_HALO_KNOCKOUT=$ko BENCH_B=138 BENCH_SCALE=2 BENCH_ONLY=enc1c2,up2c2,enc2c2 timeout 200 python tools/bench_conv.py fwd 10 2>&1 | grep -v "^total" | tee -a $O/knock.txt
done
