#!/bin/bash
# round 3 call AF: wgrad_taps with the two groups one phase apart: parity (op-level + network), per-layer and step A/B
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3af; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -k "wgrad" > $O/pytest_wgrad.log 2>&1; tail -2 $O/pytest_wgrad.log
timeout 900 python -m pytest tests/test_gpu_replay.py tests/test_gpu_unet.py -x -q > $O/pytest_net.log 2>&1; tail -2 $O/pytest_net.log
for s in 1 0 1 0; do
  echo "== per layer wgrad stag=$s"
  MPU_WGRAD_TAPS_STAG=$s BENCH_ONLY=enc0c2,enc1c2,enc2c2,up2c2,up3c2 timeout 300 python tools/bench_conv.py wgrad 20 2>&1 | grep -v amdgpu
done
for s in 1 0 1 0 1 0; do
  MPU_WGRAD_TAPS_STAG=$s timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('stag=$s', d['ms_per_step'], d['wgrad']['kernel_ms_per_step'], d['wgrad']['frac'])"
done
