#!/bin/bash
# round 3 call E: deeper request rings (wgrad_taps DIST 3, halo8 NWS 4) A/B + in-loop stamps
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3e; mkdir -p $O
cd $R
for d in 0 1; do
  MPU_WGRAD_TAPS_DEEP=$d MPU_STAMPS=1 timeout 300 python tools/stamps.py wgrad enc0c2,enc2c2,up2c2 > $O/stamps_wgrad_deep$d.txt 2>&1
  echo "== wgrad deep=$d"; grep -v amdgpu $O/stamps_wgrad_deep$d.txt
done
for n in 3 4; do
  MPU_HALO8_NWS=$n MPU_STAMPS=1 timeout 300 python tools/stamps.py fwd enc1c2,enc2c2,up2c2 > $O/stamps_fwd_nws$n.txt 2>&1
  echo "== halo8 nws=$n"; grep -v amdgpu $O/stamps_fwd_nws$n.txt
done
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for cfg in "0 3" "1 4"; do
  set -- $cfg
  MPU_WGRAD_TAPS_DEEP=$1 MPU_HALO8_NWS=$2 timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks > $O/bench_$1_$2.log 2>&1
  tail -1 $O/bench_$1_$2.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('deep=$1 nws=$2', d['ms_per_step'], d['ms_per_step_median'], d['roofline']['kernel_ms_per_step'], d['wgrad']['kernel_ms_per_step'])"
done
