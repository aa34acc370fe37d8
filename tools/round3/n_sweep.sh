#!/bin/bash
# round 3 call N: grouped launches: workgroups per wgrad_taps job x K-split target of the wgrad_glds jobs
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3n; mkdir -p $O
cd $R
for cfg in "128 512" "64 512" "128 256" "128 128" "64 256" "128 512"; do
  set -- $cfg
  MPU_WGRAD_TAPS_WGS=$1 MPU_WGRAD_SPLIT_TARGET=$2 timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks > $O/bench_$1_$2.log 2>&1
  tail -1 $O/bench_$1_$2.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('taps_wgs=$1 split_target=$2', d['ms_per_step'], d['ms_per_step_median'], d['roofline']['kernel_ms_per_step'], d['wgrad']['kernel_ms_per_step'], d['wgrad']['frac'])"
done
