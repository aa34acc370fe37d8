#!/bin/bash
# round 3 call X: train step A/B of conv_halo8 SCHED 0 vs SCHED 1 with s_setprio 1 in the load phase (three alternations)
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3x; mkdir -p $O
cd $R
for cfg in "0 0" "1 2" "0 0" "1 2" "0 0" "1 2" "1 3"; do
  set -- $cfg
  MPU_HALO8_SCHED=$1 MPU_HALO8_PRIO=$2 timeout 300 python bench.py --no-predict --no-cpu-baseline --no-peaks > $O/bench_$1_$2.log 2>&1
  tail -1 $O/bench_$1_$2.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sched=$1 prio=$2', d['ms_per_step'], d.get('ms_per_step_median'), d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])"
done
