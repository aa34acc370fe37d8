#!/bin/bash
# round-2 call E: conv_pipe v2 (hand-placed stream): parity, stamps, per-layer time, bench A/B
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r2e; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_unet.py -q -x 2>&1 | tail -3
MPU_PIPE_DEBUG=32 BENCH_ONLY=enc3c2,botc2 timeout 200 python tools/bench_conv.py fwd 1 2> $O/stamps.txt > /dev/null; grep stamps $O/stamps.txt | head -8
for P in 1 0; do MPU_CONV_PIPE=$P timeout 200 python tools/bench_conv.py fwd 20 > $O/conv_fwd_pipe$P.txt 2>&1; done
paste $O/conv_fwd_pipe1.txt $O/conv_fwd_pipe0.txt | awk -F'\t' '{print substr($1,1,66), "|", substr($2,40,26)}' | grep -E "enc3|bot|up0|dg_|total"
cd /tmp
BENCH_ONLY=enc3c1,enc3c2,botc2,up0c2,dg_botc1 rocprofv3 --kernel-trace --stats -d $O/v2 -o t -- python $R/tools/bench_conv.py fwd 20 > $O/v2.log 2>&1
cd $R
for P in 1 0; do MPU_CONV_PIPE=$P timeout 300 python bench.py --steps 30 --warmup 5 --no-predict --no-cpu-baseline > $O/bench_pipe$P.json 2> $O/bench$P.err; done
python - <<'PY'
import json
for p in (1, 0):
    try:
        d = json.load(open("gpurun_out/r2e/bench_pipe%d.json" % p))
        print("pipe", p, d["ms_per_step"], "ms/step median", d.get("ms_per_step_median"), "conv frac", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "wgrad", d["wgrad"]["frac"], d.get("guard"), d.get("measured_peaks"))
    except Exception as e:
        print("pipe", p, "failed", e)
PY
tail -3 $O/bench1.err
