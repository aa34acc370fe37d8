#!/bin/bash
# round-2 call F: conv_pipe prologue changes + experimental pipe-first dispatch for level 1
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r2f; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py -q -x -k "deep or forward_dgrad" 2>&1 | tail -2
MPU_PIPE_DEBUG=32 BENCH_ONLY=enc3c2 timeout 200 python tools/bench_conv.py fwd 1 2> $O/stamps.txt > /dev/null; grep stamps $O/stamps.txt | tail -4
for F in 0 1; do MPU_PIPE_FIRST=$F timeout 200 python tools/bench_conv.py fwd 20 > $O/conv_fwd_first$F.txt 2>&1; done
paste $O/conv_fwd_first0.txt $O/conv_fwd_first1.txt | awk -F'\t' '{print substr($1,1,66), "|", substr($2,40,26)}' | grep -vE "amdgpu"
timeout 200 python tools/bench_conv.py wgrad 20 > $O/conv_wgrad.txt 2>&1; grep -v amdgpu $O/conv_wgrad.txt
for F in 0 1; do MPU_PIPE_FIRST=$F timeout 300 python bench.py --steps 30 --warmup 5 --no-predict --no-cpu-baseline > $O/bench_first$F.json 2> $O/bench$F.err; done
python - <<'PY'
import json
for p in (0, 1):
    try:
        d = json.load(open("gpurun_out/r2f/bench_first%d.json" % p))
        print("pipe_first", p, d["ms_per_step"], "ms/step median", d.get("ms_per_step_median"), "conv frac", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "wgrad", d["wgrad"]["frac"], d["schedules"])
    except Exception as e:
        print("pipe_first", p, "failed", e)
PY
