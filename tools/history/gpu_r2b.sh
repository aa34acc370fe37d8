#!/bin/bash
# round-2 call B: full GPU suite, conv_pipe A/B on the layer table, bench with / without the new schedule
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=10 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -E "passed|failed" $O/pytest.log | tail -3
for P in 1 0; do
  MPU_CONV_PIPE=$P timeout 200 python tools/bench_conv.py fwd 20 > $O/conv_fwd_pipe$P.txt 2>&1
done
paste $O/conv_fwd_pipe1.txt $O/conv_fwd_pipe0.txt | awk -F'\t' '{print substr($1,1,66), "|", substr($2,40,26)}'
MPU_CONV_PIPE=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-predict --no-cpu-baseline > $O/bench_pipe1.json 2> $O/bench1.err; echo "bench rc=$?"
MPU_CONV_PIPE=0 timeout 300 python bench.py --steps 30 --warmup 5 --no-predict --no-cpu-baseline > $O/bench_pipe0.json 2> $O/bench0.err
python - <<'PY'
import json
for p in (1, 0):
    try:
        d = json.load(open("gpurun_out/r2b/bench_pipe%d.json" % p))
        print("pipe", p, d["ms_per_step"], "ms/step", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["wgrad"]["frac"])
    except Exception as e:
        print("pipe", p, "failed", e)
PY
