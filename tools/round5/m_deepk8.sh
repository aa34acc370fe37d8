# R5m: conv_deepk on the 8 x 8 maps (two-way split over workgroups): parity, replay, step A/B against the 16-pixel-only build
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5m; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_replay.py tests/test_gpu_baseline_shapes.py -x -q -m gpu -k "k_split or deep_level or more_deep or replay or every or cfg1" > $O/pytest_full.log 2>&1; tail -6 $O/pytest_full.log
if grep -q "failed\|error" $O/pytest_full.log; then echo "PARITY FAILED"; grep -B40 "short test summary" $O/pytest_full.log | head -90; exit 0; fi
python tools/round5/deep_layers.py 2>&1 | grep "us per launch"
B="python bench.py --no-predict --no-cpu-baseline --no-peaks --no-e2e --steps 40 --warmup 10"
for rep in 1 2; do
  for v in 1 0; do
    MPU_CONV_DEEPK=$v timeout 300 $B > $O/bench_k${v}_$rep.json 2>/dev/null
    python - <<PY
import json
d=json.load(open("$O/bench_k${v}_$rep.json"))
print("deepk=$v rep $rep: ms_per_step", d["ms_per_step"], "median", d["ms_per_step_median"], "conv", d["roofline"]["kernel_ms_per_step"], "frac", d["roofline"]["frac"], {k:v for k,v in d["schedules"].items() if k.startswith("conv")})
PY
  done
done
