# R5i: conv_deepk with the fused BatchNorm statistics epilogues and the earlier fragment reads: parity (layer cases, replay of
# every launch of the configs[1] step, train-step tests), layer times, step A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5i; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_replay.py tests/test_gpu_unet.py tests/test_gpu_baseline_shapes.py -x -q -m gpu -k "k_split or deep_level or replay or every or cfg1 or graphed or staggered or f32_train or bf16" > $O/pytest_full.log 2>&1; tail -8 $O/pytest_full.log > $O/pytest.log
tail -5 $O/pytest.log
if grep -q "failed\|error" $O/pytest.log; then echo "PARITY FAILED"; grep -B30 "short test summary" $O/pytest_full.log | head -80; exit 0; fi
MPU_CONV_DEEP=0 python tools/round5/deep_layers.py 2>&1 | grep "us per launch"
B="python bench.py --no-predict --no-cpu-baseline --no-peaks --no-e2e --steps 40 --warmup 10"
for rep in 1 2; do
  for v in 1 0; do
    MPU_CONV_DEEPK=$v MPU_CONV_DEEP=0 timeout 300 $B > $O/bench_k${v}_$rep.json 2>/dev/null
    python - <<PY
import json
d=json.load(open("$O/bench_k${v}_$rep.json"))
print("deepk=$v rep $rep: ms_per_step", d["ms_per_step"], "median", d["ms_per_step_median"], "conv", d["roofline"]["kernel_ms_per_step"], "frac", d["roofline"]["frac"], {k:v for k,v in d["schedules"].items() if k.startswith("conv")})
PY
  done
done
