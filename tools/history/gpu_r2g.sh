#!/bin/bash
# round-2 call G: full GPU suite after the BN-bwd epilogue port + compact map_fuse; full bench
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r2g; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|ERROR" $O/pytest.log | tail -8
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2g/bench.json"))
print(d["ms_per_step"], d["ms_per_step_median"], d["value"], d["roofline"]["frac"], d["wgrad"]["frac"], d["guard"])
p = d["predict_fuse"]; print({k: p[k] for k in ("value", "seconds", "sample_ms", "unet_ms", "map_fuse_ms", "map_fuse_frac_of_hbm_peak", "unet_tflops_algorithmic")}, p.get("cpu_baseline"))
print(d.get("cpu_baseline")); print(d.get("measured_peaks"))
PY
MPU_FUSED_BN_BWD_CONV=0 timeout 300 python bench.py --steps 30 --warmup 5 --no-predict --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bn_bwd_conv=0', d['ms_per_step'], d['ms_per_step_median'])"
