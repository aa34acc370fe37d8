# R5c: conv_deep (whole-image halo-patch schedule of the deep levels): parity, then same-box A/B of the train step
# against conv_pipe (MPU_CONV_DEEP=0), and the split-K finish's statistics rows (256 / 512 / 1024)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5c; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "whole_image or deep_level" 2>&1 | tail -15 > $O/pytest_deep.log
tail -8 $O/pytest_deep.log
if grep -q "failed\|error" $O/pytest_deep.log; then echo "PARITY FAILED: skipping the A/B"; exit 0; fi
timeout 600 python -m pytest tests/test_gpu_replay.py tests/test_gpu_unet.py -x -q -m gpu -k "replay or cfg1 or every_conv or graphed or staggered" 2>&1 | tail -6 > $O/pytest_replay.log
tail -4 $O/pytest_replay.log
B="python bench.py --no-predict --no-cpu-baseline --no-peaks --no-e2e --steps 40 --warmup 10"
for rep in 1 2; do
  for v in 1 0; do
    MPU_CONV_DEEP=$v timeout 300 $B > $O/bench_deep${v}_$rep.json 2>/dev/null
    python - <<PY
import json
d=json.load(open("$O/bench_deep${v}_$rep.json"))
print("deep=$v rep $rep: ms_per_step", d["ms_per_step"], "median", d["ms_per_step_median"], "conv", d["roofline"]["kernel_ms_per_step"], "frac", d["roofline"]["frac"], d["schedules"].get("conv deep"), d["schedules"].get("conv pipe"))
PY
  done
done
for rows in 512 1024; do
  MPU_SPLITK_STATS_ROWS=$rows timeout 300 $B > $O/bench_rows$rows.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$O/bench_rows$rows.json"))
print("stats rows $rows: ms_per_step", d["ms_per_step"], "median", d["ms_per_step_median"], "conv", d["roofline"]["kernel_ms_per_step"])
PY
done
