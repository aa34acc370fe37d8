# R5h: conv_deepk (K split over the waves, no partials): parity, per-layer times, stamps, step A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5h; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "k_split or whole_image or deep_level" 2>&1 | tail -8 > $O/pytest.log
tail -5 $O/pytest.log
if grep -q "failed\|error" $O/pytest.log; then echo "PARITY FAILED"; tail -40 $O/pytest.log; exit 0; fi
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  MPU_CONV_DEEPK=$v MPU_CONV_DEEP=0 rocprofv3 --kernel-trace --stats -d $O/t$v -o t -- python $R/tools/round5/deep_layers.py > $O/l$v.log 2>&1
  DB=$(find $O/t$v -name "*.db" | head -1)
  echo "== MPU_CONV_DEEPK=$v"; python $R/tools/rocpd_stats.py $DB 6 | grep -E "conv_deep|splitk|conv_pipe"; grep "us per launch" $O/l$v.log
done
rm -rf $O/t1 $O/t0
cd $R
MPU_STAMPS=1 MPU_CONV_DEEP=0 python tools/round5/deep_layers.py stamps 2>&1 | grep -v "^$" | head -60 | tee $O/stamps.txt
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
