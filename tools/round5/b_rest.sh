# R5b: the tests call A did not reach (its -x stopped at a helper bug), the batched-candidate sampler, train_e2e again
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_fusion_train.py tests/test_gpu_cli.py tests/test_gpu_augment.py \
    tests/test_gpu_conv.py::test_halo16_subprocess "tests/test_gpu_conv.py::test_halo16_cases" \
    tests/test_gpu_unet.py -q -m gpu 2>&1 | tail -40 > $O/pytest_a.log
tail -30 $O/pytest_a.log
timeout 600 python bench.py --no-predict --no-cpu-baseline --no-peaks --steps 20 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5b/bench.json"))
print(d["ms_per_step"], json.dumps(d["train_e2e"]), json.dumps(d["f32_mode"]))
PY
tail -3 $O/bench.err
