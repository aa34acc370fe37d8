# R5a: first GPU call of round 5 -- the changed / new GPU tests (exact integer parity, 128^3 oracle chain, fusion DP, CLI incl.
# --num_GPUs 2 train_fusion, halo16p after the removal of conv_halo16 / UPQ), then the default bench line (new legs: train_e2e,
# f32_mode, float4 copy probes, cpu_baseline_predict at 128^3).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5a; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_fusion_train.py tests/test_gpu_cli.py \
    tests/test_gpu_conv.py::test_halo16_subprocess "tests/test_gpu_conv.py::test_halo16_cases" \
    tests/test_gpu_unet.py -k "not graphed_train_step_matches_eager or True" -x -q -m gpu 2>&1 | tail -25 > $O/pytest_a.log
tail -5 $O/pytest_a.log
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py -x -q -m gpu -k "predict_fuse" -s 2>&1 | tail -12 > $O/pytest_b.log
tail -6 $O/pytest_b.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 6000 $O/bench.json; tail -5 $O/bench.err
