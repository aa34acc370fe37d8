"""
bench.py under torch.distributed.run with N > 1: two ranks share the one GPU of the test box
(MPU_SHARE_GPU=1) over gloo, so that every leg the driver's SCALE run will execute on an 8-GPU node -- the
data-parallel train step with the overlapped all-reduce and its communication timing, the sharded 6-view
predict+fuse with both exchanges, --config 3 (global batch 32 of 256x256, strong scaling) and --config 4
(512^3 x 2, K = 5; here at a reduced edge) -- runs end to end and prints the JSON the driver parses.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, port, nproc=2):
    env = dict(os.environ, MPU_SHARE_GPU="1", MPU_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_two_ranks_default_config_has_train_comm_and_sharded_predict_legs():
    d = _run(["--steps", "3", "--warmup", "1", "--predict-dim", "64", "--no-cpu-baseline", "--no-peaks"], 29611)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["rccl_ranks"] == 2 and d["value"] > 0
    assert d["config"]["global_batch"] == 32 and d["config"]["parallelism"] == "dp2"
    c = d["comm"]
    assert c["steps_timed"] == 3 and c["comm_ms_per_step"] > 0 and c["allreduce_bytes"] == 4 * 31046339 + 4 * (c["allreduce_bytes"] // 4 - 31046339)
    assert 0.0 <= c["exposed_ms_per_step"] and c["overlap_fraction"] is not None and c["overlap_fraction"] <= 1.0
    pf = d["predict_fuse"]
    assert pf["n_gpus"] == 2 and pf["volume"] == "64^3x1" and set(pf["exchanges"]) == {"reduce_scatter", "all_gather"}
    for ex in pf["exchanges"].values():
        assert ex["value"] > 0 and ex["exchange_seconds_max_rank"] >= 0 and sum(ex["label_histogram"]) == 64 ** 3
    # both exchanges fuse the same predictions (fp32 sums in a different order: a handful of near-tie voxels may flip)
    ha, hb = pf["exchanges"]["reduce_scatter"]["label_histogram"], pf["exchanges"]["all_gather"]["label_histogram"]
    assert sum(abs(p - q) for p, q in zip(ha, hb)) <= 1e-4 * 64 ** 3, (ha, hb)
    assert d["roofline"]["frac"] > 0 and d["guard"]["finite"]


def test_bench_two_ranks_config3_strong_scaling_and_config4_sharded_predict():
    d = _run(["--config", "3", "--steps", "2", "--warmup", "1", "--no-predict", "--no-cpu-baseline", "--no-peaks"], 29613)
    assert d["scaling"] == "strong" and d["config"]["global_batch"] == 32 and d["config"]["slices_per_gpu"] == 16
    assert "256x256" in d["config"]["workload"] and "configs[3]" in d["config"]["workload"] and d["comm"]["comm_ms_per_step"] > 0
    d = _run(["--config", "4", "--predict-dim", "64", "--exchange", "reduce_scatter"], 29615)
    assert d["unit"] == "voxels/s" and d["n_gpus"] == 2 and "configs[4]" in d["config"]["workload"]
    pf = d["predict_fuse"]
    assert pf["volume"] == "64^3x2" and pf["classes"] == 5 and pf["exchanges"]["reduce_scatter"]["value"] == d["value"]


def test_bench_four_ranks_config3_eight_slices_per_rank():
    """four ranks (sharing the test box's GPU over gloo) on configs[3] -- 8 slices of 256 x 256 per rank, the
    three-bucket overlapped all-reduce with the weight-gradient groups flushed at the bucket boundaries."""
    d = _run(["--config", "3", "--steps", "2", "--warmup", "1", "--no-predict", "--no-cpu-baseline", "--no-peaks"], 29617, nproc=4)
    assert d["n_gpus"] == 4 and d["scaling"] == "strong" and d["config"]["global_batch"] == 32 and d["config"]["slices_per_gpu"] == 8
    c = d["comm"]
    assert c["buckets"] == 3 and c["comm_ms_per_step"] > 0 and d["guard"]["finite"] and d["guard"]["decreasing"]


def test_bench_single_gpu_default_line_follows_the_driver_contract():
    """`python bench.py` as the driver runs it at N = 1 (defaults; here with fewer steps and without the peak probes): ONE JSON line
    with the contract's keys, the metric / unit of BASELINE.json, the roofline and cpu_baseline objects, and the legs added since
    (predict_fuse, train_e2e, f32_mode, bf16x3_mode)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--no-peaks"], cwd=ROOT,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic" and d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] - 16 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]          # slices/s of 16 slices per step
    # (BASELINE.json: "2D slices/sec/GPU (train) + voxels/sec ..."; the driver's contract wants the WHOLE-JOB aggregate as `value`)
    assert d["unit"] == "slices/s" and d["metric"].startswith("2D slices/sec") and base["metric"].startswith("2D slices/sec")
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["launch"] in ("graph", "eager")
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["traffic"] is None or rf["traffic"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == d["unit"] and cb["sample"]
    assert d["predict_fuse"]["value"] > 0 and d["predict_fuse"]["roofline"]["bound"] == "hbm"
    assert 0.5 < d["train_e2e"]["fraction_of_headline"] < 1.2
    assert d["f32_mode"]["ms_per_step"] > d["bf16x3_mode"]["ms_per_step"] > d["ms_per_step"]
    assert d["guard"]["finite"]
