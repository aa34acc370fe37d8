#!/usr/bin/env python
"""
bench.py -- headline benchmark of the MI355X hot path (driver contract).

  python bench.py --gpus N --steps K --warmup W          (N>1: under torch.distributed.run)

Primary line (BASELINE.json configs[1], "2D slices/sec (train)"):
  one step = one Keras-equivalent train step (forward with batch-stat BN, sparse
  CE, backward, gradient SUM all-reduce over RCCL when N>1, Adam, weight repack)
  of the depth-4 / 64-filter U-Net on a batch of 16 bf16 128x128x1 slices per
  GPU, inputs resident in HBM. value = N*16*K / max-over-ranks time.

Added objects:
  roofline     : the dominant kernel (conv_igemm: forward + data-gradient MFMA
                 convolutions), algorithmic FLOPs / HIP-event kernel time measured
                 inside the timed region on the launch stream, vs 2.5 PFLOP/s dense bf16.
  wgrad        : same for the weight-gradient kernel.
  cpu_baseline : the oracle (torch-CPU fp32 restatement of the reference train
                 step) on the host cores, bounded sample (rank 0, N=1 only).
  predict_fuse : secondary metric of BASELINE.json -- voxels/s of the 6-view
                 predict+fuse pipeline on one 256^3 volume (N=1 only; --no-predict skips).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (guides/MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
FWD_GFLOP_PER_SLICE_128 = 27.27
TRAIN_GFLOP_PER_SLICE_128 = 81.8


def unet_forward_gflop(dim, n_channels=1, n_classes=3, depth=4, cf=1.0):
    """sum 2*M*N*K over the 23 convs (2x2 up-convs at output resolution) -- SURVEY.md 8a."""
    f = [int(64 * 2 ** l * np.sqrt(cf)) for l in range(depth + 1)]
    tot, cin = 0.0, n_channels
    for l in range(depth):
        m = (dim >> l) ** 2
        tot += 2 * m * 9 * (cin * f[l] + f[l] * f[l]); cin = f[l]
    m = (dim >> depth) ** 2
    tot += 2 * m * 9 * (cin * f[depth] + f[depth] * f[depth]); cin = f[depth]
    for j in range(depth):
        l = depth - 1 - j
        m = (dim >> l) ** 2
        tot += 2 * m * (4 * cin * f[l] + 9 * 2 * f[l] * f[l] + 9 * f[l] * f[l]); cin = f[l]
    tot += 2 * dim * dim * cin * n_classes
    return tot / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-predict", action="store_true")
    ap.add_argument("--predict-only", action="store_true", help="only the 6-view predict+fuse leg (profiling aid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch the step's kernels eagerly instead of replaying a HIP graph")
    args = ap.parse_args()

    from multiplanarunet_amd import distributed as D
    from multiplanarunet_amd import _lib
    from multiplanarunet_amd.unet import UNet
    rank, world, device = D.init_from_env()
    if device.type != "cuda":
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    quiet = lambda *a, **k: None
    B, dim = args.batch, args.dim
    if args.predict_only:
        print(json.dumps({"predict_fuse": bench_predict(device, quiet)}), flush=True)
        return

    model = UNet(n_classes=3, dim=dim, n_channels=1, depth=4, complexity_factor=1,
                 flatten_output=True, dtype=args.dtype, logger=quiet, seed=0, device=device)
    model.compile("Adam", "SparseCategoricalCrossentropy")
    if world > 1:
        D.DataParallelTrainer(model)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    x = torch.randn(B, dim, dim, 1, generator=g).to(device)
    y = torch.randint(0, 3, (B, dim * dim, 1), generator=g, dtype=torch.uint8).to(device)
    sw = torch.ones(B, device=device)

    def step():
        model.train_step(x, y, sw, want_loss=False)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    lib = _lib.load()
    events = not args.no_kernel_events
    # N=1: the whole step (fwd + bwd + Adam + repack, ~260 launches) is replayed from one HIP graph; the
    # per-launch HIP events of the roofline leg need eager launches, so they are taken over a second region
    # of the same K steps right after (same kernels, same arguments). N>1 keeps eager launches (RCCL between).
    graphed = world == 1 and not args.no_graph
    dt_eager = None
    if graphed:
        replay = model.make_graphed_train_step(x, y, sw)
        replay()
        run = replay
    else:
        run = step
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    barrier()
    dt = time.perf_counter() - t0
    if events:                                   # roofline leg: same K steps again, eager, per-launch HIP events on
        barrier()
        lib.mpu_profile_enable(1)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt_eager = time.perf_counter() - t0
    roof = {}
    if events:
        import ctypes as C
        for kind, name in ((0, "conv_igemm"), (1, "wgrad_igemm")):
            ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
            _lib.check(lib.mpu_profile_summary(kind, C.byref(ms), C.byref(fl), C.byref(n)), "mpu_profile_summary")
            roof[name] = (ms.value, fl.value, n.value)
        lib.mpu_profile_enable(0)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        gf_slice = 3.0 * unet_forward_gflop(dim)
        out = {
            "metric": "2D slices/sec (train), whole job", "value": round(value, 2), "unit": "slices/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "2-D U-Net train step, depth 4, 64 base filters (complexity_factor=1), "
                                   "%d slices of %dx%dx1 per GPU, 3 classes, Adam + sparse CE (BASELINE.json configs[1])"
                                   % (B, dim, dim),
                       "slices_per_gpu": B, "parallelism": "dp%d" % world,
                       "launch": "hip-graph replay" if graphed else "eager",
                       "algorithmic_gflop_per_slice": round(gf_slice, 2)},
            "step_tflops_algorithmic": round(gf_slice * B * world / 1e3 / (ms_step / 1e3), 1),
        }
        if dt_eager is not None:
            out["ms_per_step_eager_with_events"] = round(dt_eager / args.steps * 1e3, 4)
        traffic = {}
        try:   # HBM bytes per launch from the rocprofv3 PMC passes of this round (profiles/, see its note)
            tfile = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_hbm_traffic_pmc.json"))[-1]
            with open(os.path.join(ROOT, "profiles", tfile)) as f:
                traffic = {k: v["hbm_bytes_per_launch"] for k, v in json.load(f)["classes"].items()}
            out["config"]["traffic_source"] = "profiles/" + tfile
        except Exception:
            pass
        if events:
            def leg(name):
                ms, fl, n = roof[name]
                ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
                return {"bound": "mfma", "kernel": name, "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                        "traffic": round(traffic[name]) if name in traffic else None,
                        "launches_per_step": n // max(args.steps, 1),
                        "avg_launch_us": round(ms * 1e3 / max(n, 1), 2),
                        "kernel_ms_per_step": round(ms / args.steps, 4),
                        "algorithmic_gflop_per_step": round(fl / args.steps / 1e9, 1)}
            out["roofline"] = leg("conv_igemm")
            out["wgrad"] = leg("wgrad_igemm")

    # ---- secondary metric: 6-view predict+fuse on 256^3 (N=1) -------------------
    if rank == 0 and world == 1 and not args.no_predict:
        out["predict_fuse"] = bench_predict(device, quiet)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(B, dim)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def bench_predict(device, quiet, D=256, V=6, K=3, reps=2, batch=None):
    """BASELINE.json configs[2]: 6-view predict+fuse on one 256^3x1 synthetic volume."""
    from multiplanarunet_amd.unet import UNet
    from multiplanarunet_amd.fusion_model import FusionModel
    from multiplanarunet_amd.interpolation import Volume
    from multiplanarunet_amd.predict import multi_view_predict
    rng = np.random.RandomState(0)
    vol_np = rng.randn(D, D, D, 1).astype(np.float32)
    vol = Volume(vol_np, None, np.eye(4), bg_value=0.0, scaler=(np.array([0.0]), np.array([1.349])), device=device)
    views = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0], [0.5, 0.5, 0.707], [-0.6, 0.64, 0.48], [0.7, -0.5, 0.5]], float)[:V]
    model = UNet(n_classes=K, dim=D, n_channels=1, depth=4, complexity_factor=1, dtype="bf16", logger=quiet,
                 seed=0, device=device)
    fm = FusionModel(V, K, verbose=False, device=device)
    batch = batch or (int(os.environ["MPU_BENCH_PREDICT_BATCH"]) if "MPU_BENCH_PREDICT_BATCH" in os.environ else None)
    multi_view_predict(model, vol, views, D, float(D), fm, batch_size=batch, want_probs=False)      # warm-up
    torch.cuda.synchronize()
    best, tim = None, None
    for _ in range(reps):
        t = {}
        t0 = time.perf_counter()
        multi_view_predict(model, vol, views, D, float(D), fm, batch_size=batch, want_probs=False, timings=t)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if best is None or el < best:
            best, tim = el, t
    P = D + 20
    fuse_bytes = D ** 3 * (V * K * 4 + 1)                        # labels only (73 B/voxel)
    samp_bytes = V * (4 * D ** 3 + 4 * P * D * D)
    gflop = V * P * unet_forward_gflop(D)
    return {"metric": "voxels/sec (6-view predict+fuse)", "value": round(D ** 3 / best, 1), "unit": "voxels/s",
            "volume": "%d^3x1" % D, "views": V, "planes_per_view": P, "seconds": round(best, 4),
            "sample_ms": round(tim["sample_ms"], 2), "unet_ms": round(tim["unet_ms"], 2),
            "map_fuse_ms": round(tim["map_fuse_ms"], 3),
            "unet_tflops_algorithmic": round(gflop / tim["unet_ms"], 1),
            "sample_GBs_compulsory": round(samp_bytes / tim["sample_ms"] / 1e6, 1),
            "map_fuse_GBs_algorithmic": round(fuse_bytes / tim["map_fuse_ms"] / 1e6, 1),
            "map_fuse_frac_of_hbm_peak": round(fuse_bytes / tim["map_fuse_ms"] / 1e6 / PEAK_HBM_GBS, 4)}


def cpu_baseline(B, dim, budget_s=12.0):
    """The oracle restatement of the reference train step (torch-CPU fp32) on the host cores."""
    from oracle import unet_ref as U
    rng = np.random.RandomState(0)
    w = U.init_weights(3, 1, 4, 1, seed=0)
    bs = min(B, 4)                                                # bounded sample: 4 slices per step
    x = rng.randn(bs, dim, dim, 1).astype(np.float32)
    y = rng.randint(0, 3, (bs, dim * dim, 1)).astype(np.uint8)
    sw = np.ones(bs, np.float32)
    U.train_step(w, x, y, sw)                                     # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        U.train_step(w, x, y, sw)
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 20:
            break
    el = time.perf_counter() - t0
    return {"value": round(n * bs / el, 3), "unit": "slices/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d oracle train steps (torch-CPU fp32 autograd + NumPy Adam) on %d slices of %dx%d"
                      % (n, bs, dim, dim)}


if __name__ == "__main__":
    main()
