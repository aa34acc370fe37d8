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
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-predict", action="store_true")
    ap.add_argument("--predict-only", action="store_true", help="only the 6-view predict+fuse leg (profiling aid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-peaks", action="store_true", help="skip the MFMA / stream-triad peak probes (profiling runs)")
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
    # synthetic 3-class targets that depend on the image (thresholds of a smoothed copy), so that the correctness
    # guard below can require the loss to fall over the timed steps
    xs = torch.nn.functional.avg_pool2d(x.permute(0, 3, 1, 2), 5, 1, 2)[:, 0]
    y = ((xs > 0.1).to(torch.uint8) + (xs > 0.45).to(torch.uint8)).reshape(B, dim * dim, 1).contiguous()
    sw = torch.ones(B, device=device)

    def current_loss():
        """mean per-pixel loss of the fixed batch under the current weights (train-mode forward; no update)."""
        state = model.bn_state.clone()
        _, l = model.forward_backward(x, y, sw, want_loss=True)
        model.bn_state.copy_(state)
        return float(l.mean().item())

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
    loss_first = current_loss()
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
    loss_last = current_loss()
    # per-step times of a further K steps (events around each step): median next to the mean of the timed region
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a_, b_ in evs:
        a_.record(); run(); b_.record()
    torch.cuda.synchronize()
    per_step = sorted(a_.elapsed_time(b_) for a_, b_ in evs)
    sched = {}
    if True:                                     # which kernel schedules one step dispatches to (tests assert the same);
        import ctypes as C                       # every rank runs the step (it contains the all-reduce when N > 1)
        lib.mpu_schedule_log_enable(1)
        step()
        n_ = lib.mpu_schedule_log_read(None, 0)
        buf = C.create_string_buffer(int(n_) + 1)
        lib.mpu_schedule_log_read(buf, n_ + 1)
        lib.mpu_schedule_log_enable(0)
        for line in buf.value.decode().splitlines():
            k = " ".join(line.split()[:2])
            sched[k] = sched.get(k, 0) + 1
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
    guard_bad = 0.0 if (np.isfinite(loss_first) and np.isfinite(loss_last) and loss_last <= 1.05 * loss_first) else 1.0
    if world > 1:
        t = torch.tensor([dt, guard_bad], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt, guard_bad = float(t[0].item()), float(t[1].item())
    if guard_bad:                                # NaN / diverging on some rank: every rank stops (no half-dead job)
        if world > 1:
            torch.distributed.destroy_process_group()
        raise SystemExit("bench.py correctness guard failed on rank %d view: loss %r -> %r" % (rank, loss_first, loss_last))

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
            "ms_per_step_median": round(per_step[len(per_step) // 2], 4),
            "ms_per_step_min": round(per_step[0], 4),
            "guard": {"loss_before_timed_steps": round(loss_first, 5), "loss_after_timed_steps": round(loss_last, 5),
                      "finite": bool(np.isfinite(loss_first) and np.isfinite(loss_last)),
                      "decreasing": bool(loss_last < loss_first)},
            "schedules": sched,
        }
        if world > 1:
            out["rccl_ranks"] = torch.distributed.get_world_size()
            out["dist_backend"] = torch.distributed.get_backend()
            out["config"]["dp_overlap"] = bool(getattr(model._grad_hook, "overlap", False))
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
    if rank == 0 and world == 1 and not args.no_peaks:
        out["measured_peaks"] = measured_peaks(device)
        if "roofline" in out:
            for k in ("roofline", "wgrad"):
                out[k]["peak_measured"] = out["measured_peaks"]["mfma_bf16_tflops"]
                out[k]["frac_of_measured_peak"] = round(out[k]["achieved"] / out["measured_peaks"]["mfma_bf16_tflops"], 4)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(B, dim)
        if "predict_fuse" in out:
            out["predict_fuse"]["cpu_baseline"] = cpu_baseline_predict()
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
    best, tim, labels = None, None, None
    for _ in range(reps):
        t = {}
        t0 = time.perf_counter()
        _, labels = multi_view_predict(model, vol, views, D, float(D), fm, batch_size=batch, want_probs=False, timings=t)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if best is None or el < best:
            best, tim = el, t
    hist = torch.bincount(labels.reshape(-1).long(), minlength=K).tolist()          # correctness guard of the leg
    if sum(hist) != D ** 3 or len(hist) != K or sum(1 for h in hist if h > 0) < 2:
        raise SystemExit("bench.py predict guard failed: label histogram %r" % (hist,))
    P = D + 20
    fuse_bytes = D ** 3 * (V * K * 4 + 1)                        # labels only (73 B/voxel)
    samp_bytes = V * (4 * D ** 3 + 4 * P * D * D)
    gflop = V * P * unet_forward_gflop(D)
    fuse_traffic = fuse_traffic_src = None       # HBM bytes per launch of the fused back-mapping from the round's PMC passes
    if D == 256 and V == 6 and K == 3:
        try:
            gf = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_geometry_pmc.json"))[-1]
            with open(os.path.join(ROOT, "profiles", gf)) as fh:
                fuse_traffic = int(json.load(fh)["kernels"]["map_fuse_fast_kernel<3,2>"]["hbm_bytes_per_launch"])
            fuse_traffic_src = "profiles/" + gf
        except (IndexError, KeyError, OSError, ValueError):
            pass
    return {"metric": "voxels/sec (6-view predict+fuse)", "value": round(D ** 3 / best, 1), "unit": "voxels/s",
            "volume": "%d^3x1" % D, "views": V, "planes_per_view": P, "seconds": round(best, 4),
            "sample_ms": round(tim["sample_ms"], 2), "unet_ms": round(tim["unet_ms"], 2),
            "map_fuse_ms": round(tim["map_fuse_ms"], 3),
            "unet_tflops_algorithmic": round(gflop / tim["unet_ms"], 1),
            "sample_GBs_compulsory": round(samp_bytes / tim["sample_ms"] / 1e6, 1),
            "map_fuse_GBs_algorithmic": round(fuse_bytes / tim["map_fuse_ms"] / 1e6, 1),
            "map_fuse_frac_of_hbm_peak": round(fuse_bytes / tim["map_fuse_ms"] / 1e6 / PEAK_HBM_GBS, 4),
            "roofline": {"bound": "hbm", "kernel": "map_fuse", "achieved": round(fuse_bytes / tim["map_fuse_ms"] / 1e6, 1),
                         "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(fuse_bytes / tim["map_fuse_ms"] / 1e6 / PEAK_HBM_GBS, 4), "traffic": fuse_traffic,
                         "traffic_source": fuse_traffic_src,
                         "algorithmic_bytes_per_launch": fuse_bytes},
            "label_histogram": hist}


def cpu_baseline(B, dim, budget_s=20.0):
    """The oracle restatement of the reference train step (torch-CPU fp32) on the host cores, on the FULL batch of
    the workload (B slices per step), repeated until ~budget_s of CPU work."""
    from oracle import unet_ref as U
    rng = np.random.RandomState(0)
    w = U.init_weights(3, 1, 4, 1, seed=0)
    x = rng.randn(B, dim, dim, 1).astype(np.float32)
    y = rng.randint(0, 3, (B, dim * dim, 1)).astype(np.uint8)
    sw = np.ones(B, np.float32)
    n, t0 = 0, time.perf_counter()
    while True:
        U.train_step(w, x, y, sw)
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 10:
            break
    el = time.perf_counter() - t0
    return {"value": round(n * B / el, 3), "unit": "slices/s", "cores": torch.get_num_threads(),
            "host_cpus": os.cpu_count(), "kind": "port",
            "sample": "%d oracle train steps (torch-CPU fp32 autograd + NumPy Adam) on %d slices of %dx%d; the reference's "
                      "TensorFlow-CPU path cannot run here (TF absent)" % (n, B, dim, dim)}


def cpu_baseline_predict(D=64, V=6, K=3):
    """The oracle 6-view predict+fuse pipeline (NumPy restatement of get_view_from / map_real_space_pred / FusionLayer,
    pinned by the reference goldens, + the torch-CPU U-Net) on a bounded sample: one D^3 volume."""
    from oracle import unet_ref as U
    from oracle import geometry as G
    rng = np.random.RandomState(0)
    vol = rng.randn(D, D, D, 1).astype(np.float32)
    views = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0], [0.5, 0.5, 0.707], [-0.6, 0.64, 0.48], [0.7, -0.5, 0.5]], float)[:V]
    w = U.init_weights(K, 1, 4, 1, seed=0)
    Wf, bf = np.ones((V, K), np.float32), np.zeros((1, K), np.float32)
    tu = [0.0]

    def pred(X):
        t = time.perf_counter()
        out = np.concatenate([U.predict(w, X[i:i + 28], depth=4) for i in range(0, X.shape[0], 28)])
        tu[0] += time.perf_counter() - t
        return out
    t0 = time.perf_counter()
    G.multi_view_predict(vol, np.eye(4), views, D, float(D), pred, Wf, bf, bg_value=[0.0], center=np.array([0.0]),
                         scale=np.array([1.349]))
    el = time.perf_counter() - t0
    return {"value": round(D ** 3 / el, 1), "unit": "voxels/s", "cores": torch.get_num_threads(), "host_cpus": os.cpu_count(),
            "kind": "port", "seconds": round(el, 2), "unet_seconds": round(tu[0], 2),
            "sample": "one %d^3x1 volume, %d views x %d planes of %dx%d through the oracle pipeline (NumPy geometry "
                      "restatement, threads as NumPy/torch choose; U-Net = torch-CPU fp32)" % (D, V, D + 20, D, D)}


def measured_peaks(device):
    """MFMA and HBM peaks measured on this box, quoted next to the spec values."""
    import ctypes as C
    from multiplanarunet_amd import _lib
    lib = _lib.load()
    st = _lib.stream_ptr()
    sink = torch.zeros(16, device=device)
    fl = C.c_double()
    ncu = torch.cuda.get_device_properties(device).multi_processor_count
    best = 0.0
    for blocks in (ncu * 2, ncu * 4):
        lib.mpu_probe_mfma_bf16(blocks, 2000, _lib.ptr(sink), C.byref(fl), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            lib.mpu_probe_mfma_bf16(blocks, 2000, _lib.ptr(sink), C.byref(fl), st)
        e1.record(); torch.cuda.synchronize()
        best = max(best, 3 * fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    n = 1 << 28                                                   # 3 x 1 GiB arrays: far beyond the 256 MB Infinity Cache
    a = torch.empty(n, device=device); b = torch.ones(n, device=device); c = torch.ones(n, device=device)
    lib.mpu_probe_stream_triad(_lib.ptr(a), _lib.ptr(b), _lib.ptr(c), n, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        lib.mpu_probe_stream_triad(_lib.ptr(a), _lib.ptr(b), _lib.ptr(c), n, st)
    e1.record(); torch.cuda.synchronize()
    triad = 5 * 12.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b, c
    return {"mfma_bf16_tflops": round(best, 1), "mfma_bf16_spec_tflops": PEAK_BF16_TFLOPS,
            "stream_triad_GBs": round(triad, 1), "hbm_spec_GBs": PEAK_HBM_GBS, "compute_units": ncu}


if __name__ == "__main__":
    main()
