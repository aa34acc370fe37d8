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
  predict_fuse : secondary metric of BASELINE.json -- voxels/s (whole node) of the 6-view
                 predict+fuse pipeline on one 256^3 volume; N>1: the sharded pipeline
                 (plane-chunk work items, reduce-scatter of the partial fusion sums + label
                 all-gather, and the literal all-gather-of-per-view-volumes variant), exchange
                 time reported separately (--no-predict skips).
  comm         : N>1 -- per-step gradient all-reduce time on the communication stream, the part of
                 it Adam waits for, and the overlap fraction (HIP events).
  train_e2e    : N=1 -- what `mp train` delivers: the GPU plane sampler cutting batches from a 128^3 synthetic volume on a
                 side stream one batch ahead of the graphed train step (multiplanarunet_amd/pipeline.py), slices/s over
                 >= 100 steps next to the serial loop of round 4 (sampler, eager step, host read of the loss every step).
  bf16x3_mode  : N=1 -- the same for dtype "bf16x3" (f32 storage, split-bf16 products: tolerance-grade at ~2.6x the f32 mode's speed)
  f32_mode     : N=1 -- ms per step of the SAME workload in dtype f32 (exact-f32 MFMA): the mode that meets the north star's
                 logits tolerance (atol 1e-4); the headline line is the bf16 storage mode (Dice delta <= 1e-3).

--config selects the BASELINE.json configuration: 1 (default; the line above), 2 (predict leg only),
3 (train, GLOBAL batch 32 of 256x256 split 32/N per GPU: strong scaling), 4 (predict of a 512^3 x 2
volume, 5 classes, sharded over the N ranks).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC for RCCL (before the HIP runtime initialises)
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
    ap.add_argument("--graph", action="store_true", help="replay the HIP graph (default at N=1: whichever of the two launch modes the warm-up measures faster)")
    ap.add_argument("--config", type=int, default=1, choices=(1, 2, 3, 4), help="BASELINE.json configs[] index (see the module docstring)")
    ap.add_argument("--exchange", default="both", choices=("reduce_scatter", "all_gather", "both"),
                    help="N>1 predict leg: which exchange(s) to time")
    ap.add_argument("--predict-dim", type=int, default=0, help="override the predict volume edge (tests)")
    ap.add_argument("--cf", type=float, default=1.0, help="complexity_factor of the train-leg network (2 = the default project YAML)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the train_e2e (sampler -> step pipeline) and f32_mode legs")
    ap.add_argument("--e2e-only", action="store_true", help="only the train_e2e leg (dev aid: prints its JSON object)")
    args = ap.parse_args()

    from multiplanarunet_amd import distributed as D
    from multiplanarunet_amd import _lib
    from multiplanarunet_amd.unet import UNet
    rank, world, device = D.init_from_env()
    if device.type != "cuda":
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    quiet = lambda *a, **k: None
    B, dim = args.batch, args.dim
    scaling = "weak"
    if args.config == 3:                       # configs[3]: GLOBAL batch 32 of 256x256, split over the ranks
        if 32 % world:
            raise SystemExit("--config 3 needs a world size that divides 32")
        B, dim, scaling = 32 // world, 256, "strong"
    pD, pC, pK = (512, 2, 5) if args.config == 4 else (256, 1, 3)
    if args.predict_dim:
        pD = args.predict_dim
    if args.predict_only or args.config in (2, 4):
        if world == 1:
            res = bench_predict(device, quiet, D=pD, K=pK, C=pC)
        else:
            res = bench_predict_sharded(device, quiet, rank, world, args.exchange, D=pD, K=pK, C=pC)
        if rank == 0:
            if args.predict_only:
                print(json.dumps({"predict_fuse": res}), flush=True)
            else:
                line = {"metric": "voxels/sec whole-node (6-view predict+fuse)", "value": res["value"], "unit": "voxels/s",
                        "n_gpus": world, "steps": res.get("reps", 1), "warmup": 1, "ms_per_step": round(res["seconds"] * 1e3, 3),
                        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16",
                        "data": "synthetic",
                        "config": {"workload": "6-view predict+fuse of one %s volume, %d classes (BASELINE.json configs[%d])"
                                               % (res["volume"], pK, args.config), "parallelism": "plane-chunk x%d" % world},
                        "predict_fuse": res}
                if "roofline" in res:
                    line["roofline"] = res["roofline"]
                print(json.dumps(line), flush=True)
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return

    model = UNet(n_classes=3, dim=dim, n_channels=1, depth=4, complexity_factor=args.cf,
                 flatten_output=True, dtype=args.dtype, logger=quiet, seed=0, device=device)
    model.compile("Adam", "SparseCategoricalCrossentropy")
    trainer = D.DataParallelTrainer(model) if world > 1 else None
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
    if args.e2e_only:
        replay = model.make_graphed_train_step(x, y, sw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(args.steps):
            replay()
        torch.cuda.synchronize()
        head = B * args.steps / (time.perf_counter() - t0)
        print(json.dumps({"headline_slices_per_s": round(head, 1), "train_e2e": bench_train_e2e(model, device, B, dim, headline=head)}), flush=True)
        return
    loss_first = current_loss()
    events = not args.no_kernel_events
    # N=1: the whole step (fwd + bwd + Adam + repack, ~260 launches) is replayed from one HIP graph; the
    # per-launch HIP events of the roofline leg need eager launches, so they are taken over a second region
    # of the same K steps right after (same kernels, same arguments). N>1 keeps eager launches (RCCL between).
    graphed = world == 1 and not args.no_graph
    dt_eager = None
    launch_cal = None
    if graphed:
        replay = model.make_graphed_train_step(x, y, sw)
        replay()
        run = replay
        if not args.graph:
            # Round 6: the step's tail has two branches (the optimizer beside the weight gradients); replayed from a graph
            # every edge between the branches costs ~10 us on this runtime, launched eagerly from two streams less (gpurun R6h:
            # 2.52-2.54 ms eager against 2.55 replayed). Both launch modes run the same kernels; the untimed warm-up measures
            # each and the timed region uses the faster one (config.launch says which, launch_calibration_ms both).
            def timed(fn, n):
                barrier(); t = time.perf_counter()
                for _ in range(n):
                    fn()
                barrier()
                return (time.perf_counter() - t) / n * 1e3
            ncal = max(10, args.warmup)
            cal = {"graph": [], "eager": []}
            for _ in range(2):
                cal["graph"].append(timed(replay, ncal)); cal["eager"].append(timed(step, ncal))
            launch_cal = {k: round(min(v), 4) for k, v in cal.items()}
            if launch_cal["eager"] < launch_cal["graph"]:
                graphed, run = False, step
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
    comm = None
    if trainer is not None:                      # N>1: the all-reduce on the communication stream, timed by events
        barrier()                                # over a further K steps (outside the timed region)
        trainer.start_timing()
        for _ in range(args.steps):
            step()
        comm = trainer.stop_timing()
        if comm is not None:                     # the slowest rank's view
            t = torch.tensor([comm["comm_ms_per_step"], comm["exposed_ms_per_step"]], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            comm["comm_ms_per_step_max_rank"], comm["exposed_ms_per_step_max_rank"] = round(float(t[0]), 4), round(float(t[1]), 4)
            comm["allreduce_bytes"] = int(model.grads.numel() * 4)
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
        gf_slice = 3.0 * unet_forward_gflop(dim, cf=args.cf)
        cfg_name = {1: "configs[1]", 3: "configs[3]"}.get(args.config, "configs[1]")
        if args.cf != 1.0:
            cfg_name += ", complexity_factor=%g as the default project YAML" % args.cf
        out = {
            "metric": "2D slices/sec (train), whole job", "value": round(value, 2), "unit": "slices/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "2-D U-Net train step, depth 4, %d base filters (complexity_factor=%g), "
                                   "%d slices of %dx%dx1 per GPU, 3 classes, Adam + sparse CE (BASELINE.json %s)"
                                   % (model.filters[0] if hasattr(model, "filters") else int(64 * np.sqrt(args.cf)), args.cf, B, dim, dim, cfg_name),
                       "slices_per_gpu": B, "global_batch": B * world, "parallelism": "dp%d" % world,
                       "launch": "hip-graph replay" if graphed else "eager",
                       "launch_calibration_ms": launch_cal,
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
            if comm is not None:
                out["comm"] = comm
        if dt_eager is not None:
            out["ms_per_step_eager_with_events"] = round(dt_eager / args.steps * 1e3, 4)
        if world == 1 and not args.no_peaks:
            # what the step's cycle counts are worth in time: the chip does not hold its 2.4 GHz maximum under these kernels
            out["shader_clock_mhz_during_step"] = clock_during(lambda: [run() for _ in range(14)], device)
        traffic = {}
        try:   # HBM bytes per launch from the rocprofv3 PMC passes of this round (profiles/, see its note)
            tfile = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_hbm_traffic_pmc.json"))[-1]
            from multiplanarunet_amd.srchash import source_sha16, CONV_SOURCES
            with open(os.path.join(ROOT, "profiles", tfile)) as f:
                tj = json.load(f)
            if args.config != 1:     # the counter passes ran on configs[1]: their bytes per launch are not this workload's
                out["config"]["traffic_source"] = None
            elif tj.get("source_sha16") == source_sha16(CONV_SOURCES):
                traffic = {k: v["hbm_bytes_per_launch"] for k, v in tj["classes"].items()}
                out["config"]["traffic_source"] = "from_file:profiles/" + tfile      # a separate rocprofv3 --pmc pass, not this run
            else:   # the kernels changed after that counter pass: its bytes are not this build's
                out["config"]["traffic_source"] = "stale:profiles/%s (kernel sources changed since that PMC pass)" % tfile
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

    # ---- secondary metric: 6-view predict+fuse on 256^3 (N>1: sharded over the ranks, whole-node voxels/s) ---
    if not args.no_predict and args.config == 1:
        if world == 1:
            out["predict_fuse"] = bench_predict(device, quiet, D=pD)
        else:
            del model                                            # (the train-leg network is not needed any more)
            torch.cuda.empty_cache()
            pf = bench_predict_sharded(device, quiet, rank, world, args.exchange, D=pD)
            if rank == 0:
                out["predict_fuse"] = pf
    if rank == 0 and world == 1 and args.config == 1 and not args.no_e2e:
        out["train_e2e"] = bench_train_e2e(model, device, B, dim, headline=out["value"])
        out["f32_mode"] = bench_f32_mode(device, quiet, B, dim, args.cf, x, y, sw)
        out["bf16x3_mode"] = bench_f32_mode(device, quiet, B, dim, args.cf, x, y, sw, dtype="bf16x3")
    if rank == 0 and world == 1 and not args.no_peaks:
        out["measured_peaks"] = measured_peaks(device)       # informational (box- and clock-dependent); every `frac` in this
                                                             # line is against the SPEC peaks of MI355X_MICROARCH.md
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(B, dim)
        if "predict_fuse" in out:
            out["predict_fuse"]["cpu_baseline"] = cpu_baseline_predict()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


VIEWS6 = [[0, 0, 1], [1, 0, 0], [0, 1, 0], [0.5, 0.5, 0.707], [-0.6, 0.64, 0.48], [0.7, -0.5, 0.5]]


def _predict_setup(device, quiet, D, V, K, C):
    from multiplanarunet_amd.unet import UNet
    from multiplanarunet_amd.fusion_model import FusionModel
    from multiplanarunet_amd.interpolation import Volume
    rng = np.random.RandomState(0)
    vol_np = rng.randn(D, D, D, C).astype(np.float32)
    vol = Volume(vol_np, None, np.eye(4), bg_value=0.0, scaler=(np.zeros(C), np.full(C, 1.349)), device=device)
    views = np.array(VIEWS6, float)[:V]
    model = UNet(n_classes=K, dim=D, n_channels=C, depth=4, complexity_factor=1, dtype="bf16", logger=quiet,
                 seed=0, device=device)
    fm = FusionModel(V, K, verbose=False, device=device)
    if K > 2:          # a biased head / fusion bias so that every class appears in the synthetic output (guard below)
        with torch.no_grad():
            fm.b.copy_(torch.linspace(-0.02, 0.02, K, device=device).reshape(1, K))
    return vol, views, model, fm


def bench_predict(device, quiet, D=256, V=6, K=3, reps=5, batch=None, C=1):
    """BASELINE.json configs[2] (D=256, C=1, K=3) / configs[4] (D=512, C=2, K=5): 6-view predict+fuse on one synthetic
    volume, one GPU. Median of `reps` runs (the per-stage times are those of the median run)."""
    from multiplanarunet_amd.predict import multi_view_predict
    vol, views, model, fm = _predict_setup(device, quiet, D, V, K, C)
    batch = batch or (int(os.environ["MPU_BENCH_PREDICT_BATCH"]) if "MPU_BENCH_PREDICT_BATCH" in os.environ else None)
    multi_view_predict(model, vol, views, D, float(D), fm, batch_size=batch, want_probs=False)      # warm-up
    torch.cuda.synchronize()
    runs, labels = [], None
    for _ in range(reps):
        t = {}
        t0 = time.perf_counter()
        _, labels = multi_view_predict(model, vol, views, D, float(D), fm, batch_size=batch, want_probs=False, timings=t)
        torch.cuda.synchronize()
        runs.append((time.perf_counter() - t0, t))
    runs.sort(key=lambda r: r[0])
    best, tim = runs[len(runs) // 2]
    clk = clock_during(lambda: multi_view_predict(model, vol, views, D, float(D), fm, batch_size=batch, want_probs=False),
                       device, n=400, naps=16)
    hist = torch.bincount(labels.reshape(-1).long(), minlength=K).tolist()          # correctness guard of the leg
    if sum(hist) != D ** 3 or len(hist) != K or sum(1 for h in hist if h > 0) < 2:
        raise SystemExit("bench.py predict guard failed: label histogram %r" % (hist,))
    P = D + 20
    fuse_bytes = D ** 3 * (V * K * 4 + 1)                        # labels only (73 B/voxel)
    samp_bytes = V * (4 * D ** 3 * C + 4 * P * D * D * C)
    gflop = V * P * unet_forward_gflop(D, n_channels=C, n_classes=K)
    fuse_traffic = fuse_traffic_src = None       # HBM bytes per launch of the fused back-mapping from the round's PMC passes
    if D == 256 and V == 6 and K == 3:
        try:
            gf = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_geometry_pmc.json"))[-1]
            from multiplanarunet_amd.srchash import source_sha16, GEOMETRY_SOURCES
            with open(os.path.join(ROOT, "profiles", gf)) as fh:
                gj = json.load(fh)
            if gj.get("source_sha16") == source_sha16(GEOMETRY_SOURCES):
                fuse_traffic = int(gj["kernels"]["map_fuse_fast_kernel<3,2>"]["hbm_bytes_per_launch"])
                fuse_traffic_src = "from_file:profiles/" + gf
            else:
                fuse_traffic_src = "stale:profiles/%s (geometry.hip changed since that PMC pass)" % gf
        except (IndexError, KeyError, OSError, ValueError):
            pass
    return {"metric": "voxels/sec (6-view predict+fuse)", "value": round(D ** 3 / best, 1), "unit": "voxels/s",
            "volume": "%d^3x%d" % (D, C), "views": V, "classes": K, "planes_per_view": P, "seconds": round(best, 4),
            "reps": reps, "statistic": "median", "seconds_all": [round(r[0], 4) for r in runs],
            "shader_clock_mhz_during_predict": clk,
            "sample_ms": round(tim["sample_ms"], 2), "unet_ms": round(tim["unet_ms"], 2),
            "map_fuse_ms": round(tim["map_fuse_ms"], 3),
            "unet_tflops_algorithmic": round(gflop / tim["unet_ms"], 1),
            "unet_frac_of_mfma_peak": round(gflop / tim["unet_ms"] / PEAK_BF16_TFLOPS, 4),
            "sample_GBs_compulsory": round(samp_bytes / tim["sample_ms"] / 1e6, 1),
            "map_fuse_GBs_algorithmic": round(fuse_bytes / tim["map_fuse_ms"] / 1e6, 1),
            "map_fuse_frac_of_hbm_peak": round(fuse_bytes / tim["map_fuse_ms"] / 1e6 / PEAK_HBM_GBS, 4),
            "roofline": {"bound": "hbm", "kernel": "map_fuse", "achieved": round(fuse_bytes / tim["map_fuse_ms"] / 1e6, 1),
                         "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(fuse_bytes / tim["map_fuse_ms"] / 1e6 / PEAK_HBM_GBS, 4), "traffic": fuse_traffic,
                         "traffic_source": fuse_traffic_src,
                         "algorithmic_bytes_per_launch": fuse_bytes},
            "label_histogram": hist}


def bench_predict_sharded(device, quiet, rank, world, exchange="both", D=256, V=6, K=3, C=1, reps=3):
    """The same volume over all ranks (multiplanarunet_amd.distributed.multi_view_predict_sharded):
    every rank holds the volume, runs the U-Net on its (view, plane-chunk) work items and joins the exchange.
    value = D^3 / (barrier-to-barrier time, max over ranks) = whole-node voxels/s; the exchange (reduce-scatter of
    the partial fusion sums + finalize + label all-gather, or the all-gather of per-view volumes) is timed on its own."""
    import torch.distributed as dist
    from multiplanarunet_amd import distributed as Dm
    vol, views, model, fm = _predict_setup(device, quiet, D, V, K, C)
    legs = {}
    for ex in (("reduce_scatter", "all_gather") if exchange == "both" else (exchange,)):
        Dm.multi_view_predict_sharded(model, vol, views, D, float(D), fm, exchange=ex)      # warm-up
        runs = []
        for _ in range(reps):
            torch.cuda.synchronize(); dist.barrier()
            tm = {}
            t0 = time.perf_counter()
            labels = Dm.multi_view_predict_sharded(model, vol, views, D, float(D), fm, exchange=ex, timings=tm)
            torch.cuda.synchronize(); dist.barrier()
            el = time.perf_counter() - t0
            t = torch.tensor([el, tm["exchange_s"], tm["compute_s"]], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            runs.append((float(t[0]), float(t[1]), float(t[2]), tm))
        runs.sort(key=lambda r: r[0])
        el, ex_s, comp_s, tm = runs[len(runs) // 2]
        hist = torch.bincount(labels.reshape(-1).long(), minlength=K).tolist()
        if sum(hist) != D ** 3 or sum(1 for h in hist if h > 0) < 2:
            raise SystemExit("bench.py sharded predict guard failed: label histogram %r" % (hist,))
        legs[ex] = {"value": round(D ** 3 / el, 1), "unit": "voxels/s", "seconds": round(el, 4),
                    "exchange_seconds_max_rank": round(ex_s, 4), "compute_seconds_max_rank": round(comp_s, 4),
                    "exchange_bytes_per_rank": tm.get("exchange_bytes_per_rank"),
                    "work_items_rank0": tm.get("work_items"), "planes_rank0": tm.get("planes"),
                    "reps": reps, "statistic": "median", "label_histogram": hist}
    main_leg = legs.get("reduce_scatter") or next(iter(legs.values()))
    out = {"metric": "voxels/sec whole-node (6-view predict+fuse, sharded)", "value": main_leg["value"], "unit": "voxels/s",
           "volume": "%d^3x%d" % (D, C), "views": V, "classes": K, "planes_per_view": D + 20, "n_gpus": world,
           "seconds": main_leg["seconds"], "reps": reps, "exchanges": legs,
           "rccl_ranks": dist.get_world_size(), "dist_backend": dist.get_backend()}
    return out


def bench_train_e2e(model, device, B, dim, headline, steps=120, warmup=50):
    """sampler -> step, end to end, as `mp train` runs it (pipeline.TrainPipeline): planes of a 128^3
    synthetic volume cut by the HIP sampler (6 views, noise, foreground balancing: one 8-byte host read per candidate) on a
    side stream while the previous batch's graphed train step runs. Reported next to the serial form (same sampler, eager
    step, `.item()` per step -- the round-4 loop) and the sampler alone."""
    from multiplanarunet_amd.data import make_toy_volume, as_volume, random_views, TrainSampler
    from multiplanarunet_amd.pipeline import TrainPipeline
    img, lab, aff = make_toy_volume(128, 77)
    vol = as_volume(img, lab, aff, "1pct", "RobustScaler", device, "toy128")
    views = random_views(6, 60.0, 0)
    mk = lambda seed: TrainSampler([vol], views, dim, float(dim), B, 3, noise_sd=0.1, fg_batch_fraction=0.5, seed=seed)
    res = {"unit": "slices/s", "steps": steps, "volume": "128^3x1 synthetic", "views": 6}
    # the sampler alone
    s0 = mk(5)
    for _ in range(3):
        s0()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rounds = 0
    for _ in range(30):
        s0()
        rounds += getattr(s0, "rounds", 0)
    torch.cuda.synchronize()
    res["sampler_alone_slices_per_s"] = round(30 * B / (time.perf_counter() - t0), 1)
    res["sampler_host_reads_per_batch"] = round(rounds / 30.0, 2)      # one per round of candidates (data.TrainSampler)
    # serial loop (round 4): cut, eager step, host read of the loss
    s1 = mk(6)
    for _ in range(3):
        xb, yb, wb = s1(); float(model.train_step(xb, yb, wb).mean().item())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40):
        xb, yb, wb = s1(); float(model.train_step(xb, yb, wb).mean().item())
    torch.cuda.synchronize()
    res["serial_slices_per_s"] = round(40 * B / (time.perf_counter() - t0), 1)
    # overlapped pipeline
    pipe = TrainPipeline(model, mk(7))
    pipe.run_epoch(warmup)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = pipe.run_epoch(steps)                               # ends with ONE device read (the epoch loss)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res["value"] = round(steps * B / dt, 1)
    res["ms_per_step"] = round(dt / steps * 1e3, 4)
    res["epoch_loss"] = round(loss, 5)
    res["fraction_of_headline"] = round(res["value"] / headline, 4) if headline else None
    res["producer_stream_latency_us"] = round(pipe.side_latency_us, 1) if pipe.side_latency_us is not None else None
    res["producer_stream_candidates_ms_per_step"] = pipe.side_loop_ms     # the real-loop windows the stream was chosen by (warm-up steps)
    res["producer_stream_note"] = ("the stream is the fastest of the candidates' real-loop windows; producer_stream_latency_us is the fill "
                                   "probe's figure for it, a ranking hint only (in this process, after the other legs' streams, it reads "
                                   "several ms for every candidate although the loop runs at the step's rate)")
    res["launch"] = "sampler on a side stream (picked by measurement: fill probe, then the first candidates under the real loop) one batch ahead; step = hip-graph replay; loss summed on the device"
    return res


def bench_f32_mode(device, quiet, B, dim, cf, x, y, sw, steps=8, dtype="f32"):
    """The same train step in dtype f32 (v_mfma_f32_32x32x2_f32: exact f32 products, 1/16 of the bf16 matrix rate)."""
    from multiplanarunet_amd.unet import UNet
    m = UNet(n_classes=3, dim=dim, n_channels=1, depth=4, complexity_factor=cf, flatten_output=True, dtype=dtype,
             logger=quiet, seed=0, device=device)
    m.compile("Adam", "SparseCategoricalCrossentropy")
    replay = m.make_graphed_train_step(x, y, sw)
    replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del replay, m
    torch.cuda.empty_cache()
    note = {"f32": "parity mode: inference logits within 1e-6 of the f64 oracle (north-star bound 1e-4)",
            "bf16x3": "f32 storage, every product as three bf16 MFMAs on hi + lo split operands: inference logits within 1e-5 of the "
                      "f64 oracle (north-star bound 1e-4), every conv launch within 1.3e-5 of fp64 on its own inputs "
                      "(tests/test_gpu_replay.py)"}[dtype]
    return {"dtype": dtype, "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3), "value": round(steps * B / dt, 1),
            "unit": "slices/s", "note": note}


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


def cpu_baseline_predict(D=128, V=6, K=3, views_timed=1):
    """The oracle 6-view predict+fuse pipeline (NumPy restatement of get_view_from / map_real_space_pred / FusionLayer,
    pinned by the reference goldens, + the torch-CPU U-Net) on a bounded sample of a D^3 volume (round 5: 128^3; rounds 1-4
    ran 64^3): `views_timed` of the V views run in full -- every view is the same amount of work: D + 20 planes of D x D
    sampled, predicted and mapped back -- and their time is scaled by V / views_timed; the fusion of all V mapped views
    is timed in full (the timed views' volumes repeated). The 256^3 workload of the metric is 8.0x the voxels and 7.5x the
    U-Net FLOPs of this sample."""
    from oracle import unet_ref as U
    from oracle import geometry as G
    rng = np.random.RandomState(0)
    vol = rng.randn(D, D, D, 1).astype(np.float32)
    views = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0], [0.5, 0.5, 0.707], [-0.6, 0.64, 0.48], [0.7, -0.5, 0.5]], float)[:V]
    sel = [views[3], views[0]][:views_timed] if views_timed <= 2 else list(views[:views_timed])   # an oblique view first
    w = U.init_weights(K, 1, 4, 1, seed=0)
    Wf, bf = np.ones((V, K), np.float32), np.zeros((1, K), np.float32)
    tu = 0.0
    t0 = time.perf_counter()
    vg = G.voxel_grid_real_space(vol.shape[:3], np.eye(4))
    t_grid = time.perf_counter() - t0
    mapped = []
    t0 = time.perf_counter()
    for view in sel:
        Xs, _, grid, ib = G.get_view_from(vol, None, np.eye(4), view, D, float(D), bg_value=[0.0], center=np.array([0.0]),
                                          scale=np.array([1.349]))
        X = np.moveaxis(Xs, 2, 0)
        t = time.perf_counter()
        pred = np.concatenate([U.predict(w, X[i:i + 37], depth=4) for i in range(0, X.shape[0], 37)])
        tu += time.perf_counter() - t
        mapped.append(G.map_real_space_pred(np.moveaxis(pred, 0, 2), grid, ib, vg))
    t_views = time.perf_counter() - t0
    combined = np.stack([mapped[i % len(mapped)] for i in range(V)])
    t0 = time.perf_counter()
    G.merge_multi_view_preds(combined, Wf, bf, False)
    t_fuse = time.perf_counter() - t0
    factor = V / float(len(sel))
    el = t_grid + t_views * factor + t_fuse
    return {"value": round(D ** 3 / el, 1), "unit": "voxels/s", "cores": torch.get_num_threads(), "host_cpus": os.cpu_count(),
            "kind": "port" if len(sel) == V else "port-extrapolated",     #  views_timed < V: time x V / views_timed
            "seconds": round(el, 2), "seconds_measured": round(t_grid + t_views + t_fuse, 2),
            "unet_seconds": round(tu * factor, 2), "fuse_seconds": round(t_fuse, 2), "views_timed": len(sel),
            "extrapolation_factor_over_views": factor,
            "sample": "one %d^3x1 volume: %d of the %d views x %d planes of %dx%d in full through the oracle pipeline (NumPy "
                      "geometry restatement, threads as NumPy/torch choose; U-Net = torch-CPU fp32), their time x %.1f, + the "
                      "voxel grid and the fusion of all %d views in full; the metric's 256^3 volume is 8.0x the voxels"
                      % (D, len(sel), V, D + 20, D, D, factor, V)}


def clock_during(fn, device, n=400, naps=6):
    """Effective shader clock (MHz) while fn()'s kernels run: one sampler wave on a side stream (mpu_probe_clock) records
    (shader cycles, 100-MHz ticks) pairs; fn must keep the GPU busy for longer than the sampler (~n * naps * 8 k cycles)."""
    from multiplanarunet_amd import _lib
    lib = _lib.load()
    buf = torch.zeros(2 * n, dtype=torch.int64, device=device)
    side = torch.cuda.Stream(device=device)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        _lib.check(lib.mpu_probe_clock(_lib.ptr(buf), n, naps, _lib.stream_ptr()), "mpu_probe_clock")
    fn()
    torch.cuda.synchronize()
    s = buf.cpu().numpy().reshape(n, 2)
    lo, hi = n // 8, n - 1                                        # (skip the ramp at the start)
    dt = float(s[hi, 1] - s[lo, 1])
    return round(float(s[hi, 0] - s[lo, 0]) / dt * 100.0, 0) if dt > 0 else None


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
    # the same loop on pseudo-random operands: the rate (and clock) the part sustains under its power limit on real data
    blocks = ncu * 4
    lib.mpu_probe_mfma_bf16_random(blocks, 2000, _lib.ptr(sink), C.byref(fl), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        lib.mpu_probe_mfma_bf16_random(blocks, 2000, _lib.ptr(sink), C.byref(fl), st)
    e1.record(); torch.cuda.synchronize()
    best_rand = 6 * fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12
    clk_rand = clock_during(lambda: [lib.mpu_probe_mfma_bf16_random(blocks, 2000, _lib.ptr(sink), C.byref(fl), st) for _ in range(12)],
                            device, n=200, naps=4)
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
    clk_triad = clock_during(lambda: [lib.mpu_probe_stream_triad(_lib.ptr(a), _lib.ptr(b), _lib.ptr(c), n, st) for _ in range(6)], device, n=200, naps=4)
    # float4 copy (the guide's 6.29 TB/s figure): default policy, non-temporal, default policy grid-stride; 1 GiB -> 1 GiB
    copies = {}
    for variant, name in ((0, "copy"), (1, "copy_nt"), (2, "copy_gridstride")):
        _lib.check(lib.mpu_probe_stream_copy(_lib.ptr(a), _lib.ptr(b), n, variant, st), "mpu_probe_stream_copy")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.mpu_probe_stream_copy(_lib.ptr(a), _lib.ptr(b), n, variant, st)
        e1.record(); torch.cuda.synchronize()
        copies[name] = round(5 * 8.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.copy_(b); torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        a.copy_(b)                                                # the runtime's own device-to-device copy, for reference
    e1.record(); torch.cuda.synchronize()
    copies["copy_hipMemcpyDtoD"] = round(5 * 8.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
    # the same GiB read once in permuted runs of 128 B ... 4 KB (coalesced requests, DRAM pages opened out of order): what the
    # gather kernels can expect from HBM; "stream" = one run (the read half of a copy)
    perm = {}
    for run in (128, 256, 512, 1024, 4096, 1 << 30):
        _lib.check(lib.mpu_probe_permuted_read(_lib.ptr(b), _lib.ptr(a), n, run, st), "mpu_probe_permuted_read")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.mpu_probe_permuted_read(_lib.ptr(b), _lib.ptr(a), n, run, st)
        e1.record(); torch.cuda.synchronize()
        perm["stream" if run == 1 << 30 else str(run)] = round(5 * 4.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
    del a, b, c
    clk_mfma = clock_during(lambda: [lib.mpu_probe_mfma_bf16(ncu * 4, 2000, _lib.ptr(sink), C.byref(fl), st) for _ in range(12)], device, n=200, naps=4)
    return {"mfma_bf16_tflops": round(best, 1), "mfma_bf16_spec_tflops": PEAK_BF16_TFLOPS,
            "mfma_bf16_tflops_random_operands": round(best_rand, 1), "shader_clock_mhz_during_random_mfma_probe": clk_rand,
            "shader_clock_mhz_during_mfma_probe": clk_mfma, "shader_clock_mhz_during_triad": clk_triad, "shader_clock_mhz_max": 2400,
            "stream_triad_GBs": round(triad, 1), "stream_float4_GBs": copies, "read_once_permuted_runs_GBs": perm, "hbm_guide_copy_GBs": 6290.0,
            "hbm_spec_GBs": PEAK_HBM_GBS, "compute_units": ncu,
            "note": "in-house probes under this box's power / clock state (non-zero operands; 2 reads + 1 write): they sit "
                    "10-20 % below the guide's best micro-benchmarks (2495 TFLOP/s, 6.29 TB/s copy) and are NOT used as "
                    "denominators anywhere"}


if __name__ == "__main__":
    main()
