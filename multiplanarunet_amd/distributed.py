"""
One process per GPU over torch.distributed (backend "nccl" == RCCL over xGMI on
ROCm; "gloo" on CPU for the tests). Replaces the reference's only multi-GPU
mechanism, tf.distribute.MirroredStrategy (mpunet/bin/train.py:349,
mpunet/bin/predict.py:214), for the two phases of the hot path:

  training : pure data parallelism over slices; one SUM all-reduce of the flat
             gradient buffer per step (the Keras loss is unreduced, so replica
             gradients are summed, not averaged); BatchNorm statistics stay
             per-replica as under MirroredStrategy.
  predict  : (view, plane-chunk) work items dealt over ranks; each rank
             accumulates W_v * nearest(x_v) for its planes into z_partial, then
             reduce-scatter(SUM) over X-slabs -> +b, softmax, argmax on the slab
             -> all-gather of the uint8 label slabs. Exact (fusion is linear in
             the views before the softmax, fusion_model.py:39) and V times less
             traffic than all-gathering per-view logits.
"""
import os
import numpy as np
import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise from torchrun's RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; returns (rank, world, device)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        # MPU_SHARE_GPU=1 (testing aid): every rank uses GPU 0, e.g. with MPU_DIST_BACKEND=gloo on a 1-GPU box
        if os.environ.get("MPU_SHARE_GPU") == "1":
            local = 0
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
    device = torch.device("cuda", local) if use_cuda else torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or os.environ.get("MPU_DIST_BACKEND") or ("nccl" if use_cuda else "gloo")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, device


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def allreduce_sum_(flat, bucket_bytes=64 << 20):
    """
    In-place SUM all-reduce of a flat gradient buffer in buckets issued back to
    back as async collectives (RCCL pipelines them over the xGMI links) and
    waited on together. 124 MB of fp32 gradients = 2 buckets at the default size.
    """
    if world_size() == 1:
        return flat
    n = flat.numel()
    step = max(1, bucket_bytes // flat.element_size())
    works = [dist.all_reduce(flat[s:s + step], op=dist.ReduceOp.SUM, async_op=True)
             for s in range(0, n, step)]
    for w in works:
        w.wait()
    return flat


def plan_buckets(ready_points, n, bucket_bytes, elem_bytes=4):
    """
    Gradient buckets for the overlapped all-reduce. ready_points: descending float offsets; after point k every
    gradient at offset >= ready_points[k] is final (the backward pass fills the flat buffer from its end). Returns
    [(point index, lo, hi)] in completion order: bucket [lo, hi) may be reduced once point `index` has fired. A
    bucket is closed at the first ready point that makes it >= bucket_bytes; the last one takes whatever remains, so
    the buckets tile [0, n) exactly once.
    """
    step = max(1, bucket_bytes // elem_bytes)
    buckets, hi = [], n
    for k, off in enumerate(ready_points):
        last = k == len(ready_points) - 1
        lo = 0 if last else off
        if hi - lo >= step or (last and (hi > lo or not buckets)):
            buckets.append((k, lo, hi))
            hi = lo
    return buckets


class DataParallelTrainer:
    """
    Wires the gradient all-reduce into UNet.train_step (model._grad_hook).

    overlap=True (default on GPUs): the backward pass finishes the flat gradient buffer from its end towards its
    start and records an event per gradient-ready point (mpu_unet_backward_events); buckets of >= bucket_bytes are
    all-reduced on a communication stream as soon as their event fires, while the remaining data/weight
    gradients are still being computed. Adam waits for all buckets. Same result as one all-reduce after the
    backward pass (SUM is element-wise).
    """

    def __init__(self, model, bucket_bytes=32 << 20, broadcast_weights=True, overlap=None):
        self.model = model
        self.bucket_bytes = bucket_bytes
        dev = getattr(model, "device", None)
        can = dev is not None and dev.type == "cuda" and hasattr(model, "grad_ready_points")
        if overlap is None and os.environ.get("MPU_DP_OVERLAP") == "0":
            overlap = False                        # A/B switch: one bucketed all-reduce after the backward pass
        self.overlap = can if overlap is None else (bool(overlap) and can)
        self.ready_events = None
        self.buckets = []                          # (point index, lo, hi) in the order the backward pass completes them
        if hasattr(model, "grad_ready_points"):    # the same bucket sequence with and without overlap (and on gloo / CPU:
            pts = model.grad_ready_points()        # the world-8 test runs exactly the collectives an 8-GPU step issues)
            self.buckets = plan_buckets(pts, model.grads.numel(), bucket_bytes)
        if self.overlap:
            self.ready_events = [None] * len(pts)
            for k, _, _ in self.buckets:
                ev = torch.cuda.Event()
                ev.record()                        # forces the underlying hipEvent_t into existence
                self.ready_events[k] = ev
            self.comm = torch.cuda.Stream(device=model.device)
            torch.cuda.synchronize()
        model._grad_hook = self
        if broadcast_weights and world_size() > 1:
            dist.broadcast(model.params, src=0)
            dist.broadcast(model.bn_state, src=0)
            model._repack()

    def __call__(self, grads):
        if world_size() == 1:
            return
        timing = getattr(self, "timing", False) and grads.is_cuda
        if timing:                                 # where the backward pass ends on the compute stream
            e_bwd = torch.cuda.Event(enable_timing=True)
            e_bwd.record()
        if not self.overlap:
            if timing:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if self.buckets and self.buckets[0][2] == grads.numel():
                works = [dist.all_reduce(grads[lo:hi], op=dist.ReduceOp.SUM, async_op=True) for _, lo, hi in self.buckets]
                for w in works:
                    w.wait()
            else:
                allreduce_sum_(grads, self.bucket_bytes)
            if timing:
                e1.record()
                self._timed.append((e_bwd, [(e0, e1)]))
            return
        # The comm stream is ordered behind the backward pass ONLY through the ready events. If this step's
        # backward did not record them (forward_backward called without ready_events), they are stale: order
        # the comm stream behind everything enqueued so far instead (no overlap, but never a race).
        fresh = getattr(self.model, "_ready_events_fresh", False)
        self.model._ready_events_fresh = False
        if not fresh:
            self.comm.wait_stream(torch.cuda.current_stream())
        spans = []
        with torch.cuda.stream(self.comm):
            for k, lo, hi in self.buckets:
                if fresh:
                    self.comm.wait_event(self.ready_events[k])
                if timing:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                w = dist.all_reduce(grads[lo:hi], op=dist.ReduceOp.SUM, async_op=True)
                w.wait()                           # the COMM stream waits for the collective (buckets run in order anyway)
                if timing:
                    e1.record()
                    spans.append((e0, e1))
        torch.cuda.current_stream().wait_stream(self.comm)
        if timing:
            self._timed.append((e_bwd, spans))

    # ---- communication timing (bench.py): events on the comm stream around every bucket ----------------------
    def start_timing(self):
        self.timing, self._timed = True, []

    def stop_timing(self):
        """Per-step means over the timed steps: comm_ms = sum of the bucket spans on the communication stream
        (from 'bucket ready' to 'collective done'), exposed_ms = how long after the end of the backward pass the
        last bucket finished (what Adam actually waits for), overlap_fraction = 1 - exposed / comm."""
        self.timing = False
        if not getattr(self, "_timed", None):
            return None
        torch.cuda.synchronize()
        comm, exposed = [], []
        for e_bwd, spans in self._timed:
            comm.append(sum(a.elapsed_time(b) for a, b in spans))
            exposed.append(max(0.0, e_bwd.elapsed_time(spans[-1][1])) if spans else 0.0)
        self._timed = []
        c, x = float(np.mean(comm)), float(np.mean(exposed))
        return {"comm_ms_per_step": round(c, 4), "exposed_ms_per_step": round(x, 4),
                "overlap_fraction": round(1.0 - x / c, 4) if c > 0 else None, "buckets": len(self.buckets) or 1,
                "bucket_mb": [round((hi - lo) * 4 / 2 ** 20, 1) for _, lo, hi in self.buckets] or None,
                "steps_timed": len(comm)}


# --------------------------------------------------------------------------- #
# predict sharding
# --------------------------------------------------------------------------- #
def plane_work_items(n_views, n_planes, world, chunks_per_view=None):
    """
    Deal contiguous plane chunks of every view round-robin over ranks. Returns
    per rank a list of (view, p_lo, p_hi). With V=6 plain view sharding would
    leave 2 of 8 GPUs idle; chunking planes keeps all ranks busy.
    """
    if chunks_per_view is None:
        chunks_per_view = world // np.gcd(world, n_views) if world > 1 else 1
        while n_views * chunks_per_view < world:
            chunks_per_view += 1
    items = []
    for v in range(n_views):
        cuts = np.linspace(0, n_planes, chunks_per_view + 1).round().astype(int)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            if hi > lo:
                items.append((v, int(lo), int(hi)))
    per_rank = [[] for _ in range(world)]
    for i, it in enumerate(items):
        per_rank[i % world].append(it)
    return per_rank


def slab_bounds(X, world):
    cuts = np.linspace(0, X, world + 1).round().astype(int)
    return [(int(a), int(b)) for a, b in zip(cuts[:-1], cuts[1:])]


def reduce_scatter_slabs(z_partial, force_pad=False):
    """
    SUM over ranks of z_partial [X,...]; each rank keeps its X-slab (slab_bounds). RCCL: one
    reduce_scatter_tensor; a ragged X (X % world != 0) is handled by scattering equal slabs of
    ceil(X / world) planes cut from a zero-padded copy laid out slab by slab (each rank's real planes first),
    so the traffic stays 1/world of an all-reduce. gloo has no reduce-scatter: all-reduce + slice
    (force_pad exercises the padded layout there too).
    """
    world = world_size()
    if world == 1:
        return z_partial, (0, z_partial.shape[0])
    rank = dist.get_rank()
    X = z_partial.shape[0]
    bounds = slab_bounds(X, world)
    lo, hi = bounds[rank]
    nccl = dist.get_backend() == "nccl"
    if X % world == 0 and nccl and not force_pad:
        out = torch.empty((X // world,) + tuple(z_partial.shape[1:]), dtype=z_partial.dtype,
                          device=z_partial.device)
        dist.reduce_scatter_tensor(out, z_partial.contiguous(), op=dist.ReduceOp.SUM)
        return out, (lo, hi)
    if nccl or force_pad:
        w = max(b - a for a, b in bounds)
        padded = torch.zeros((world * w,) + tuple(z_partial.shape[1:]), dtype=z_partial.dtype,
                             device=z_partial.device)
        for r, (a, b) in enumerate(bounds):
            padded[r * w:r * w + (b - a)] = z_partial[a:b]
        if nccl:
            out = torch.empty((w,) + tuple(z_partial.shape[1:]), dtype=z_partial.dtype, device=z_partial.device)
            dist.reduce_scatter_tensor(out, padded, op=dist.ReduceOp.SUM)
        else:
            dist.all_reduce(padded, op=dist.ReduceOp.SUM)
            out = padded[rank * w:(rank + 1) * w]
        return out[:hi - lo].contiguous(), (lo, hi)
    dist.all_reduce(z_partial, op=dist.ReduceOp.SUM)
    return z_partial[lo:hi].contiguous(), (lo, hi)


def all_gather_slabs(slab, X):
    """Concatenate every rank's slab along axis 0 (ragged slabs padded to the widest)."""
    world = world_size()
    if world == 1:
        return slab
    bounds = slab_bounds(X, world)
    wmax = max(b - a for a, b in bounds)
    pad = torch.zeros((wmax,) + tuple(slab.shape[1:]), dtype=slab.dtype, device=slab.device)
    pad[:slab.shape[0]] = slab
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:b - a] for p, (a, b) in zip(parts, bounds)], dim=0)


def multi_view_predict_sharded(model, volume, views, dim, real_space_span, fusion_model=None,
                               sum_fusion=False, batch_size=None, n_planes="same+20", exchange=None, timings=None,
                               want_probs=False):
    """
    multiplanarunet_amd.predict.multi_view_predict over all ranks. Every rank
    holds the full input volume; returns the full uint8 label volume on every rank -- with want_probs=True
    (`mp predict --no_argmax`) the pair (fused [X,Y,Z,K] f32 volume, labels): the slabs of the fused volume are
    all-gathered as well (K * 4 more bytes per voxel than the labels).

    exchange="reduce_scatter" (default; MPU_PREDICT_EXCHANGE overrides): plane-chunk work items, partial
    fusion sums, reduce-scatter + label all-gather (module docstring). exchange="all_gather": the literal
    scheme of the north star / mpunet/bin/predict.py:307-366 -- whole views dealt over ranks, every rank maps
    its views to the voxel grid (`mapped_v [X,Y,Z,K]`), ALL-GATHER of the per-view volumes rebuilds
    `combined [V,X,Y,Z,K]` on every rank, then the FusionLayer + argmax runs locally. V times the traffic and
    at most V busy ranks; kept for equivalence testing and for fusion models that are not linear in the views.
    """
    exchange = exchange or os.environ.get("MPU_PREDICT_EXCHANGE") or "reduce_scatter"
    if exchange == "all_gather":
        return _multi_view_predict_allgather(model, volume, views, dim, real_space_span, fusion_model,
                                             sum_fusion, batch_size, n_planes, timings, want_probs)
    if exchange != "reduce_scatter":
        raise ValueError("exchange must be 'reduce_scatter' or 'all_gather'")
    from .interpolation import ViewGeometry, sample_view, map_accumulate, fusion_finalize
    world = world_size()
    rank = dist.get_rank() if world > 1 else 0
    K = model.n_classes
    X, Y, Z = (int(v) for v in volume.image.shape[:3])
    if timings is not None:
        import time
        if volume.device.type == "cuda":
            torch.cuda.synchronize()
        timings["_t0"] = time.perf_counter()
    z = torch.zeros((X, Y, Z, K), dtype=torch.float32, device=volume.device)
    geoms = [ViewGeometry(v, dim, real_space_span, n_planes) for v in views]
    items = plane_work_items(len(views), geoms[0].n_planes, world)[rank]
    for (vi, lo, hi) in items:
        g = geoms[vi]
        sub = ViewGeometry(views[vi], dim, real_space_span, n_planes)
        sub.offsets = g.offsets[lo:hi]
        sub.n_planes = hi - lo
        Xs, _ = sample_view(volume, sub, want_labels=False)
        pred = model.predict(Xs, batch_size=batch_size)
        if pred.ndim == 3:
            pred = pred.reshape(Xs.shape[0], dim, dim, -1)
        if sum_fusion:
            Wv = torch.ones(K, dtype=torch.float32, device=volume.device)
        else:
            Wv = fusion_model.W[vi]
        map_accumulate(volume, pred, (g.real_axis, g.real_axis, g.offsets), g.inv_basis, Wv,
                       lo, hi, owns_oob=(lo == 0), z=z)
    # timings (bench.py): the rank's own work up to here, then the exchange (reduce-scatter + finalize + all-gather);
    # with the blocking collectives a host clock around a synchronised region is the honest measure
    if timings is not None:
        import time
        torch.cuda.synchronize() if z.is_cuda else None
        t1 = time.perf_counter()
        timings["compute_s"] = t1 - timings.pop("_t0", t1)
        timings["work_items"] = len(items)
        timings["planes"] = sum(hi_ - lo_ for _, lo_, hi_ in items)
    zs, (lo, hi) = reduce_scatter_slabs(z)
    b = None if sum_fusion else fusion_model.b
    probs_s, labels = fusion_finalize(zs, b, sum_fusion=sum_fusion, want_probs=want_probs)
    out = all_gather_slabs(labels, X)
    probs = all_gather_slabs(probs_s, X) if want_probs else None
    if timings is not None:
        torch.cuda.synchronize() if z.is_cuda else None
        timings["exchange_s"] = time.perf_counter() - t1
        timings["exchange_bytes_per_rank"] = int(z.numel() * 4 * (world - 1) // max(world, 1) + out.numel() * (world - 1) // max(world, 1))
    return (probs, out) if want_probs else out


def _multi_view_predict_allgather(model, volume, views, dim, real_space_span, fusion_model, sum_fusion,
                                  batch_size, n_planes, timings=None, want_probs=False):
    from .interpolation import ViewGeometry, sample_view, map_real_space_pred, pred_to_class
    import time
    world = world_size()
    t_ex = 0.0
    sync = (lambda: torch.cuda.synchronize()) if volume.device.type == "cuda" else (lambda: None)
    if timings is not None:
        sync(); t_begin = time.perf_counter()
    rank = dist.get_rank() if world > 1 else 0
    K, V = model.n_classes, len(views)
    X, Y, Z = (int(v) for v in volume.image.shape[:3])
    dev = volume.device
    rounds = -(-V // world)
    combined = torch.empty((V, X, Y, Z, K), dtype=torch.float32, device=dev)
    for r in range(rounds):
        vi = r * world + rank                               # view of this rank in this round (or none)
        mine = torch.zeros((X, Y, Z, K), dtype=torch.float32, device=dev)
        if vi < V:
            g = ViewGeometry(views[vi], dim, real_space_span, n_planes)
            Xs, _ = sample_view(volume, g, want_labels=False)
            pred = model.predict(Xs, batch_size=batch_size)
            if pred.ndim == 3:
                pred = pred.reshape(Xs.shape[0], dim, dim, -1)
            mine = map_real_space_pred(pred.permute(1, 2, 0, 3), (g.real_axis, g.real_axis, g.offsets),
                                       g.inv_basis, volume)
        if world > 1:
            if timings is not None:
                sync(); t1 = time.perf_counter()
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            if timings is not None:
                sync(); t_ex += time.perf_counter() - t1
        else:
            parts = [mine]
        for q, part in enumerate(parts):
            if r * world + q < V:
                combined[r * world + q] = part
    x = torch.movedim(combined, 0, -2).reshape(-1, V, K)    # predict.py:354-356
    if sum_fusion:
        merged = x.sum(dim=1)
    else:
        merged = fusion_model.predict(x, batch_size=10 ** 4)
        if not torch.is_tensor(merged):
            merged = torch.as_tensor(merged, device=dev)
    out = pred_to_class(merged.reshape(X, Y, Z, K), img_dims=3)
    if timings is not None:
        sync()
        timings["exchange_s"] = t_ex
        timings["compute_s"] = time.perf_counter() - t_begin - t_ex
        timings["exchange_bytes_per_rank"] = int(rounds * X * Y * Z * K * 4 * (world - 1))
    return (merged.reshape(X, Y, Z, K), out) if want_probs else out
