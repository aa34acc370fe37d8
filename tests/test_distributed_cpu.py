"""world_size-2 gloo tests (CPU) of the multi-GPU orchestration: gradient SUM all-reduce, plane sharding, slab exchange."""
import os
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multiplanarunet_amd import distributed as D


def test_plane_work_items_cover_every_plane_once():
    for V, P, world in ((6, 276, 8), (6, 532, 8), (6, 276, 4), (6, 276, 2), (3, 37, 8), (6, 276, 1), (1, 5, 8)):
        per_rank = D.plane_work_items(V, P, world)
        assert len(per_rank) == world
        seen = np.zeros((V, P), int)
        for items in per_rank:
            for v, lo, hi in items:
                assert 0 <= lo < hi <= P
                seen[v, lo:hi] += 1
        assert (seen == 1).all()
        if V * P >= world * 8:
            loads = [sum(hi - lo for _, lo, hi in it) for it in per_rank]
            assert min(loads) > 0 and max(loads) <= 1.35 * (V * P / world) + 1, (V, P, world, loads)
        owners = [sum(1 for it in per_rank for v, lo, hi in it if v == vv and lo == 0) for vv in range(V)]
        assert owners == [1] * V          # exactly one rank adds each view's OOB term


def test_gradient_buckets_tile_the_flat_buffer():
    """plan_buckets (overlapped all-reduce): buckets tile [0, n) exactly once, in completion order (descending
    offsets), each is released by a ready point at or below its low end, and every bucket but the last holds at
    least bucket_bytes."""
    rng = np.random.RandomState(0)
    for trial in range(200):
        n = int(rng.randint(1, 10 ** 6))
        k = int(rng.randint(1, 12))
        pts = sorted({int(v) for v in rng.randint(0, n, size=k - 1)} - {0}, reverse=True) + [0]
        bb = int(rng.choice([4, 1024, 1 << 16, 1 << 22]))
        b = D.plan_buckets(pts, n, bb)
        assert b and b[0][2] == n and b[-1][1] == 0
        for (k0, lo0, hi0), (k1, lo1, hi1) in zip(b[:-1], b[1:]):
            assert lo0 == hi1 and k1 > k0 and hi0 > lo0
        for i, (kk, lo, hi) in enumerate(b):
            assert pts[kk] <= lo                       # everything in [lo, hi) is final once point kk has fired
            if i < len(b) - 1:
                assert (hi - lo) * 4 >= bb
    # BASELINE configs[1]: 124 MB of fp32 gradients, 32 MB buckets -> a handful of buckets, the first closed early
    pts = [31030720, 30900000, 28000000, 20000000, 12000000, 4700000, 1200000, 300000, 40000, 0]
    b = D.plan_buckets(pts, 31046339, 32 << 20)
    assert [x[0] for x in b] == [3, 5, 9] and b[0][1] == 20000000


def test_slab_bounds():
    for X, w in ((256, 8), (30, 4), (7, 8)):
        b = D.slab_bounds(X, w)
        assert b[0][0] == 0 and b[-1][1] == X and all(b[i][1] == b[i + 1][0] for i in range(w - 1))


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, dev = D.init_from_env("gloo")
    assert (r, w) == (rank, world) and dev.type == "cpu"
    # 1. bucketed gradient SUM all-reduce (sum, not mean: Keras reduction=NONE)
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    D.allreduce_sum_(g, bucket_bytes=1024)
    ok1 = torch.equal(g, torch.arange(1000, dtype=torch.float32) * sum(range(1, world + 1)))
    # 2. DataParallelTrainer hook + weight broadcast on a stand-in model
    class M:
        pass
    m = M()
    m.params = torch.full((10,), float(rank)); m.bn_state = torch.full((4,), float(rank))
    m._repack = lambda: None
    t = D.DataParallelTrainer(m, bucket_bytes=16)
    ok2 = bool((m.params == 0).all() and (m.bn_state == 0).all())
    gg = torch.ones(10) * (rank + 1)
    m._grad_hook(gg)
    ok2 = ok2 and bool((gg == 3).all())
    # 3. reduce-scatter over X slabs + all-gather (ragged X)
    X = 7
    z = torch.ones((X, 2, 3, 2)) * (rank + 1) + torch.arange(X).reshape(X, 1, 1, 1)
    zs, (lo, hi) = D.reduce_scatter_slabs(z.clone())
    exp = (torch.ones((X, 2, 3, 2)) * 3 + 2 * torch.arange(X).reshape(X, 1, 1, 1))[lo:hi]
    ok3 = torch.equal(zs, exp)
    # ragged X through the padded equal-slab layout RCCL's reduce_scatter_tensor needs (all-reduce stands in on gloo)
    for Xr in (7, 9, 2, 1):
        zr = torch.ones((Xr, 2, 3)) * (rank + 1) + torch.arange(Xr).reshape(Xr, 1, 1)
        zp, (plo, phi) = D.reduce_scatter_slabs(zr.clone(), force_pad=True)
        ok3 = ok3 and (plo, phi) == D.slab_bounds(Xr, world)[rank] and \
            torch.equal(zp, (torch.ones((Xr, 2, 3)) * 3 + 2 * torch.arange(Xr).reshape(Xr, 1, 1))[plo:phi])
    lab = zs[..., 0].to(torch.uint8)
    full = D.all_gather_slabs(lab, X)
    ok3 = ok3 and full.shape == (X, 2, 3) and torch.equal(full, (3 + 2 * torch.arange(X)).reshape(X, 1, 1).expand(X, 2, 3).to(torch.uint8))
    q.put((rank, ok1, ok2, ok3))
    dist.destroy_process_group()


def test_gloo_world2_collectives():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(r[0] for r in res) == [0, 1]
    for r in res:
        assert r[1] and r[2] and r[3], r
