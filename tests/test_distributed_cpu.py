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


# --------------------------------------------------------------------------- #
# world size 8: the WHOLE sharded predict and the bucketed gradient all-reduce with eight gloo
# ranks on CPU. The HIP kernels of the product are replaced by the oracle's NumPy restatements (test infrastructure),
# so what runs is the product's orchestration: work items, the OOB owner, ragged X slabs, both exchanges.
# --------------------------------------------------------------------------- #
def _toy_problem():
    rng = np.random.RandomState(5)
    X, Y, Z, C, K, V, dim = 13, 11, 9, 1, 3, 3, 12
    image = rng.rand(X, Y, Z, C).astype(np.float32)
    affine = np.diag([1.0, 1.1, 0.9, 1.0])
    views = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.2], [0.3, 0.2, 1.0]])
    views /= np.linalg.norm(views, axis=1, keepdims=True)
    W = rng.rand(V, K).astype(np.float32) + 0.5
    b = (rng.rand(1, K).astype(np.float32) - 0.5) * 0.1
    A = rng.randn(C, K).astype(np.float32) * 3
    return image, affine, views, W, b, A, dim, 14.0, K


def _toy_predict(A):
    def f(Xs):                                             # [P,d,d,C] -> softmax over a fixed per-pixel linear map
        z = np.asarray(Xs, np.float32) @ A
        e = np.exp(z - z.max(-1, keepdims=True))
        return (e / e.sum(-1, keepdims=True)).astype(np.float32)
    return f


def _install_oracle_standins(monkey_targets, image, affine):
    """Replace the HIP-backed functions of multiplanarunet_amd.interpolation by oracle restatements (CPU)."""
    from oracle import geometry as G
    I = monkey_targets
    vg = G.voxel_grid_real_space(image.shape[:3], affine)

    def sample_view(volume, geom, want_labels=True, out=None):
        planes = []
        for off in geom.offsets:
            grid, _g, _ib = G.sample_plane_at(geom.view, geom.dim, geom.span, off)
            im, _ = G.view_interpolate(image, None, affine, 0.0, 0, grid)
            planes.append(im)
        return torch.tensor(np.stack(planes, 0)), None

    def nearest_idx(grid, inv_basis):
        pts = np.stack([vg[i].ravel() for i in range(3)], axis=1)
        pts = inv_basis.dot(pts.T).T
        idx, _dist, oob = G.rgi_find_indices(pts.T, grid)
        res = []
        for i, yi, g in zip(idx, _dist, grid):
            res.append(np.where(yi <= .5, i, i + 1))
        return res, oob

    def map_accumulate(volume, pred_chunk, grid, inv_basis, Wv, p_lo, p_hi, owns_oob, z):
        (i0, i1, i2), oob = nearest_idx(grid, inv_basis)
        K = pred_chunk.shape[-1]
        pc = np.asarray(pred_chunk)                        # [p_hi - p_lo, d, d, K]; full view = [P][row g0][col g1]
        zz = z.numpy().reshape(-1, K)
        Wn = np.asarray(Wv, np.float32).reshape(K)
        mine = (~oob) & (i2 >= p_lo) & (i2 < p_hi)
        zz[mine] += Wn * pc[i2[mine] - p_lo, i0[mine], i1[mine]]
        if owns_oob:
            fill = np.zeros(K, np.float32); fill[0] = 1.0
            zz[oob] += Wn * fill
        return z

    def fusion_finalize(z, b=None, sum_fusion=False, want_probs=True):
        zz = z.numpy() + (0 if sum_fusion else np.asarray(b, np.float32).reshape(-1))
        return None, torch.tensor(zz.argmax(-1).astype(np.uint8))

    def map_real_space_pred(pred, grid, inv_basis, volume, method="nearest"):
        return torch.tensor(G.map_real_space_pred(np.asarray(pred), grid, inv_basis, vg))

    I.sample_view, I.map_accumulate, I.fusion_finalize, I.map_real_space_pred = \
        sample_view, map_accumulate, fusion_finalize, map_real_space_pred


def _worker8(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    D.init_from_env("gloo")
    from multiplanarunet_amd import interpolation as I
    from oracle import geometry as G
    image, affine, views, W, b, A, dim, span, K = _toy_problem()
    _install_oracle_standins(I, image, affine)
    f = _toy_predict(A)

    class Vol:
        device = torch.device("cpu")
        n_channels = image.shape[-1]
    vol = Vol(); vol.image = torch.tensor(image)

    class Model:
        n_classes = K
        def predict(self, Xs, batch_size=None):
            return torch.tensor(f(Xs.numpy()))

    class Fusion:
        def predict(self, x, batch_size=None):             # FusionLayer: softmax(sum_v W_v x_v + b)
            z = (x.numpy() * W[None]).sum(1) + b
            e = np.exp(z - z.max(-1, keepdims=True))
            return torch.tensor(e / e.sum(-1, keepdims=True))
    fus = Fusion(); fus.W = torch.tensor(W); fus.b = torch.tensor(b)
    # single-process oracle pipeline (every rank computes it: cheap at this size)
    merged, _, _ = G.multi_view_predict(image, affine, views, dim, span, f, W, b)
    ref = merged.argmax(-1).astype(np.uint8)
    ok = {}
    for ex in ("reduce_scatter", "all_gather"):
        t = {}
        out = D.multi_view_predict_sharded(Model(), vol, views, dim, span, fusion_model=fus, exchange=ex, timings=t)
        ok[ex] = bool(np.array_equal(out.numpy(), ref)) and out.shape == ref.shape
        ok[ex + "_items"] = t.get("work_items", -1)
    out = D.multi_view_predict_sharded(Model(), vol, views, dim, span, sum_fusion=True, exchange="reduce_scatter")
    ms, _, _ = G.multi_view_predict(image, affine, views, dim, span, f, W, b, sum_fusion=True)
    ok["sum_fusion"] = bool(np.array_equal(out.numpy(), ms.argmax(-1).astype(np.uint8)))
    # the bucketed gradient all-reduce with the configs[1] ready points scaled down (three buckets, as on the GPU)
    class M:
        pass
    m = M()
    n = 31046
    m.params = torch.full((16,), float(rank)); m.bn_state = torch.zeros(4); m._repack = lambda: None
    m.grads = torch.zeros(n)
    m.grad_ready_points = lambda: [31030, 30900, 28000, 20000, 12000, 4700, 1200, 300, 40, 0]
    tr = D.DataParallelTrainer(m, bucket_bytes=32 << 10)
    ok["buckets"] = [x[0] for x in tr.buckets] == [3, 5, 9] and not tr.overlap
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    m._grad_hook(g)
    ok["allreduce"] = bool(torch.equal(g, torch.arange(n, dtype=torch.float32) * 36.0)) and bool((m.params == 0).all())
    # ragged reduce-scatter (padded layout) + label all-gather at world 8
    for Xr in (13, 8, 5, 1):
        zr = torch.ones((Xr, 2, 3)) * (rank + 1) + torch.arange(Xr).reshape(Xr, 1, 1)
        zp, (lo, hi) = D.reduce_scatter_slabs(zr.clone(), force_pad=True)
        exp = (torch.ones((Xr, 2, 3)) * 36 + 8 * torch.arange(Xr).reshape(Xr, 1, 1))[lo:hi]
        full = D.all_gather_slabs(zp[..., 0].contiguous(), Xr)
        ok["rs%d" % Xr] = bool(torch.equal(zp, exp)) and bool(torch.equal(full, (36 + 8 * torch.arange(Xr)).reshape(Xr, 1).expand(Xr, 2).float()))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_gloo_world8_sharded_predict_and_bucketed_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(r[0] for r in res) == list(range(8))
    for rank, ok in res:
        assert all(v for k, v in ok.items() if not k.endswith("_items")), (rank, ok)
    items = sorted(ok["reduce_scatter_items"] for _, ok in res)
    assert items[0] >= 1 and sum(items) == 24            # 3 views x 8 chunks dealt over 8 ranks


# --------------------------------------------------------------------------- #
# SURVEY 8e row 3 (round 5): data-parallel FusionModel.fit. Two gloo ranks, each holding every second point of every
# batch, must end with the weights / losses / val_dice of ONE rank fitting all points (shuffle off; fp32 sum order aside).
# The three HIP-backed pieces (local gradient sums, apply, forward) are replaced by torch-CPU restatements built on
# oracle/fusion_train_ref.py, so what runs here is the product's orchestration: batch shares, the two all-reduces,
# ranks that run out of points, the shared early-stopping decision.
# --------------------------------------------------------------------------- #
def _fusion_points(n=1500, V=3, K=3, seed=3):
    rng = np.random.RandomState(seed)
    y = rng.randint(0, K, n).astype(np.uint8)
    x = rng.rand(n, V, K).astype(np.float32)
    x[np.arange(n), :, y] += 0.6 * rng.rand(n, V).astype(np.float32)       # views lean towards the target class
    x /= x.sum(-1, keepdims=True)
    return x, y


def _install_fusion_standins(fm):
    from oracle import fusion_train_ref as FR
    from oracle.unet_ref import adam_update
    V, K = fm.n_inputs, fm.n_classes

    def local_sums(xd, yd):
        out = torch.zeros(V * K + K + 2, dtype=torch.float64)
        n = int(xd.shape[0])
        out[-1] = n
        if n:
            W = fm.W.double().clone().requires_grad_(True)
            b = fm.b.double().clone().requires_grad_(True)
            ls = FR.sparse_generalized_dice_loss(yd.long(), FR.fusion_forward(W, b, xd.double()), fm.weight).sum()
            ls.backward()
            out[:V * K] = W.grad.reshape(-1); out[V * K:V * K + K] = b.grad.reshape(-1); out[V * K + K] = ls.detach()
        return out

    def apply_sums(sums, apply=True, want_grads=False):
        n = float(sums[-1])
        W, b = fm.W.double(), fm.b.double()
        loss = sums[V * K + K] / n + FR.reg(W) + FR.reg(b)
        gW = (sums[:V * K].reshape(V, K) / n + 2e-6 * W / (V * K)).float().numpy()
        gb = (sums[V * K:V * K + K].reshape(1, K) / n + 2e-6 * b / K).float().numpy()
        if apply:
            fm.iterations += 1
            k = fm.optimizer_kwargs
            mW, mb = fm._adam_m[:V * K].reshape(V, K).numpy(), fm._adam_m[V * K:].reshape(1, K).numpy()
            vW, vb = fm._adam_v[:V * K].reshape(V, K).numpy(), fm._adam_v[V * K:].reshape(1, K).numpy()
            W2, mW2, vW2 = adam_update(fm.W.numpy(), gW, mW, vW, fm.iterations, k["lr"], k["beta_1"], k["beta_2"], k["epsilon"])
            b2, mb2, vb2 = adam_update(fm.b.numpy(), gb, mb, vb, fm.iterations, k["lr"], k["beta_1"], k["beta_2"], k["epsilon"])
            fm.W, fm.b = torch.tensor(np.asarray(W2, np.float32)), torch.tensor(np.asarray(b2, np.float32))
            fm._adam_m = torch.tensor(np.concatenate([np.ravel(mW2), np.ravel(mb2)]).astype(np.float32))
            fm._adam_v = torch.tensor(np.concatenate([np.ravel(vW2), np.ravel(vb2)]).astype(np.float32))
        return loss.float().reshape(1), None

    fm._local_sums = local_sums
    fm._apply_sums = apply_sums
    fm.predict = lambda x, batch_size=0, verbose=0: FR.fusion_forward(fm.W.double(), fm.b.double(), torch.as_tensor(x).double()).float()


def _fit_fusion(x, y, xv, yv, batch_size, epochs, early):
    from multiplanarunet_amd.fusion_model import FusionModel
    fm = FusionModel(x.shape[1], x.shape[2], weight="Simple", verbose=False, device="cpu", logger=lambda *a: None)
    fm.compile("Adam", optimizer_kwargs={"lr": 5e-2})
    _install_fusion_standins(fm)
    h = fm.fit(x, y, batch_size=batch_size, epochs=epochs, shuffle=False, validation_data=(xv, yv), early_stopping=early,
               data_parallel=True)
    return fm.get_weights(), h, fm.iterations


def _worker_fusion(rank, world, port, q, scenario):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    D.init_from_env("gloo")
    x, y = _fusion_points()
    xv, yv = _fusion_points(400, seed=11)
    if scenario == "interleaved":          # rank r holds points r, r + world, ...: the union of the shares of step j is batch j
        xs, ys, xvs, yvs = x[rank::world], y[rank::world], xv[rank::world], yv[rank::world]
    else:                                  # "starved": rank 1 holds no points at all (fewer images than ranks)
        xs, ys, xvs, yvs = (x, y, xv, yv) if rank == 0 else (x[:0], y[:0], xv[:0], yv[:0])
    (W, b), h, it = _fit_fusion(xs, ys, xvs, yvs, 400, 6, 2)
    q.put((rank, W, b, h, it))
    dist.destroy_process_group()


@pytest.mark.parametrize("scenario", ["interleaved", "starved"])
def test_gloo_world2_fusion_fit_equals_one_rank(scenario):
    x, y = _fusion_points()
    xv, yv = _fusion_points(400, seed=11)
    # one rank, all points. "starved": rank 0 takes ceil(400 / 2) = 200 points per step, so the reference run uses batch 200
    (W1, b1), h1, it1 = _fit_fusion(x, y, xv, yv, 400 if scenario == "interleaved" else 200, 6, 2)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + (7 if scenario == "starved" else 0)
    procs = [ctx.Process(target=_worker_fusion, args=(r, 2, port, q, scenario)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
    assert [r[0] for r in res] == [0, 1]
    for rank, W, b, h, it in res:
        assert it == it1 and len(h["loss"]) == len(h1["loss"])              # same number of steps and epochs on every rank
        np.testing.assert_allclose(W, W1, rtol=0, atol=2e-6)
        np.testing.assert_allclose(b, b1, rtol=0, atol=2e-6)
        np.testing.assert_allclose(h["loss"], h1["loss"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(h["val_dice"], h1["val_dice"], rtol=0, atol=1e-7)
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])   # identical on both ranks, bitwise
    assert np.abs(W1 - 1.0).max() > 1e-2                                                   # (the fit moved the weights)
