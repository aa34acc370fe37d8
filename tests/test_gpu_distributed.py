"""
Two ranks sharing the single test GPU (gloo backend; RCCL needs one GPU per rank): the
plane-sharded multi-view predict (reduce-scatter of partial fusion sums + all-gather of label
slabs) equals the single-process pipeline, and data-parallel training sums replica gradients.
"""
import os
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), MPU_SHARE_GPU="1", MPU_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from multiplanarunet_amd import distributed as D
    from multiplanarunet_amd.unet import UNet
    from multiplanarunet_amd.fusion_model import FusionModel
    from multiplanarunet_amd.interpolation import Volume
    from multiplanarunet_amd.predict import multi_view_predict
    quiet = lambda *a, **k: None
    r, w, dev = D.init_from_env()
    rng = np.random.RandomState(0)
    Dv, K = 32, 3
    vol = (rng.randn(Dv, Dv, Dv - 3, 1) * 40 + 90).astype(np.float32)       # ragged X slabs / Z
    views = np.array([[0, 0, 1], [1, 0, 0], [0.3, 0.5, 0.8]], float)
    model = UNet(n_classes=K, dim=Dv, depth=2, dtype="f32", logger=quiet, seed=3, device=dev)
    fm = FusionModel(len(views), K, verbose=False, device=dev)
    fm.set_weights([rng.uniform(.5, 1.5, (3, K)).astype(np.float32), rng.uniform(-.1, .1, (1, K)).astype(np.float32)])
    v = Volume(vol, None, np.eye(4), bg_value=[0.0], device=dev)
    _, ref = multi_view_predict(model, v, views, Dv, float(Dv), fm, batch_size=8)
    got = D.multi_view_predict_sharded(model, v, views, Dv, float(Dv), fm, batch_size=8)
    ok_pred = bool((got != ref).float().mean().item() <= 1e-4) and tuple(got.shape) == tuple(ref.shape)
    got2 = D.multi_view_predict_sharded(model, v, views, Dv, float(Dv), None, sum_fusion=True, batch_size=8)
    _, ref2 = multi_view_predict(model, v, views, Dv, float(Dv), None, sum_fusion=True, batch_size=8)
    ok_pred = ok_pred and bool((got2 != ref2).float().mean().item() <= 1e-4)
    # data parallel: rank-specific batches, gradients summed, weights stay identical
    m2 = UNet(n_classes=K, dim=32, depth=2, dtype="f32", logger=quiet, seed=10 + rank, device=dev)
    D.DataParallelTrainer(m2)                              # broadcasts rank 0's weights
    x = np.random.RandomState(100 + rank).randn(2, 32, 32, 1).astype(np.float32)
    y = np.random.RandomState(200 + rank).randint(0, K, (2, 32 * 32, 1)).astype(np.uint8)
    m2.forward_backward(x, y, None)
    local = m2.grads.clone()
    m2._grad_hook(m2.grads)
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    ok_dp = torch.allclose(m2.grads, sum(gathered), rtol=1e-5, atol=1e-6)
    m2.apply_gradients()
    ps = [torch.zeros_like(m2.params) for _ in range(world)]
    dist.all_gather(ps, m2.params)
    ok_dp = ok_dp and torch.equal(ps[0], ps[1])
    q.put((rank, ok_pred, ok_dp))
    dist.destroy_process_group()


def test_two_ranks_sharded_predict_and_dp_training():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    for r in res:
        assert r[1] and r[2], r
