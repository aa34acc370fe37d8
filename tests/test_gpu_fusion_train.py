"""
GPU parity of the fusion-model training step (csrc/fusion_train.hip via FusionModel.fit/train_on_batch)
against the oracle restatement of the reference's GDL loss + FusionLayer + Keras Adam
(oracle/fusion_train_ref.py). f32 kernel vs f64 oracle: loss 1e-6, gradients 1e-6 + 1e-5 relative,
weights after 20 Adam steps 2e-5 (Adam's sqrt(v) normalisation amplifies f32 rounding of tiny gradients).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
quiet = lambda *a, **k: None


def make_points(N, V, K, seed):
    rng = np.random.RandomState(seed)
    y = rng.randint(0, K, N).astype(np.uint8)
    x = rng.rand(N, V, K).astype(np.float32)
    x[:, 0, :] += 2.0 * np.eye(K, dtype=np.float32)[y]          # view 0 is informative
    x /= x.sum(-1, keepdims=True)
    return x, y


@pytest.mark.parametrize("V,K,N", [(6, 3, 5000), (3, 2, 777), (1, 5, 300), (16, 8, 2048)])
def test_loss_and_gradients_match_oracle(V, K, N):
    from multiplanarunet_amd.fusion_model import FusionModel
    from oracle import fusion_train_ref as F
    x, y = make_points(N, V, K, 3)
    rng = np.random.RandomState(4)
    W = (1 + 0.2 * rng.randn(V, K)).astype(np.float32)
    b = (0.1 * rng.randn(1, K)).astype(np.float32)
    fm = FusionModel(V, K, verbose=False, logger=quiet)
    fm.set_weights([W, b])
    loss, gW, gb = fm.loss_and_gradients(x, y)
    l_ref, gW_ref, gb_ref = F.loss_and_grads(W, b, x, y)
    assert abs(loss - l_ref) < 1e-6
    np.testing.assert_allclose(gW, gW_ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gb, gb_ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(fm.get_weights()[0], W)        # t = 0 does not update


def test_adam_steps_match_oracle_and_out_of_range_targets():
    from multiplanarunet_amd.fusion_model import FusionModel
    from oracle import fusion_train_ref as F
    V, K, N = 6, 3, 4096
    x, y = make_points(N, V, K, 7)
    y[::97] = 200                                                 # outside [0,K): zero one-hot, loss 1, no gradient
    fm = FusionModel(V, K, verbose=False, logger=quiet).compile()
    W = np.ones((V, K), np.float32); b = np.zeros((1, K), np.float32)
    m = dict(W=np.zeros_like(W), b=np.zeros_like(b)); v = dict(W=np.zeros_like(W), b=np.zeros_like(b))
    for t in range(1, 21):
        l_gpu = fm.train_on_batch(x, y)
        l_ref, W, b, m, v, _ = F.train_step(W, b, m, v, t, x, y)
        assert abs(l_gpu - l_ref) < 2e-6, (t, l_gpu, l_ref)
    Wg, bg = fm.get_weights()
    np.testing.assert_allclose(Wg, W, rtol=0, atol=2e-5)
    np.testing.assert_allclose(bg, b, rtol=0, atol=2e-5)


def test_fit_learns_to_trust_the_informative_view_and_stops_early():
    from multiplanarunet_amd.fusion_model import FusionModel
    V, K = 4, 3
    x, y = make_points(60000, V, K, 11)
    xv, yv = make_points(8000, V, K, 12)
    fm = FusionModel(V, K, verbose=False, logger=quiet).compile()
    h = fm.fit(x, y, batch_size=2 ** 13, epochs=40, validation_data=(xv, yv), early_stopping=3, seed=0)
    assert h["loss"][-1] < h["loss"][0]
    assert len(h["val_dice"]) == len(h["loss"]) <= 40
    W = fm.get_weights()[0]
    assert W[0].mean() > W[1:].mean()                             # the informative view got the larger weights
    assert h["val_dice"][-1] > 0.9
    # deterministic given the seed
    fm2 = FusionModel(V, K, verbose=False, logger=quiet).compile()
    h2 = fm2.fit(x, y, batch_size=2 ** 13, epochs=len(h["loss"]), validation_data=(xv, yv), seed=0)
    assert h2["loss"] == h["loss"][:len(h2["loss"])]


def test_two_halves_of_the_step_equal_the_fused_step_bitwise():
    """mpu_fusion_grad_sums + mpu_fusion_apply_sums (the data-parallel form) with ONE rank give bit for bit
    what mpu_fusion_train_step gives: loss, gradients, weights and Adam moments over several steps, incl. a short batch."""
    from multiplanarunet_amd.fusion_model import FusionModel
    V, K = 6, 3
    x, y = make_points(70000, V, K, 5)
    xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    a = FusionModel(V, K, verbose=False, logger=quiet).compile("Adam", optimizer_kwargs={"lr": 1e-2})
    b = FusionModel(V, K, verbose=False, logger=quiet).compile("Adam", optimizer_kwargs={"lr": 1e-2})
    for s, e in ((0, 32768), (32768, 65536), (65536, 70000), (0, 1), (5, 300)):
        la, ga = a._step(xd[s:e], yd[s:e], apply=True, want_grads=True)
        lb, n, gb = b._step_dp(xd[s:e], yd[s:e], apply=True, want_grads=True)
        assert float(n) == e - s
        assert torch.equal(la, lb) and torch.equal(ga, gb)
        assert torch.equal(a.W, b.W) and torch.equal(a.b, b.b)
        assert torch.equal(a._adam_m, b._adam_m) and torch.equal(a._adam_v, b._adam_v)
    # a rank without points contributes exact zeros and its count
    s0 = b._local_sums(xd[:0], yd[:0])
    assert s0.shape == (V * K + K + 2,) and float(s0.abs().sum()) == 0.0
    # fit(data_parallel=True) on one rank == the plain fit (same batches, same seed)
    ha = a.fit(xd, yd, batch_size=16384, epochs=3, validation_data=(xd[:5000], yd[:5000]), early_stopping=5, seed=3)
    hb = b.fit(xd, yd, batch_size=16384, epochs=3, validation_data=(xd[:5000], yd[:5000]), early_stopping=5, seed=3,
               data_parallel=True)
    assert ha["loss"] == hb["loss"] and torch.equal(a.W, b.W) and torch.equal(a.b, b.b)
    np.testing.assert_allclose(ha["val_dice"], hb["val_dice"], rtol=0, atol=1e-7)      # (host f32 Dice vs counts: same formula)


def _dp_fit_worker(rank, world, port, q):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), MPU_SHARE_GPU="1", MPU_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from multiplanarunet_amd import distributed as D
    from multiplanarunet_amd.fusion_model import FusionModel
    r, w, dev = D.init_from_env()
    x, y = make_points(40000, 6, 3, 21)
    xv, yv = make_points(9000, 6, 3, 22)
    fm = FusionModel(6, 3, verbose=False, logger=quiet, device=dev).compile("Adam", optimizer_kwargs={"lr": 1e-2})
    h = fm.fit(x[rank::world], y[rank::world], batch_size=8192, epochs=4, shuffle=False,
               validation_data=(xv[rank::world], yv[rank::world]), early_stopping=9)
    q.put((rank, fm.W.cpu().numpy(), fm.b.cpu().numpy(), h, fm.iterations))
    dist.destroy_process_group()


def test_two_ranks_sharing_the_gpu_fit_the_weights_one_rank_fits():
    """Two gloo ranks on this box's GPU, each holding every second point of every batch (the HIP kernels proper, the two
    all-reduces of FusionModel.fit): weights, losses and val_dice equal those of one rank fitting all points, to fp32 summation
    order (the per-block partial sums cover other points); both ranks end with bit-identical weights."""
    import os
    import torch.multiprocessing as mp
    from multiplanarunet_amd.fusion_model import FusionModel
    x, y = make_points(40000, 6, 3, 21)
    xv, yv = make_points(9000, 6, 3, 22)
    one = FusionModel(6, 3, verbose=False, logger=quiet).compile("Adam", optimizer_kwargs={"lr": 1e-2})
    h1 = one.fit(x, y, batch_size=8192, epochs=4, shuffle=False, validation_data=(xv, yv), early_stopping=9)
    W1, b1 = one.get_weights()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_dp_fit_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    for rank, W, b, h, it in res:
        assert it == one.iterations == 4 * 5
        np.testing.assert_allclose(W, W1, rtol=0, atol=2e-5)
        np.testing.assert_allclose(b, b1, rtol=0, atol=2e-5)
        np.testing.assert_allclose(h["loss"], h1["loss"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(h["val_dice"], h1["val_dice"], rtol=0, atol=1e-4)
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert np.abs(W1 - 1.0).max() > 1e-2
