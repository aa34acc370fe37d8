"""Known-answer tests of the fusion-training oracle against hand-derived closed forms (CPU)."""
import numpy as np
import torch

from oracle import fusion_train_ref as F


def test_gdl_per_point_closed_form_all_weight_types():
    rng = np.random.RandomState(0)
    N, K = 50, 4
    p = rng.dirichlet(np.ones(K), size=N)
    y = rng.randint(0, K, N)
    pc = p[np.arange(N), y]
    expect = 1 - (2.0 / K) * pc / (1 + pc + 1e-6)          # only the true class contributes; all class weights are 1
    for tw in ("Simple", "Square", "Uniform"):
        got = F.sparse_generalized_dice_loss(torch.tensor(y), torch.tensor(p), tw).numpy().reshape(-1)
        np.testing.assert_allclose(got, expect, rtol=1e-12, atol=1e-12)


def test_out_of_range_target_gives_loss_one_and_no_gradient():
    p = torch.tensor([[0.2, 0.8]], dtype=torch.float64, requires_grad=True)
    l = F.sparse_generalized_dice_loss(torch.tensor([5]), p, "Simple")
    assert float(l) == 1.0
    l.sum().backward()
    assert float(p.grad.abs().sum()) == 0.0


def test_gradient_matches_analytic_formula():
    rng = np.random.RandomState(1)
    N, V, K = 40, 3, 3
    x = rng.rand(N, V, K)
    y = rng.randint(0, K, N)
    W = 1 + 0.1 * rng.randn(V, K); b = 0.1 * rng.randn(1, K)
    loss, gW, gb = F.loss_and_grads(W, b, x, y)
    z = (W[None] * x).sum(1) + b
    p = np.exp(z - z.max(1, keepdims=True)); p /= p.sum(1, keepdims=True)
    pc = p[np.arange(N), y]
    dLdp = -(2.0 / K) * (1 + 1e-6) / (1 + pc + 1e-6) ** 2
    dz = dLdp[:, None] * pc[:, None] * (np.eye(K)[y] - p)
    np.testing.assert_allclose(gb.reshape(-1), dz.mean(0) + 2e-6 * b.reshape(-1) / K, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(gW, (dz[:, None, :] * x).mean(0) + 2e-6 * W / (V * K), rtol=1e-9, atol=1e-12)
    l_exp = (1 - (2.0 / K) * pc / (1 + pc + 1e-6)).mean() + 1e-6 * (W ** 2).mean() + 1e-6 * (b ** 2).mean()
    assert abs(loss - l_exp) < 1e-12


def test_train_step_moves_weights_towards_informative_view():
    rng = np.random.RandomState(2)
    N, V, K = 2000, 2, 3
    y = rng.randint(0, K, N)
    good = np.eye(K)[y] * 0.8 + 0.1 * rng.rand(N, K)
    bad = rng.rand(N, K)
    x = np.stack([good, bad], 1).astype(np.float32)
    W = np.ones((V, K), np.float32); b = np.zeros((1, K), np.float32)
    m = dict(W=np.zeros_like(W), b=np.zeros_like(b)); v = dict(W=np.zeros_like(W), b=np.zeros_like(b))
    l0 = None
    for t in range(1, 31):
        loss, W, b, m, v, _ = F.train_step(W, b, m, v, t, x, y)
        l0 = loss if l0 is None else l0
    assert loss < l0 and W[0].mean() > W[1].mean()
