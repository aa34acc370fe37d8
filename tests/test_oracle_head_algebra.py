"""The identities the training head of round 6 rests on (multiplanarunet_amd/csrc/unet_ops.hip, "head_bn_*"), checked on
the CPU in float64 with torch autograd through the oracle's own loss (oracle/unet_ref.keras_sparse_ce -- the restatement of the
reference's SparseCategoricalCrossentropy on clipped probabilities, mpunet/train/trainer.py:78-97, behind the last BatchNormalization
and the 1x1 softmax head of mpunet/models/unet.py:205-216). With n = gamma * xhat + beta (training-mode BatchNorm of x over the M
pixels), logits = n Wh + bh, dz = dLoss/dlogits, dn = dz Wh^T:

    sum_m dn[m][c]              = sum_k Wh[c][k] * dbh[k]                       (dbh = sum_m dz)
    sum_m dn[m][c] * xhat[m][c] = sum_k Wh[c][k] * T[c][k],    T[c][k] = sum_m xhat[m][c] * dz[m][k]
    dWh[c][k]                   = gamma_c * T[c][k] + beta_c * dbh[k]

so neither n nor dn has to exist as a tensor: the BatchNorm-backward sums and the head's weight gradient follow from T and dbh, and
dx = gamma * invstd * (dn - mean(dn) - xhat * mean(dn * xhat)) is then formed from a RECOMPUTED dn. No GPU needed."""
import numpy as np
import torch

from oracle import unet_ref as U


def _setup(M, C, K, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.relu(torch.randn(M, C, generator=g, dtype=torch.float64) + 0.3)       # the BatchNorm input is a ReLU output
    gamma = (torch.rand(C, generator=g, dtype=torch.float64) * 3 - 1.5)             # both signs
    beta = torch.rand(C, generator=g, dtype=torch.float64) * 0.6 - 0.3
    Wh = torch.randn(C, K, generator=g, dtype=torch.float64) * 0.3
    bh = torch.randn(K, generator=g, dtype=torch.float64) * 0.1
    y = torch.randint(0, K, (M,), generator=g)
    sw = torch.where(torch.arange(M) % 3 == 0, 0.33, 1.0).to(torch.float64)
    return x, gamma, beta, Wh, bh, y, sw


def test_batchnorm_backward_sums_and_head_weight_gradient_follow_from_T_and_dbh():
    M, C, K, eps = 4096, 64, 3, 1e-3
    x, gamma, beta, Wh, bh, y, sw = _setup(M, C, K, 0)
    x.requires_grad_(True); gamma.requires_grad_(True); beta.requires_grad_(True); Wh.requires_grad_(True); bh.requires_grad_(True)
    mean = x.mean(0); var = x.var(0, unbiased=False); invstd = 1.0 / torch.sqrt(var + eps)
    xhat = (x - mean) * invstd
    n = gamma * xhat + beta
    n.retain_grad()
    logits = n @ Wh + bh
    logits.retain_grad()
    probs = torch.softmax(logits, -1)
    loss = U.keras_sparse_ce(probs.reshape(1, M, K), y.reshape(1, M, 1), torch.ones(1, dtype=torch.float64)).reshape(M) * sw
    loss.sum().backward()                                          # the gradient of the SUM (trainer.py: reduction NONE, summed)
    dz, dn = logits.grad, n.grad
    xh = xhat.detach()
    T = xh.T @ dz                                                  # [C][K]
    dbh = dz.sum(0)
    W = Wh.detach()
    # the three identities
    assert torch.allclose(dn.sum(0), W @ dbh, rtol=1e-10, atol=1e-12)
    assert torch.allclose((dn * xh).sum(0), (W * T).sum(1), rtol=1e-10, atol=1e-12)
    assert torch.allclose(Wh.grad, gamma.detach()[:, None] * T + beta.detach()[:, None] * dbh[None, :], rtol=1e-10, atol=1e-12)
    # ... and what they are used for: BatchNorm's parameter gradients and the data gradient from a recomputed dn
    s0, s1 = W @ dbh, (W * T).sum(1)
    assert torch.allclose(beta.grad, s0, rtol=1e-10, atol=1e-12) and torch.allclose(gamma.grad, s1, rtol=1e-10, atol=1e-12)
    dn_re = dz @ W.T
    dx = gamma.detach() * invstd.detach() * (dn_re - s0 / M - xh * (s1 / M))
    assert torch.allclose(x.grad, dx, rtol=1e-9, atol=1e-12)
    assert torch.allclose(bh.grad, dbh, rtol=1e-12, atol=1e-14)


def test_pool_backward_can_recompute_the_argmax_from_the_batchnorm_input():
    """maxpool_bwd_add_kernel<RECOMP> / maxpool_bwd_bn_fold_kernel: the pooled tensor is max over 2x2 windows of n = scale * x + shift
    (the affine output: gamma may be negative, so NOT the window's max of x); recomputing n from x with the same scale / shift gives
    the same arg-max, hence the same routing of the pooled gradient -- in float64 here, in the kernels with the forward's bf16 rounding."""
    rng = np.random.RandomState(1)
    B, H, W, C = 2, 8, 8, 16
    x = np.maximum(rng.randn(B, H, W, C), 0)
    scale, shift = rng.uniform(-1.5, 1.5, C), rng.uniform(-.3, .3, C)
    n = x * scale + shift
    g = rng.randn(B, H // 2, W // 2, C)
    nt = torch.tensor(n.transpose(0, 3, 1, 2), requires_grad=True)
    torch.nn.functional.max_pool2d(nt, 2).backward(torch.tensor(g.transpose(0, 3, 1, 2)))
    ref = nt.grad.numpy().transpose(0, 2, 3, 1)
    win = (x * scale + shift).reshape(B, H // 2, 2, W // 2, 2, C).transpose(0, 1, 3, 5, 2, 4).reshape(B, H // 2, W // 2, C, 4)
    arg = win.argmax(-1)                                           # first maximum, as the kernels (strict > over positions 1..3)
    out = np.zeros((B, H // 2, W // 2, C, 4)); np.put_along_axis(out, arg[..., None], g[..., None], -1)
    got = out.reshape(B, H // 2, W // 2, C, 2, 2).transpose(0, 1, 4, 2, 5, 3).reshape(B, H, W, C)
    assert np.array_equal(got != 0, ref != 0) and np.allclose(got, ref)
