"""
Known-answer tests anchoring the U-Net oracle (oracle/unet_ref.py): the
reference's own tests hold no vectors for this part of the path,
and TensorFlow is absent, so the restatement is checked against independent
naive NumPy loops on hand-sized cases. CPU only.
"""
import numpy as np
import torch
from oracle import unet_ref as U


def naive_conv_same(x, k, b, relu=True):
    """x [H,W,Ci], k HWIO; TF SAME: pad_total=k-1, before=pad_total//2."""
    H, W, Ci = x.shape
    kh, kw, _, Co = k.shape
    pt, pl = (kh - 1) // 2, (kw - 1) // 2
    y = np.zeros((H, W, Co))
    for i in range(H):
        for j in range(W):
            for a in range(kh):
                for c in range(kw):
                    ii, jj = i + a - pt, j + c - pl
                    if 0 <= ii < H and 0 <= jj < W:
                        y[i, j] += x[ii, jj] @ k[a, c]
    y += b
    return np.maximum(y, 0) if relu else y


def naive_bn(x, g, b, mean, var):
    g, b, mean, var = (np.asarray(t, np.float64) for t in (g, b, mean, var))
    return (x - mean) / np.sqrt(var + 1e-3) * g + b


def naive_unet_depth1(w, x, training):
    """x [B,H,W,C] float64; returns probs, batch stats."""
    B = x.shape[0]
    def conv(t, n, relu=True):
        return np.stack([naive_conv_same(t[i], w[n + "/kernel"].astype(np.float64),
                                         w[n + "/bias"].astype(np.float64), relu)
                         for i in range(B)])
    def bn(t, n):
        if training:
            m, v = t.mean((0, 1, 2)), t.var((0, 1, 2))
        else:
            m, v = w[n + "/moving_mean"], w[n + "/moving_variance"]
        return naive_bn(t, w[n + "/gamma"], w[n + "/beta"], m, v)
    c = conv(conv(x, "encoder_L0_conv1"), "encoder_L0_conv2")
    skip = bn(c, "encoder_L0_BN")
    Bn, H, W, F = skip.shape
    p = skip.reshape(Bn, H // 2, 2, W // 2, 2, F).max((2, 4))
    bt = bn(conv(conv(p, "bottom_conv1"), "bottom_conv2"), "bottom_BN")
    up = np.repeat(np.repeat(bt, 2, 1), 2, 2)
    u1 = bn(conv(up, "upsample_L0_conv1"), "upsample_L0_BN1")
    cat = np.concatenate([skip, u1], -1)
    o = bn(conv(conv(cat, "upsample_L0_conv2"), "upsample_L0_conv3"),
           "upsample_L0_BN2")
    z = conv(o, "conv2d", relu=False)
    e = np.exp(z - z.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True)


def small_weights(seed=3):
    w = U.init_weights(3, n_channels=2, depth=1, complexity_factor=1 / 64., seed=seed)
    rng = np.random.RandomState(seed)
    for k in w:
        if k.endswith("kernel"):
            w[k] = rng.randint(-2, 3, w[k].shape).astype(np.float32) * 0.25
        elif k.endswith("bias"):
            w[k] = rng.randint(-1, 2, w[k].shape).astype(np.float32) * 0.5
        elif k.endswith("gamma"):
            w[k] = rng.uniform(0.5, 1.5, w[k].shape).astype(np.float32)
            w[k][0] = -0.75      # negative gamma: pool must follow the affine
        elif k.endswith("beta") or k.endswith("moving_mean"):
            w[k] = rng.uniform(-.5, .5, w[k].shape).astype(np.float32)
        elif k.endswith("moving_variance"):
            w[k] = rng.uniform(0.5, 2, w[k].shape).astype(np.float32)
    return w


def test_filter_counts_and_param_count():
    assert [U.filters_at(i, 2) for i in range(5)] == [90, 181, 362, 724, 1448]
    specs = U.layer_specs(3, 1, 4, 1)
    n_conv = sum(int(np.prod(s)) + s[-1] for _, k, s in specs if k == "conv")
    n_bn = sum(4 * s[0] for _, k, s in specs if k == "bn")
    assert n_conv == 31030723 and n_bn == 15616


def test_conv_same_asymmetric_2x2():
    x = np.arange(9, dtype=np.float64).reshape(1, 3, 3, 1)
    k = np.array([[1., 10.], [100., 1000.]]).reshape(2, 2, 1, 1)
    y = U._conv(torch.tensor(x).permute(0, 3, 1, 2), torch.tensor(k),
                torch.zeros(1, dtype=torch.float64), relu=False)[0, 0].numpy()
    # out[i,j] = x[i,j] + 10 x[i,j+1] + 100 x[i+1,j] + 1000 x[i+1,j+1], 0 beyond
    assert y[0, 0] == 0 + 10 * 1 + 100 * 3 + 1000 * 4
    assert y[2, 2] == 8 and y[0, 2] == 2 + 100 * 5 and y[2, 0] == 6 + 10 * 7


def test_depth1_forward_matches_naive_loops():
    w = small_weights()
    x = np.random.RandomState(1).randn(2, 4, 4, 2)
    for training in (False, True):
        p = U.to_torch(w, torch.float64)
        got = U.forward(p, torch.tensor(x), depth=1, training=training).numpy()
        ref = naive_unet_depth1(w, x, training)
        np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-11)


def test_bn_moving_stats_update():
    w = small_weights()
    x = np.random.RandomState(2).randn(2, 4, 4, 2).astype(np.float32)
    y = np.zeros((2, 4, 4), np.uint8)
    r = U.train_step(w, x, y, np.ones(2), depth=1, dtype=torch.float64)
    c = np.stack([naive_conv_same(naive_conv_same(
        x[i].astype(np.float64), w["encoder_L0_conv1/kernel"], w["encoder_L0_conv1/bias"]),
        w["encoder_L0_conv2/kernel"], w["encoder_L0_conv2/bias"]) for i in range(2)])
    n = 2 * 4 * 4
    np.testing.assert_allclose(
        r["weights"]["encoder_L0_BN/moving_mean"],
        0.99 * w["encoder_L0_BN/moving_mean"] + 0.01 * c.mean((0, 1, 2)), rtol=1e-6)
    np.testing.assert_allclose(
        r["weights"]["encoder_L0_BN/moving_variance"],
        0.99 * w["encoder_L0_BN/moving_variance"] + 0.01 * c.var((0, 1, 2)) * n / (n - 1),
        rtol=1e-6)


def test_ce_sum_gradient_finite_difference():
    w = small_weights(5)
    rng = np.random.RandomState(4)
    x = rng.randn(2, 4, 4, 2)
    y = rng.randint(0, 3, (2, 4, 4)).astype(np.uint8)
    sw = np.array([1.0, 0.33])
    r = U.train_step(w, x, y, sw, depth=1, dtype=torch.float64)

    def total(wd):
        p = U.to_torch(wd, torch.float64)
        pr = U.forward(p, torch.tensor(x), 1, True)
        return float(U.keras_sparse_ce(pr, torch.tensor(y.astype(np.int64)),
                                       torch.tensor(sw)).sum())
    for name, idx in (("conv2d/kernel", (0, 0, 0, 1)), ("upsample_L0_conv1/kernel", (1, 0, 0, 0)),
                      ("encoder_L0_conv1/bias", (0,)), ("bottom_BN/gamma", (0,)),
                      ("upsample_L0_BN1/beta", (0,))):
        h = 1e-6
        wp = {k: np.array(v, np.float64) for k, v in w.items()}
        wm = {k: np.array(v, np.float64) for k, v in w.items()}
        wp[name][idx] += h
        wm[name][idx] -= h
        fd = (total(wp) - total(wm)) / (2 * h)
        np.testing.assert_allclose(r["grads"][name][idx], fd, rtol=2e-5, atol=1e-7)


def test_keras_ce_value_and_weighting():
    probs = torch.tensor([[[[0.7, 0.2, 0.1], [1.0, 0.0, 0.0]]]], dtype=torch.float64)
    y = torch.tensor([[[1, 1]]])
    l = U.keras_sparse_ce(probs, y, torch.tensor([0.33], dtype=torch.float64)).numpy()
    q = np.clip(np.array([1.0, 0.0, 0.0]), 1e-7, 1 - 1e-7)
    np.testing.assert_allclose(l[0, 0, 0], -0.33 * np.log(0.2), rtol=1e-12)
    np.testing.assert_allclose(l[0, 0, 1], 0.33 * (-np.log(q[1]) + np.log(q.sum())), rtol=1e-12)


def test_adam_three_step_trajectory():
    th, m, v = np.float64(1.0), 0.0, 0.0
    gs = [0.5, -0.25, 2.0]
    out = []
    for t, g in enumerate(gs, 1):
        th, m, v = U.adam_update(th, g, m, v, t)
        out.append(th)
    # hand: step 1: m=.05 v=.00025 alpha=5e-5*sqrt(.001)/.1 -> th = 1 - 5e-5*(.5/(.5+~0))
    lr, b1, b2, eps = 5e-5, .9, .999, 1e-8
    th2, m2, v2 = 1.0, 0.0, 0.0
    ref = []
    for t, g in enumerate(gs, 1):
        m2 = b1 * m2 + (1 - b1) * g
        v2 = b2 * v2 + (1 - b2) * g * g
        th2 -= lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t) * m2 / (np.sqrt(v2) + eps)
        ref.append(th2)
    np.testing.assert_allclose(out, ref, rtol=1e-13)
    np.testing.assert_allclose(out[0], 1 - 5e-5 * 0.5 / (0.5 + 1e-8 / np.sqrt(1e-3) * 1.0), rtol=1e-9)


def test_l2_kernel_regulariser_known_answer():
    """regularizers.l2(l) on the 3x3 / 2x2 conv kernels only (unet.py:122-177,189): the gradient grows by exactly
    2*l*W there and nowhere else; the reported term is l * sum W^2."""
    w = small_weights(6)
    rng = np.random.RandomState(8)
    x = rng.randn(2, 4, 4, 2)
    y = rng.randint(0, 3, (2, 4, 4)).astype(np.uint8)
    sw = np.array([1.0, 0.33])
    l2 = 0.25
    r0 = U.train_step(w, x, y, sw, depth=1, dtype=torch.float64)
    r1 = U.train_step(w, x, y, sw, depth=1, dtype=torch.float64, l2_reg=l2)
    assert r0["reg_loss"] is None
    tot = 0.0
    for name in r0["grads"]:
        d = r1["grads"][name] - r0["grads"][name]
        if name.endswith("/kernel") and not name.startswith("conv2d/"):
            np.testing.assert_allclose(d, 2 * l2 * np.asarray(w[name], np.float64), rtol=0, atol=1e-12)
            tot += float((np.asarray(w[name], np.float64) ** 2).sum())
        else:
            assert np.abs(d).max() <= 1e-13, name
    np.testing.assert_allclose(r1["reg_loss"], l2 * tot, rtol=1e-12)


def test_encoder_block_backward_independent_numpy_derivation():
    """An independent BACKWARD of one whole encoder block (conv-ReLU-conv-ReLU-BatchNorm(train)-MaxPool, skip tap
    included; mpunet/models/unet.py:114-134) written out with NumPy loops from the textbook formulas -- conv
    gradients as explicit correlations over the SAME-padded window, the batch-statistics BatchNorm gradient
    dx = g*invstd*(dn - mean(dn) - xhat*mean(dn*xhat)), max-pool gradient routed to the FIRST maximum of each
    window -- against autograd through the oracle's own layer functions. Anchors the part of the oracle the
    reference's tests (and TensorFlow's absence) leave unpinned a second, independently derived way."""
    import torch.nn.functional as F
    rng = np.random.RandomState(11)
    B, H, W, C0, C1 = 2, 6, 4, 2, 3
    x = rng.randn(B, H, W, C0)
    k1 = rng.randn(3, 3, C0, C1) * 0.5; b1 = rng.randn(C1) * 0.2
    k2 = rng.randn(3, 3, C1, C1) * 0.5; b2 = rng.randn(C1) * 0.2
    gam = rng.uniform(0.5, 1.5, C1); gam[0] = -0.7                 # a negative gamma: pooling follows the affine
    bet = rng.randn(C1) * 0.3
    R_skip = rng.randn(B, H, W, C1)                                # cotangents of the two outputs of the block
    R_pool = rng.randn(B, H // 2, W // 2, C1)
    eps = 1e-3

    # ---- oracle side: torch autograd through unet_ref's layer functions ------------------------------------------
    t = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    tx, tk1, tb1, tk2, tb2, tg, tbt = t(x), t(k1), t(b1), t(k2), t(b2), t(gam), t(bet)
    p = {"bn/gamma": tg, "bn/beta": tbt}
    c1 = U._conv(tx.permute(0, 3, 1, 2), tk1, tb1)
    c2 = U._conv(c1, tk2, tb2)
    n = U._bn(c2, p, "bn", True, None)
    pool = F.max_pool2d(n, 2, 2)
    loss = (n.permute(0, 2, 3, 1) * torch.tensor(R_skip)).sum() + (pool.permute(0, 2, 3, 1) * torch.tensor(R_pool)).sum()
    loss.backward()

    # ---- independent side: forward and backward in explicit loops -------------------------------------------------
    def conv_fwd(inp, k, b):
        out = np.zeros(inp.shape[:3] + (k.shape[3],))
        for bb in range(B):
            out[bb] = naive_conv_same(inp[bb], k, b, relu=False)
        return out

    def conv_bwd(inp, k, dz):
        """dz = gradient at the pre-activation; returns d inp, dk, db."""
        dinp, dk = np.zeros_like(inp), np.zeros_like(k)
        for bb in range(B):
            for i in range(H):
                for j in range(W):
                    for a in range(3):
                        for c in range(3):
                            ii, jj = i + a - 1, j + c - 1
                            if 0 <= ii < H and 0 <= jj < W:
                                dk[a, c] += np.outer(inp[bb, ii, jj], dz[bb, i, j])
                                dinp[bb, ii, jj] += k[a, c] @ dz[bb, i, j]
        return dinp, dk, dz.sum((0, 1, 2))

    z1 = conv_fwd(x, k1, b1); a1 = np.maximum(z1, 0)
    z2 = conv_fwd(a1, k2, b2); a2 = np.maximum(z2, 0)
    mean, var = a2.mean((0, 1, 2)), a2.var((0, 1, 2))
    invstd = 1.0 / np.sqrt(var + eps)
    xhat = (a2 - mean) * invstd
    nn_ = xhat * gam + bet
    # max-pool backward: the first maximum in row-major window order receives the gradient
    dn = R_skip.copy()
    for bb in range(B):
        for i in range(H // 2):
            for j in range(W // 2):
                for ch in range(C1):
                    win = nn_[bb, 2 * i:2 * i + 2, 2 * j:2 * j + 2, ch]
                    a, c = np.unravel_index(np.argmax(win), (2, 2))
                    dn[bb, 2 * i + a, 2 * j + c, ch] += R_pool[bb, i, j, ch]
    dgam, dbet = (dn * xhat).sum((0, 1, 2)), dn.sum((0, 1, 2))
    N = B * H * W
    da2 = gam * invstd * (dn - dbet / N - xhat * dgam / N)
    dz2 = da2 * (z2 > 0)
    da1, dk2, db2 = conv_bwd(a1, k2, dz2)
    dz1 = da1 * (z1 > 0)
    dx, dk1, db1 = conv_bwd(x, k1, dz1)

    for name, got, ref in (("gamma", dgam, tg.grad), ("beta", dbet, tbt.grad), ("k2", dk2, tk2.grad), ("b2", db2, tb2.grad),
                           ("k1", dk1, tk1.grad), ("b1", db1, tb1.grad), ("x", dx, tx.grad)):
        np.testing.assert_allclose(got, ref.numpy(), rtol=1e-9, atol=1e-10, err_msg=name)


def test_up_block_backward_independent_numpy_derivation():
    """An independent BACKWARD of one whole up block -- UpSampling2D(2, nearest) -> 2x2 conv with TensorFlow's SAME
    padding for even kernels (0 before, 1 after) + ReLU -> BatchNorm(train) -> concat [skip | up] -> 3x3 conv + ReLU ->
    3x3 conv + ReLU -> BatchNorm(train) (mpunet/models/unet.py:148-180) -- written out with NumPy loops from the
    textbook formulas: the 2x2 conv gradient as explicit correlations over the asymmetrically padded window, the
    gradient of the nearest up-sampling as the sum over each 2x2 block, the concat gradient split into its skip and up
    halves. Checked against autograd through the oracle's own layer functions (the encoder-block KAT
    above covered the contracting path only)."""
    import torch.nn.functional as F
    rng = np.random.RandomState(21)
    B, h, w_, Cp, C = 2, 3, 2, 3, 2                 # low-res input [B, h, w, Cp]; the block works at [2h, 2w] with C filters
    H, W = 2 * h, 2 * w_
    xlow = rng.randn(B, h, w_, Cp)
    skip = rng.randn(B, H, W, C)
    ku = rng.randn(2, 2, Cp, C) * 0.5; bu = rng.randn(C) * 0.2
    k2 = rng.randn(3, 3, 2 * C, C) * 0.4; b2 = rng.randn(C) * 0.2
    k3 = rng.randn(3, 3, C, C) * 0.5; b3 = rng.randn(C) * 0.2
    g1 = rng.uniform(0.5, 1.5, C); g1[0] = -0.6
    be1 = rng.randn(C) * 0.3
    g2 = rng.uniform(0.5, 1.5, C); be2 = rng.randn(C) * 0.3
    R = rng.randn(B, H, W, C)                       # cotangent of the block's output
    eps = 1e-3

    # ---- oracle side -------------------------------------------------------------------------------------------------
    t = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    txl, tsk, tku, tbu, tk2, tb2, tk3, tb3, tg1, tbe1, tg2, tbe2 = (t(a) for a in (xlow, skip, ku, bu, k2, b2, k3, b3, g1, be1, g2, be2))
    p = {"bn1/gamma": tg1, "bn1/beta": tbe1, "bn2/gamma": tg2, "bn2/beta": tbe2}
    up = F.interpolate(txl.permute(0, 3, 1, 2), scale_factor=2, mode="nearest")
    u1 = U._conv(up, tku, tbu)
    n1 = U._bn(u1, p, "bn1", True, None)
    cat = torch.cat([tsk.permute(0, 3, 1, 2), n1], dim=1)        # skip first (unet.py:163)
    c2 = U._conv(cat, tk2, tb2)
    c3 = U._conv(c2, tk3, tb3)
    n2 = U._bn(c3, p, "bn2", True, None)
    (n2.permute(0, 2, 3, 1) * torch.tensor(R)).sum().backward()

    # ---- independent side --------------------------------------------------------------------------------------------
    def conv3_fwd(inp, k, b):
        out = np.zeros(inp.shape[:3] + (k.shape[3],))
        for bb in range(B):
            out[bb] = naive_conv_same(inp[bb], k, b, relu=False)
        return out

    def conv3_bwd(inp, k, dz):
        dinp, dk = np.zeros_like(inp), np.zeros_like(k)
        for bb in range(B):
            for i in range(H):
                for j in range(W):
                    for a in range(3):
                        for c in range(3):
                            ii, jj = i + a - 1, j + c - 1
                            if 0 <= ii < H and 0 <= jj < W:
                                dk[a, c] += np.outer(inp[bb, ii, jj], dz[bb, i, j])
                                dinp[bb, ii, jj] += k[a, c] @ dz[bb, i, j]
        return dinp, dk, dz.sum((0, 1, 2))

    def bn_fwd(a, gam, bet):
        mean, var = a.mean((0, 1, 2)), a.var((0, 1, 2))
        invstd = 1.0 / np.sqrt(var + eps)
        xhat = (a - mean) * invstd
        return xhat * gam + bet, xhat, invstd

    def bn_bwd(dn, xhat, invstd, gam):
        N = dn.shape[0] * dn.shape[1] * dn.shape[2]
        dgam, dbet = (dn * xhat).sum((0, 1, 2)), dn.sum((0, 1, 2))
        return gam * invstd * (dn - dbet / N - xhat * dgam / N), dgam, dbet

    # forward
    xup = np.zeros((B, H, W, Cp))
    for i in range(H):
        for j in range(W):
            xup[:, i, j] = xlow[:, i // 2, j // 2]
    zu = np.zeros((B, H, W, C))                      # 2x2 conv, window rows i..i+1, cols j..j+1, zero beyond the far edge
    for bb in range(B):
        for i in range(H):
            for j in range(W):
                acc = bu.copy()
                for a in range(2):
                    for c in range(2):
                        if i + a < H and j + c < W:
                            acc = acc + xup[bb, i + a, j + c] @ ku[a, c]
                zu[bb, i, j] = acc
    au = np.maximum(zu, 0)
    nn1, xh1, is1 = bn_fwd(au, g1, be1)
    catn = np.concatenate([skip, nn1], -1)
    z2 = conv3_fwd(catn, k2, b2); a2 = np.maximum(z2, 0)
    z3 = conv3_fwd(a2, k3, b3); a3 = np.maximum(z3, 0)
    nn2, xh2, is2 = bn_fwd(a3, g2, be2)
    np.testing.assert_allclose(nn2, n2.detach().permute(0, 2, 3, 1).numpy(), rtol=1e-10, atol=1e-11)

    # backward
    da3, dg2, dbe2 = bn_bwd(R, xh2, is2, g2)
    dz3 = da3 * (z3 > 0)
    da2, dk3, db3 = conv3_bwd(a2, k3, dz3)
    dz2 = da2 * (z2 > 0)
    dcat, dk2, db2 = conv3_bwd(catn, k2, dz2)
    dskip, dn1 = dcat[..., :C], dcat[..., C:]        # the concat gradient: first C channels to the skip, the rest to the up path
    dau, dg1, dbe1 = bn_bwd(dn1, xh1, is1, g1)
    dzu = dau * (zu > 0)
    dku, dxup = np.zeros_like(ku), np.zeros_like(xup)
    for bb in range(B):
        for i in range(H):
            for j in range(W):
                for a in range(2):
                    for c in range(2):
                        if i + a < H and j + c < W:
                            dku[a, c] += np.outer(xup[bb, i + a, j + c], dzu[bb, i, j])
                            dxup[bb, i + a, j + c] += ku[a, c] @ dzu[bb, i, j]
    dbu = dzu.sum((0, 1, 2))
    dxlow = np.zeros_like(xlow)                      # nearest up-sampling: every low-res pixel collects its 2x2 block
    for i in range(H):
        for j in range(W):
            dxlow[:, i // 2, j // 2] += dxup[:, i, j]

    for name, got, ref in (("gamma2", dg2, tg2.grad), ("beta2", dbe2, tbe2.grad), ("k3", dk3, tk3.grad), ("b3", db3, tb3.grad),
                           ("k2", dk2, tk2.grad), ("b2", db2, tb2.grad), ("skip", dskip, tsk.grad), ("gamma1", dg1, tg1.grad),
                           ("beta1", dbe1, tbe1.grad), ("k_up", dku, tku.grad), ("b_up", dbu, tbu.grad), ("x_low", dxlow, txl.grad)):
        np.testing.assert_allclose(got, ref.numpy(), rtol=1e-9, atol=1e-10, err_msg=name)
