"""
Teacher-forced replay of the benchmarked train step ("tighten bf16 parity"): BASELINE configs[1]
(depth 4, 64 filters, 16 bf16 slices of 128x128) runs ONE real forward + backward pass with a launch tap installed
(mpu_unet_set_launch_tap). For every one of its 22 forward, 25 data-gradient and 22 weight-gradient convolution
launches the test takes the launch's OWN inputs out of the workspace, computes the same convolution independently
(torch-CPU, fp64 accumulation) from exactly those inputs, and compares with what the HIP kernel stored.

This pins each launch *in situ* -- the kernel, its schedule, its epilogue, on the tensors of the real step --
without the ~45x error amplification of the train-mode BatchNorm network between launches that forces the
whole-network bf16 tests to noise-floor bounds. Tolerances are those of the per-layer suite (tests/test_gpu_conv.py):
forward / data gradient 1.2e-2 of the tensor max (one bf16 rounding of the stored output is 2^-9 relative), weight and
bias gradient 2e-3 of the tensor max (fp32 sums of exact bf16 products).

Forward and data-gradient references are evaluated on the first CMP images of the batch (a convolution's output for
an image depends on that image only); weight gradients sum over the whole batch and are evaluated on all of it.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
quiet = lambda *a, **k: None
CMP = 2


def _hip():
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        try:
            return C.CDLL(name)
        except OSError:
            pass
    raise RuntimeError("libamdhip64 not found")


def _d2h_bf16(hip, dptr, shape):
    """device bf16 tensor -> float64 numpy (exact)."""
    n = int(np.prod(shape))
    host = np.empty(n, np.uint16)
    rc = hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), C.c_size_t(2 * n), C.c_int(2))
    assert rc == 0, rc
    return (host.astype(np.uint32) << 16).view(np.float32).reshape(shape).astype(np.float64)


def _bf16_round(a):
    return torch.tensor(a, dtype=torch.float32).to(torch.bfloat16).to(torch.float64)


def _layer_forward(mode, x_nhwc, w_hwio, bias):
    """y = conv(x) + b (pre-ReLU) of the reference layer: 3x3 SAME (mode 0) or UpSampling2D(2) + 2x2 SAME with
    TensorFlow's 0/1 padding (mode 1; mpunet/models/unet.py:148-160); NHWC in, NHWC out; fp64."""
    x = x_nhwc.permute(0, 3, 1, 2)
    w = w_hwio.permute(3, 2, 0, 1)
    if mode == 0:
        y = F.conv2d(x, w, bias, padding=1)
    else:
        up = x.repeat_interleave(2, 2).repeat_interleave(2, 3)
        y = F.conv2d(F.pad(up, (0, 1, 0, 1)), w, bias)
    return y.permute(0, 2, 3, 1)


def test_cfg1_bf16_step_every_conv_launch_against_fp64_on_its_own_inputs():
    _replay_conv_launches(16, 128, 1)


def test_default_yaml_network_cf2_every_conv_launch_against_fp64():
    """the same replay on the default project YAML's network (complexity_factor 2: 90 / 181 / 362 / 724 /
    1448 filters, every layer with channel tails; the concat weight gradients run as one job per source), B = 8 of 128 x 128."""
    _replay_conv_launches(8, 128, 2)


def test_configs3_per_gpu_share_every_conv_launch_against_fp64():
    """... and on one configs[3]-shaped step: 4 slices of 256 x 256 (the per-GPU share of the global batch 32 at N = 8)."""
    _replay_conv_launches(4, 256, 1)


def test_split_bf16_step_every_conv_launch_against_fp64():
    """dtype "bf16x3" (round 6): the same replay in the f32-storage / split-bf16-product mode -- forward and data gradients on the
    f32 kernels' split path, weight gradients as ONE bf16 reduction over three hi / lo plane pairs (batch 3 B) through the bf16
    schedules, grouped launches included. Each launch against fp64 on its own f32 inputs: 5e-5 of the tensor maximum (a product
    is good to ~2^-16), 240x tighter than the bf16 replay's bound."""
    _replay_conv_launches(4, 128, 1, dtype="bf16x3", tol_act=5e-5, tol_w=5e-5)


@pytest.mark.parametrize("B,cf", [(16, 1), (8, 2)], ids=["configs1", "default_yaml_cf2"])
def test_split_bf16_benchmarked_and_odd_channel_networks_every_conv_launch_against_fp64(B, cf):
    """The same at the BENCHMARKED shape (16 slices: all input planes in one launch, eleven layers on two stored planes through the
    grouped wgrad_taps launch, bias gradients out of the accumulators) and on the default-YAML network (complexity_factor 2:
    90 / 181 / 362 / 724 / 1448 filters -- channel counts that are no multiples of 64: three stored planes, concat sources as
    separate jobs, BatchNorms outside the folded kernels' shapes)."""
    _replay_conv_launches(B, 128, cf, dtype="bf16x3", tol_act=5e-5, tol_w=5e-5)


def _replay_conv_launches(B, dim, cf, dtype="bf16", tol_act=1.2e-2, tol_w=2e-3):
    from multiplanarunet_amd import _lib
    from multiplanarunet_amd.unet import UNet
    from oracle import unet_ref as U
    hip = _hip()
    K = 3
    w0 = U.init_weights(K, 1, 4, cf, seed=31)
    rng = np.random.RandomState(32)
    for k in w0:
        v = k.split("/")[1]
        if v == "bias":
            w0[k] = rng.uniform(-.1, .1, w0[k].shape).astype(np.float32)
        elif v == "gamma":
            w0[k] = rng.uniform(.5, 1.5, w0[k].shape).astype(np.float32)
            w0[k][0] = -0.8
        elif v == "beta":
            w0[k] = rng.uniform(-.3, .3, w0[k].shape).astype(np.float32)
    x = rng.randn(B, dim, dim, 1).astype(np.float32)
    y = (rng.randint(0, K, (B, dim, dim)) * (rng.rand(B, dim, dim) < 0.5)).astype(np.uint8).reshape(B, -1, 1)
    sw = np.where(np.arange(B) % 3 == 0, 0.33, 1.0).astype(np.float32)
    m = UNet(n_classes=K, dim=dim, n_channels=1, depth=4, complexity_factor=cf, flatten_output=True, dtype=dtype,
             logger=quiet)
    bf16 = dtype == "bf16"
    d2h = _d2h_bf16 if bf16 else _d2h_f32                       # activations / gradients as stored: bf16, or f32 ("bf16x3", "f32")
    m.set_weights_dict(w0)
    params = m.params.cpu().numpy()
    lib = _lib.load()
    torch.set_num_threads(max(1, torch.get_num_threads()))

    fwd_err, dg_err, wg_ref = [], [], []

    def kernel_of(li, kk):
        """bf16-rounded (as the MFMA operand) kernel [k][k][Cin_p][Cout_p] and fp32 bias of the layer."""
        Cin = li.C0 + li.C1 if li.kind != 1 else li.n_cnt_layer
        w = params[li.w_off:li.w_off + kk * kk * Cin * li.Cout].reshape(kk, kk, Cin, li.Cout)
        return (_bf16_round(w) if bf16 else torch.tensor(w, dtype=torch.float64)), \
            torch.tensor(params[li.b_off:li.b_off + li.Cout], dtype=torch.float64)

    class Rec:
        pass

    def on_launch(_user, pinfo):
        li = pinfo.contents
        torch.cuda.synchronize()
        kk = 2 if li.mode == 1 else 3
        r = Rec()
        for f, _t in _lib.LaunchInfo._fields_:
            setattr(r, f, getattr(li, f))
        Hi, Wi = (li.H // 2, li.W // 2) if li.mode == 1 else (li.H, li.W)     # kind 0 / 2: input resolution of the layer
        if li.kind == 0:
            r.n_cnt_layer = li.C0 + li.C1
            xin = d2h(hip, li.in0, (B, Hi, Wi, li.C0))[:CMP]
            if li.C1:
                xin = np.concatenate([xin, d2h(hip, li.in1, (B, Hi, Wi, li.C1))[:CMP]], -1)
            got = d2h(hip, li.out, (B, li.H, li.W, li.Cout))[:CMP]
            w, b = kernel_of(r, kk)
            ref = torch.relu(_layer_forward(li.mode, torch.tensor(xin), w, b)).numpy()
            fwd_err.append((li.conv_index, float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)), ref.shape))
        elif li.kind == 1:
            # data gradient w.r.t. input channels [n_off, n_off + n_cnt) of the layer; li.H/W = resolution of the result
            cin_layer = (params_shapes[li.conv_index])
            r.n_cnt_layer = cin_layer
            Hz, Wz = (2 * li.H, 2 * li.W) if li.mode == 1 else (li.H, li.W)
            dz = torch.tensor(d2h(hip, li.dz, (B, Hz, Wz, li.Cout))[:CMP])
            w, _b = kernel_of(r, kk)
            x0 = torch.zeros(CMP, li.H, li.W, cin_layer, dtype=torch.float64, requires_grad=True)
            _layer_forward(li.mode, x0, w, None).backward(dz)
            ref = x0.grad[..., li.n_off:li.n_off + li.n_cnt].numpy()
            if li.mask:
                ref = ref * (d2h(hip, li.mask, (B, li.H, li.W, li.n_cnt))[:CMP] > 0)
            got = d2h(hip, li.out, (B, li.H, li.W, li.n_cnt))[:CMP]
            dg_err.append((li.conv_index, li.n_off, float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)), ref.shape))
        elif li.kind == 2:
            # weight gradient: reference from the launch's x and dz (whole batch; fp32 mkldnn convolutions per chunk of
            # 4 images, chunk results accumulated in fp64: the chunk error ~1e-6 is far below the 2e-3 bound)
            xin = d2h(hip, li.in0, (B, Hi, Wi, li.C0))
            if li.C1:
                xin = np.concatenate([xin, d2h(hip, li.in1, (B, Hi, Wi, li.C1))], -1)
            dz = d2h(hip, li.dz, (B, li.H, li.W, li.Cout))
            Cin = li.C0 + li.C1
            dW = np.zeros((kk, kk, Cin, li.Cout), np.float64)
            for s in range(0, B, 4):
                wz = torch.zeros(kk, kk, Cin, li.Cout, dtype=torch.float32, requires_grad=True)
                _layer_forward(li.mode, torch.tensor(xin[s:s + 4], dtype=torch.float32), wz, None) \
                    .backward(torch.tensor(dz[s:s + 4], dtype=torch.float32))
                dW += wz.grad.numpy().astype(np.float64)
            wg_ref.append((li.conv_index, li.w_off, li.b_off, dW, dz.sum((0, 1, 2))))

    # input channels (padded) of every conv layer, by creation order (for the sliced data gradients)
    params_shapes = {}
    names = [n for n in m._keras_order() if n.endswith("/kernel")]
    for idx, n in enumerate(names):
        params_shapes[idx] = m._tensors[n][2][2]

    cb = _lib.LAUNCH_TAP_FN(on_launch)
    _lib.call("mpu_unet_set_launch_tap", m._h, C.cast(cb, C.c_void_p), None)
    lib.mpu_schedule_log_enable(1)
    try:
        m.forward_backward(x, y, sw)
        torch.cuda.synchronize()
        nlog = lib.mpu_schedule_log_read(None, 0)
        lbuf = C.create_string_buffer(int(nlog) + 1)
        lib.mpu_schedule_log_read(lbuf, nlog + 1)
    finally:
        lib.mpu_schedule_log_enable(0)
        _lib.call("mpu_unet_set_launch_tap", m._h, None, None)
    g = m.grads.cpu().numpy().astype(np.float64)

    assert len(fwd_err) == 22 and len(dg_err) == 25 and len(wg_ref) == 22, (len(fwd_err), len(dg_err), len(wg_ref))
    worst_f = max(fwd_err, key=lambda t: t[1]); worst_d = max(dg_err, key=lambda t: t[2])
    print("replay: forward launches worst rel-to-max error %.3g (conv %d), data-gradient launches %.3g (conv %d, offset %d)"
          % (worst_f[1], worst_f[0], worst_d[2], worst_d[0], worst_d[1]))
    for ci, e, shp in fwd_err:
        assert e <= tol_act, ("forward", ci, e, shp)
    for ci, off, e, shp in dg_err:
        assert e <= tol_act, ("dgrad", ci, off, e, shp)
    worst_w = 0.0
    for ci, w_off, b_off, dW, db in wg_ref:
        got = g[w_off:w_off + dW.size].reshape(dW.shape)
        e = np.abs(got - dW).max() / (np.abs(dW).max() + 1e-30)
        eb = np.abs(g[b_off:b_off + db.size] - db).max() / (np.abs(db).max() + 1e-30)
        worst_w = max(worst_w, e, eb)
        assert e <= tol_w and eb <= tol_w, ("wgrad", ci, e, eb)
    print("replay: weight / bias gradients of the 22 layers: worst rel-to-max error %.3g" % worst_w)


def _d2h_f32(hip, dptr, shape):
    n = int(np.prod(shape))
    host = np.empty(n, np.float32)
    rc = hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), C.c_size_t(4 * n), C.c_int(2))
    assert rc == 0, rc
    return host.reshape(shape).astype(np.float64)


def _d2h_u8(hip, dptr, shape):
    n = int(np.prod(shape))
    host = np.empty(n, np.uint8)
    rc = hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), C.c_size_t(n), C.c_int(2))
    assert rc == 0, rc
    return host.reshape(shape)


def test_cfg1_bf16_step_every_non_conv_launch_against_fp64_on_its_own_inputs():
    """the launch tap also reports the BatchNormalization forward / backward launches, the max-pool
    backward + skip add, the 1x1 head forward and the head backward of the benchmarked bf16 step; each is recomputed in fp64
    (NumPy / torch-CPU) from exactly the tensors the launch read and compared with what it stored. Tolerances: bf16 tensors
    1.2e-2 of the tensor max (one rounding of the stored value is 2^-9 relative; BatchNorm-backward outputs are differences
    of O(1) terms), fp32 reductions (batch statistics, dgamma / dbeta, head weight / bias gradients) 2e-3 of the tensor max,
    probabilities 2e-5 absolute."""
    from multiplanarunet_amd import _lib
    from multiplanarunet_amd.unet import UNet
    from oracle import unet_ref as U
    hip = _hip()
    B, dim, K = 16, 128, 3
    w0 = U.init_weights(K, 1, 4, 1, seed=41)
    rng = np.random.RandomState(42)
    for k in w0:
        v = k.split("/")[1]
        if v == "bias":
            w0[k] = rng.uniform(-.1, .1, w0[k].shape).astype(np.float32)
        elif v == "gamma":
            w0[k] = rng.uniform(.5, 1.5, w0[k].shape).astype(np.float32)
            w0[k][0] = -0.8                                  # a negative gamma: the pooling must follow the affine
        elif v == "beta":
            w0[k] = rng.uniform(-.3, .3, w0[k].shape).astype(np.float32)
    x = rng.randn(B, dim, dim, 1).astype(np.float32)
    y = (rng.randint(0, K, (B, dim, dim)) * (rng.rand(B, dim, dim) < 0.5)).astype(np.uint8).reshape(B, -1, 1)
    sw = np.where(np.arange(B) % 3 == 0, 0.33, 1.0).astype(np.float32)
    m = UNet(n_classes=K, dim=dim, n_channels=1, depth=4, complexity_factor=1, flatten_output=True, dtype="bf16",
             logger=quiet)
    m.set_weights_dict(w0)
    params = m.params.cpu().numpy().astype(np.float64)
    EPS = 1e-3
    rel = lambda got, ref: float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))
    seen = {3: 0, 4: 0, 5: 0, 6: 0, 7: 0}
    worst = {3: 0.0, 4: 0.0, 5: 0.0, 6: 0.0, 7: 0.0}
    later = []                                             # (kind, w_off, b_off, ref at w_off, ref at b_off): gradients read after the pass

    def on_launch(_user, pinfo):
        li = pinfo.contents
        if li.kind < 3:
            return
        torch.cuda.synchronize()
        seen[li.kind] += 1
        H, W, Cc = li.H, li.W, li.C0
        if li.kind == 3:
            xx = _d2h_bf16(hip, li.in0, (B, H, W, Cc))
            got = _d2h_bf16(hip, li.out, (B, H, W, Cc))
            mean_d, inv_d = _d2h_f32(hip, li.aux0, (Cc,)), _d2h_f32(hip, li.aux1, (Cc,))
            g, bt = params[li.w_off:li.w_off + Cc], params[li.b_off:li.b_off + Cc]
            mu = xx.mean((0, 1, 2)); var = xx.var((0, 1, 2))
            inv = 1.0 / np.sqrt(var + EPS)
            e_stat = max(rel(mean_d, mu), rel(inv_d, inv))
            assert e_stat <= 1e-4, ("bn fwd statistics", li.conv_index, e_stat)
            ref = (xx - mu) * inv * g + bt
            e = rel(got, ref)
            assert e <= 1.2e-2, ("bn fwd", li.conv_index, e)
            worst[3] = max(worst[3], e)
            if li.mask:                                      # pooled = 2x2 max of the STORED (rounded) output, exactly
                pg = _d2h_bf16(hip, li.mask, (B, H // 2, W // 2, Cc))
                pr = got.reshape(B, H // 2, 2, W // 2, 2, Cc).max((2, 4))
                assert np.array_equal(pg, pr), ("bn fwd pooled", li.conv_index)
        elif li.kind == 4:
            dn = _d2h_bf16(hip, li.in0, (B, H, W, Cc)); xx = _d2h_bf16(hip, li.in1, (B, H, W, Cc))
            got = _d2h_bf16(hip, li.out, (B, H, W, Cc))
            mean_d, inv_d = _d2h_f32(hip, li.aux0, (Cc,)), _d2h_f32(hip, li.aux1, (Cc,))
            g = params[li.w_off:li.w_off + Cc]
            xh = (xx - mean_d) * inv_d
            M = float(B * H * W)
            dbeta = dn.sum((0, 1, 2)); dgamma = (dn * xh).sum((0, 1, 2))
            dxx = g * inv_d * (dn - dbeta / M - xh * dgamma / M)
            ref = np.where(xx > 0, dxx, 0.0)
            e = rel(got, ref)
            assert e <= 1.2e-2, ("bn bwd dz", li.conv_index, e)
            worst[4] = max(worst[4], e)
            later.append((4, li.w_off, li.b_off, dgamma, dbeta))
        elif li.kind == 5:
            n_ = _d2h_bf16(hip, li.in0, (B, H, W, Cc)); ds = _d2h_bf16(hip, li.in1, (B, H, W, Cc))
            dp = _d2h_bf16(hip, li.dz, (B, H // 2, W // 2, Cc))
            got = _d2h_bf16(hip, li.out, (B, H, W, Cc))
            win = n_.reshape(B, H // 2, 2, W // 2, 2, Cc).transpose(0, 1, 3, 5, 2, 4).reshape(B, H // 2, W // 2, Cc, 4)
            first = win.argmax(-1)                           # np.argmax: the FIRST maximum, row-major inside the window
            routed = np.zeros_like(win)
            np.put_along_axis(routed, first[..., None], dp[..., None], -1)
            routed = routed.reshape(B, H // 2, W // 2, Cc, 2, 2).transpose(0, 1, 4, 2, 5, 3).reshape(B, H, W, Cc)
            ref = ds + routed
            e = float((np.abs(got - ref) / (np.abs(ref) + 1e-6 * np.abs(ref).max())).max())
            assert e <= 2.0 ** -8, ("maxpool bwd + skip", li.conv_index, e)        # one bf16 rounding of an exact sum
            worst[5] = max(worst[5], e)
        elif li.kind == 6:
            n_ = _d2h_bf16(hip, li.in0, (B, H, W, Cc))
            got = _d2h_f32(hip, li.out, (B, H, W, li.Cout))
            Wh = params[li.w_off:li.w_off + Cc * li.Cout].reshape(Cc, li.Cout); bh = params[li.b_off:li.b_off + li.Cout]
            z = n_ @ Wh + bh
            ez = np.exp(z - z.max(-1, keepdims=True))
            ref = ez / ez.sum(-1, keepdims=True)
            e = float(np.abs(got - ref).max())
            assert e <= 2e-5, ("head fwd", e)
            worst[6] = max(worst[6], e)
        elif li.kind == 7:
            n_ = torch.tensor(_d2h_bf16(hip, li.in0, (B, H, W, Cc)), requires_grad=True)
            yy = torch.tensor(_d2h_u8(hip, li.dz, (B, H, W)).astype(np.int64))
            ww = torch.tensor(_d2h_f32(hip, li.mask, (B,)) if li.mask else np.ones(B))
            Wh = torch.tensor(params[li.w_off:li.w_off + Cc * li.Cout].reshape(Cc, li.Cout), requires_grad=True)
            bh = torch.tensor(params[li.b_off:li.b_off + li.Cout], requires_grad=True)
            probs = torch.softmax(n_ @ Wh + bh, -1)
            U.keras_sparse_ce(probs, yy, ww).sum().backward()
            got = _d2h_bf16(hip, li.out, (B, H, W, Cc))
            e = rel(got, n_.grad.numpy())
            assert e <= 1.2e-2, ("head bwd dn", e)
            worst[7] = max(worst[7], e)
            later.append((7, li.w_off, li.b_off, Wh.grad.numpy().ravel(), bh.grad.numpy()))

    cb = _lib.LAUNCH_TAP_FN(on_launch)
    _lib.call("mpu_unet_set_launch_tap", m._h, C.cast(cb, C.c_void_p), None)
    try:
        m.forward_backward(x, y, sw)
        torch.cuda.synchronize()
    finally:
        _lib.call("mpu_unet_set_launch_tap", m._h, None, None)
    g = m.grads.cpu().numpy().astype(np.float64)
    assert seen == {3: 13, 4: 13, 5: 4, 6: 1, 7: 1}, seen
    worst_red = 0.0
    for kind, w_off, b_off, rw, rb in later:
        ew, eb = rel(g[w_off:w_off + rw.size], rw), rel(g[b_off:b_off + rb.size], rb)
        worst_red = max(worst_red, ew, eb)
        assert ew <= 2e-3 and eb <= 2e-3, (kind, w_off, ew, eb)
    print("replay (non-conv): bn fwd %.3g, bn bwd %.3g, pool bwd %.3g, head fwd %.3g (abs), head bwd %.3g; fp32 reductions %.3g"
          % (worst[3], worst[4], worst[5], worst[6], worst[7], worst_red))
