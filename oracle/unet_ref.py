"""
ORACLE -- TEST INFRASTRUCTURE ONLY. Never imported by the product path.

CPU restatement (torch-CPU autograd, fp32 or fp64) of the reference's 2-D U-Net
train / predict step:

  topology, layer order, names ........ mpunet/models/unet.py:114-216
  Keras layer defaults ................ SURVEY.md section 8a row a6 (TF 2.3 API)
  loss / optimizer constants .......... mpunet/bin/defaults/MultiPlanar/train_hparams.yaml:108-126
  compile (reduction=NONE) ............ mpunet/train/trainer.py:78-97, mpunet/bin/train.py:357
  fusion layer ........................ mpunet/models/fusion_model.py:38-39

PARITY UNPINNED for this file: the arithmetic lives in tensorflow==2.3.2
(requirements.txt:10), which is absent from /root/reference and from this
image, and the reference's own tests hold no vectors for it. The
restatement is anchored on hand-computable known-answer tests
(tests/test_oracle_unet_kat.py).

Weights are a dict keyed "<keras layer name>/<var>" with Keras layouts:
conv kernel HWIO (kh,kw,Cin,Cout), bias (Cout,), BN gamma/beta/moving_mean/
moving_variance (C,). The unnamed 1x1 head is Keras' auto-name "conv2d".
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3          # keras BatchNormalization default epsilon
BN_MOMENTUM = 0.99     # keras default momentum
CE_EPS = 1e-7          # keras.backend.epsilon()


def filters_at(level, cf):
    """int(64 * 2**level * sqrt(cf)); unet.py:91,120,132,195."""
    return int(64 * (2 ** level) * np.sqrt(cf))


def layer_specs(n_classes, n_channels=1, depth=4, complexity_factor=1):
    """Ordered (name, kind, shape) list in unet.py creation order."""
    specs = []
    cin = n_channels
    for i in range(depth):
        f = filters_at(i, complexity_factor)
        specs.append(("encoder_L%d_conv1" % i, "conv", (3, 3, cin, f)))
        specs.append(("encoder_L%d_conv2" % i, "conv", (3, 3, f, f)))
        specs.append(("encoder_L%d_BN" % i, "bn", (f,)))
        cin = f
    f = filters_at(depth, complexity_factor)
    specs.append(("bottom_conv1", "conv", (3, 3, cin, f)))
    specs.append(("bottom_conv2", "conv", (3, 3, f, f)))
    specs.append(("bottom_BN", "bn", (f,)))
    cin = f
    for i in range(depth):
        f = filters_at(depth - 1 - i, complexity_factor)
        specs.append(("upsample_L%d_conv1" % i, "conv", (2, 2, cin, f)))
        specs.append(("upsample_L%d_BN1" % i, "bn", (f,)))
        specs.append(("upsample_L%d_conv2" % i, "conv", (3, 3, 2 * f, f)))
        specs.append(("upsample_L%d_conv3" % i, "conv", (3, 3, f, f)))
        specs.append(("upsample_L%d_BN2" % i, "bn", (f,)))
        cin = f
    specs.append(("conv2d", "conv", (1, 1, cin, n_classes)))
    return specs


def init_weights(n_classes, n_channels=1, depth=4, complexity_factor=1, seed=0):
    """Keras defaults: glorot_uniform kernels, zero bias, BN 1/0/0/1."""
    rng = np.random.RandomState(seed)
    w = {}
    for name, kind, shp in layer_specs(n_classes, n_channels, depth,
                                       complexity_factor):
        if kind == "conv":
            kh, kw, ci, co = shp
            lim = np.sqrt(6.0 / (kh * kw * ci + kh * kw * co))
            w[name + "/kernel"] = rng.uniform(-lim, lim, shp).astype(np.float32)
            w[name + "/bias"] = np.zeros(co, np.float32)
        else:
            c = shp[0]
            w[name + "/gamma"] = np.ones(c, np.float32)
            w[name + "/beta"] = np.zeros(c, np.float32)
            w[name + "/moving_mean"] = np.zeros(c, np.float32)
            w[name + "/moving_variance"] = np.ones(c, np.float32)
    return w


def trainable_names(w):
    return [k for k in w if not k.split("/")[1].startswith("moving")]


def _conv(x, k, b, relu=True):
    """Keras Conv2D(padding='same'), x NCHW, k HWIO. 2x2 SAME pads 0 top/left, 1 bottom/right."""
    kh, kw = k.shape[0], k.shape[1]
    wt = k.permute(3, 2, 0, 1)
    pt, pl = (kh - 1) // 2, (kw - 1) // 2
    pb, pr = kh - 1 - pt, kw - 1 - pl
    x = F.pad(x, (pl, pr, pt, pb))
    y = F.conv2d(x, wt, b)
    return torch.relu(y) if relu else y


def _bn(x, p, name, training, new_stats):
    g, b = p[name + "/gamma"], p[name + "/beta"]
    if training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        n = x.shape[0] * x.shape[2] * x.shape[3]
        if new_stats is not None:
            mm, mv = p[name + "/moving_mean"], p[name + "/moving_variance"]
            ub = var.detach() * (n / max(n - 1, 1))
            new_stats[name + "/moving_mean"] = \
                mm * BN_MOMENTUM + mean.detach() * (1 - BN_MOMENTUM)
            new_stats[name + "/moving_variance"] = \
                mv * BN_MOMENTUM + ub * (1 - BN_MOMENTUM)
    else:
        mean, var = p[name + "/moving_mean"], p[name + "/moving_variance"]
    inv = torch.rsqrt(var + BN_EPS) * g
    return x * inv[None, :, None, None] + (b - mean * inv)[None, :, None, None]


def forward(p, x_nhwc, depth=4, training=False, out_activation="softmax",
            new_stats=None, taps=None, store=None, fold_bn=False):
    """U-Net forward on torch params p (same keys as init_weights). Returns NHWC.
    store: optional callable applied wherever the HIP path writes an activation to memory (network input, every
    conv+ReLU output, every BatchNorm output, the pooled tensor, both concat inputs) -- identity for the
    reference arithmetic; bf16_matched_step passes a round-to-bf16 node to model the bf16 storage mode.
    fold_bn: the inference path applies a BatchNorm in the epilogue of the conv in front of it, so that conv's
    output is never stored on its own (one rounding after the affine instead of two)."""
    st = store if store is not None else (lambda t: t)
    sf = (lambda t: t) if fold_bn else st                  # output of a conv that is followed by a BatchNorm
    x = st(x_nhwc.permute(0, 3, 1, 2))
    skips = []
    for i in range(depth):
        n = "encoder_L%d" % i
        x = st(_conv(x, p[n + "_conv1/kernel"], p[n + "_conv1/bias"]))
        x = sf(_conv(x, p[n + "_conv2/kernel"], p[n + "_conv2/bias"]))
        x = st(_bn(x, p, n + "_BN", training, new_stats))
        skips.append(x)
        x = st(F.max_pool2d(x, 2, 2))
    x = st(_conv(x, p["bottom_conv1/kernel"], p["bottom_conv1/bias"]))
    x = sf(_conv(x, p["bottom_conv2/kernel"], p["bottom_conv2/bias"]))
    x = st(_bn(x, p, "bottom_BN", training, new_stats))
    for i in range(depth):
        n = "upsample_L%d" % i
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        x = sf(_conv(x, p[n + "_conv1/kernel"], p[n + "_conv1/bias"]))
        x = st(_bn(x, p, n + "_BN1", training, new_stats))
        x = torch.cat([st(skips[depth - 1 - i]), st(x)], dim=1)      # skip first
        x = st(_conv(x, p[n + "_conv2/kernel"], p[n + "_conv2/bias"]))
        x = sf(_conv(x, p[n + "_conv3/kernel"], p[n + "_conv3/bias"]))
        x = st(_bn(x, p, n + "_BN2", training, new_stats))
        if taps is not None:
            taps[n + "_BN2"] = x.permute(0, 2, 3, 1)
    z = _conv(x, p["conv2d/kernel"], p["conv2d/bias"], relu=False)
    z = z.permute(0, 2, 3, 1)
    if taps is not None:
        taps["logits"] = z
    if out_activation == "softmax":
        return torch.softmax(z, dim=-1)
    return z


def to_torch(w, dtype=torch.float32, requires_grad=False):
    out = {}
    for k, v in w.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        if requires_grad and not k.split("/")[1].startswith("moving"):
            t.requires_grad_(True)
        out[k] = t
    return out


def predict(w, x, depth=4, dtype=torch.float32, out_activation="softmax"):
    with torch.no_grad():
        p = to_torch(w, dtype)
        y = forward(p, torch.tensor(x, dtype=dtype), depth, False, out_activation)
    return y.numpy().astype(np.float32)


def keras_sparse_ce(probs, y, sample_w):
    """
    SparseCategoricalCrossentropy(reduction=NONE) on a softmax output that went
    through a Reshape (train.py:288 forces flatten_output) -- Keras cannot
    back-track to the logits, so it clips the probabilities to [eps, 1-eps],
    takes the log and feeds that to sparse_softmax_cross_entropy_with_logits
    (TF 2.3 keras/backend.py sparse_categorical_crossentropy). The per-pixel
    loss is multiplied by the per-image sample weight; the tape differentiates
    the SUM of the unreduced loss. probs [B,H,W,K]; y [B,H,W] int; w [B].
    """
    q = torch.clamp(probs, CE_EPS, 1 - CE_EPS)
    logq = torch.log(q)
    K = probs.shape[-1]
    l = F.cross_entropy(logq.reshape(-1, K), y.reshape(-1).long(),
                        reduction="none").reshape(y.shape)
    return l * sample_w.reshape(-1, 1, 1).to(l.dtype)


def adam_update(theta, g, m, v, t, lr=5e-5, b1=0.9, b2=0.999, eps=1e-8):
    """
    TF ApplyAdam kernel form (Keras Adam, non-amsgrad), t = 1-based step:
      alpha = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)(1-b1); v += (g^2-v)(1-b2);
      theta -= m*alpha/(sqrt(v)+eps)
    YAML :126 -> lr 5e-5, b1 .9, b2 .999, eps 1e-8, decay 0.
    """
    alpha = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    m = m + (g - m) * (1 - b1)
    v = v + (g * g - v) * (1 - b2)
    theta = theta - (m * alpha) / (np.sqrt(v) + eps)
    return theta, m, v


def train_step(w, x, y, sample_w, opt=None, depth=4, dtype=torch.float32,
               lr=5e-5, b1=0.9, b2=0.999, eps=1e-8, grad_scale_replicas=None, l2_reg=None):
    """
    One Keras Model.fit inner step. x [B,H,W,C] f32,
    y [B,H,W] (or [B,H*W,1]) u8, sample_w [B]. Returns dict(loss [B,H,W],
    grads {name: np}, weights (updated, incl. BN moving stats), opt state).
    l2_reg: regularizers.l2(l2_reg) on every 3x3 / 2x2 conv kernel -- not on the 1x1 output conv, biases or
    BatchNorm (unet.py:122-177,189,211): Keras adds l2_reg * sum(W^2) per kernel to the differentiated total.
    """
    p = to_torch(w, dtype, requires_grad=True)
    B, H, W = x.shape[:3]
    yt = torch.tensor(np.asarray(y).reshape(B, H, W).astype(np.int64))
    new_stats = {}
    probs = forward(p, torch.tensor(x, dtype=dtype), depth, True, "softmax",
                    new_stats)
    loss = keras_sparse_ce(probs, yt, torch.tensor(np.asarray(sample_w), dtype=dtype))
    total = loss.sum()
    reg = None
    if l2_reg:
        reg = sum((p[k] ** 2).sum() for k in p if k.endswith("/kernel") and not k.startswith("conv2d/")) * l2_reg
        total = total + reg
    total.backward()
    names = trainable_names(w)
    grads = {k: p[k].grad.numpy().astype(np.float64 if dtype == torch.float64
                                         else np.float32) for k in names}
    if opt is None:
        opt = {"t": 0, "m": {k: np.zeros_like(grads[k]) for k in names},
               "v": {k: np.zeros_like(grads[k]) for k in names}}
    t = opt["t"] + 1
    new_w = dict(w)
    new_opt = {"t": t, "m": {}, "v": {}}
    for k in names:
        th, m, v = adam_update(np.asarray(w[k], grads[k].dtype), grads[k],
                               opt["m"][k], opt["v"][k], t, lr, b1, b2, eps)
        new_w[k] = th.astype(np.float32)
        new_opt["m"][k], new_opt["v"][k] = m, v
    for k, v in new_stats.items():
        new_w[k] = v.numpy().astype(np.float32)
    return {"loss": loss.detach().numpy(), "probs": probs.detach().numpy(),
            "grads": grads, "weights": new_w, "opt": new_opt,
            "reg_loss": None if reg is None else float(reg.detach())}


def bf16_autograd_grads(w, x, y, sample_w, depth=4):
    """
    Yardstick for the bf16 storage mode, NOT a parity target: the same graph
    differentiated by torch-CPU autograd with every tensor held in bfloat16.
    Gradients of deep layers of a batch-norm U-Net amplify rounding noise by
    ~1e4 (f32 already shows 1e-3 relative error against f64), so a bf16 pipeline
    can only be held to what another bf16 pipeline achieves against f64.
    """
    p = to_torch(w, torch.bfloat16, requires_grad=True)
    B, H, W = x.shape[:3]
    yt = torch.tensor(np.asarray(y).reshape(B, H, W).astype(np.int64))
    probs = forward(p, torch.tensor(x).to(torch.bfloat16), depth, True, "softmax", {})
    loss = keras_sparse_ce(probs.float(), yt, torch.tensor(np.asarray(sample_w), dtype=torch.float32))
    loss.sum().backward()
    return {k: p[k].grad.float().numpy() for k in trainable_names(w)}


def _bf16_operands(w, dtype, requires_grad):
    """fp32 (or `dtype`) master parameters with the 3x3 / 2x2 conv kernels rounded to bf16 as MFMA operands
    (straight-through: gradients land on the master weight)."""
    p = to_torch(w, dtype, requires_grad=requires_grad)
    q = {}
    for k, t in p.items():
        if k.endswith("/kernel") and not k.startswith("conv2d/"):
            q[k] = t + (t.detach().to(torch.float32).to(torch.bfloat16).to(dtype) - t.detach())
        else:
            q[k] = t
    return p, q


def _store_bf16(dtype):
    class _S(torch.autograd.Function):
        """A tensor written to memory as bf16 (round-to-nearest-even) in the forward pass AND its gradient written
        as bf16 in the backward pass; arithmetic around it stays in `dtype` (the MFMA accumulates in fp32)."""

        @staticmethod
        def forward(ctx, t):
            return t.to(torch.float32).to(torch.bfloat16).to(dtype)

        @staticmethod
        def backward(ctx, g):
            return g.to(torch.float32).to(torch.bfloat16).to(dtype)
    return _S.apply


def bf16_matched_forward(w, x, depth=4, out_activation="linear", dtype=torch.float32):
    """Inference-mode model of the bf16 storage mode (BatchNorm folded into the producing conv's epilogue)."""
    with torch.no_grad():
        _, q = _bf16_operands(w, dtype, False)
        y = forward(q, torch.tensor(x, dtype=dtype), depth, False, out_activation, store=_store_bf16(dtype), fold_bn=True)
    return y.numpy().astype(np.float32)


def bf16_matched_step(w, x, y, sample_w, depth=4, dtype=torch.float32):
    """
    Model of the bf16 STORAGE mode of the HIP path (dtype="bf16"), NOT of the reference: `dtype` arithmetic with a
    round-to-bf16 at exactly the points where the kernels store a tensor -- activations and their gradients
    (forward(store=...)), and the 3x3 / 2x2 conv kernels as MFMA operands. Biases, BatchNorm parameters and the
    1x1 head stay fp32, as in the kernels.

    What this can and cannot pin: a train-mode depth-4 BatchNorm U-Net amplifies a relative perturbation of its
    input ~45x at initialisation (measured), so the occasional bf16 rounding that flips under a different fp32
    summation order (probability ~3e-5 per element) already moves the probabilities by ~1e-2 and the deep-layer
    gradients by tens of percent: two evaluations of THIS model that differ only in dtype (f32 / f64 arithmetic)
    are that far apart. The tests therefore use the distance between those two evaluations as the noise floor of
    the storage mode and hold the kernels to a small multiple of it, tensor by tensor; the head-side tensors,
    which the chaos does not reach, are held to tight absolute bounds.
    Returns dict(probs, loss, grads).
    """
    p, q = _bf16_operands(w, dtype, True)
    B, H, W = x.shape[:3]
    yt = torch.tensor(np.asarray(y).reshape(B, H, W).astype(np.int64))
    probs = forward(q, torch.tensor(x, dtype=dtype), depth, True, "softmax", {}, store=_store_bf16(dtype))
    loss = keras_sparse_ce(probs, yt, torch.tensor(np.asarray(sample_w), dtype=dtype))
    loss.sum().backward()
    return {"probs": probs.detach().numpy().astype(np.float32), "loss": loss.detach().numpy().astype(np.float32),
            "grads": {k: p[k].grad.numpy().astype(np.float64) for k in trainable_names(w)}}
