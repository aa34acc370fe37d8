"""
mpunet.models.UNet on MI355X: same constructor, attributes and Keras-Model call
surface as the reference (mpunet/models/unet.py:20-251), with every tensor
operation executed by libmpunet_hip.so (hand-written gfx950 kernels).

Host code here only owns buffers (torch device tensors), converts weights
between Keras layouts and the library's flat parameter buffer, and drives the
C ABI: mpu_unet_forward / mpu_unet_backward / mpu_adam_step /
mpu_unet_pack_weights. There is no eager / CPU fallback.
"""
import ctypes as C
import os
import numpy as np
import torch

from . import _lib


class _ScreenLogger:
    def __call__(self, *args, **kwargs):
        print(*args)


class _OutputLayerShim:
    """What mpunet/utils/utils.py:190-241 (set_bias_weights) needs from the last layer."""

    class _Act:
        def __init__(self, name):
            self.__name__ = name

    def __init__(self, model):
        self._m = model
        self.activation = self._Act(model.out_activation)
        self.name = "conv2d"

    def get_weights(self):
        d = self._m.get_weights_dict()
        return [d["conv2d/kernel"], d["conv2d/bias"]]

    def set_weights(self, weights):
        self._m.set_weights_dict({"conv2d/kernel": weights[0], "conv2d/bias": weights[1]})


class UNet:
    """
    2D UNet implementation with batch normalization and complexity factor adj.
    See mpunet/models/unet.py:26-79 for the meaning of the arguments. Extra
    keyword arguments (model_class_name, l1_reg, biased_output_layer, ...) are
    accepted and ignored, as the reference does (unet.py:41).

    Additions that have no counterpart in the reference:
      dtype  : "bf16" (default; bf16 storage + MFMA, f32 accumulate), "f32" (exact-f32 MFMA: the parity mode) or
               "bf16x3" (f32 storage, split-bf16 products: logits within the north-star tolerance at ~3x the f32 mode's speed)
               (exact-f32 MFMA; the parity mode)
      device : torch device of all buffers
      seed   : seed of the glorot-uniform initialisation
    """

    def __init__(self, n_classes, img_rows=None, img_cols=None, dim=None, n_channels=1, depth=4,
                 out_activation="softmax", activation="relu", kernel_size=3, padding="same",
                 complexity_factor=1, flatten_output=False, l2_reg=None, logger=None,
                 dtype="bf16", device="cuda", seed=None, **kwargs):
        if not ((img_rows and img_cols) or dim):
            raise ValueError("Must specify either img_rows and img_col or dim")
        if dim:
            img_rows, img_cols = dim, dim
        self.logger = logger or _ScreenLogger()
        self.img_shape = (img_rows, img_cols, n_channels)
        self.n_classes = n_classes
        self.cf = np.sqrt(complexity_factor)
        self.kernel_size = kernel_size
        self.activation = activation
        self.out_activation = out_activation
        self.l2_reg = l2_reg
        self.padding = padding
        self.depth = depth
        self.flatten_output = flatten_output
        self.label_crop = np.array([[0, 0], [0, 0]])
        self.stop_training = False
        if padding != "same" or activation != "relu" or kernel_size != 3:
            raise NotImplementedError("only padding='same', activation='relu', kernel_size=3 "
                                      "are on the MI355X path")
        if out_activation not in ("softmax", "linear"):
            raise NotImplementedError("out_activation must be 'softmax' or 'linear'")
        if img_rows % (2 ** depth) or img_cols % (2 ** depth):
            raise NotImplementedError("image dims must be multiples of 2**depth (no cropping path)")
        # "bf16x3" (round 6): f32 storage, every matrix product as three bf16 MFMAs on hi + lo split operands (mpu_dtype
        # MPU_F32X3) -- the tolerance-grade mode at bf16-class matrix rates; everything outside the MFMA loops is the f32 mode
        self.split_bf16 = isinstance(dtype, str) and dtype in ("bf16x3", "f32x3")
        if self.split_bf16:
            dtype = "f32"
        self.dtype = {"bf16": torch.bfloat16, "f32": torch.float32, "float32": torch.float32,
                      "bfloat16": torch.bfloat16}[dtype] if isinstance(dtype, str) else dtype
        self.device = torch.device(device)

        cfg = _lib.UNetConfig()
        cfg.n_classes, cfg.n_channels, cfg.depth = n_classes, n_channels, depth
        cfg.H, cfg.W = img_rows, img_cols
        cfg.dtype = _lib.MPU_BF16 if self.dtype == torch.bfloat16 else (_lib.MPU_F32X3 if self.split_bf16 else _lib.MPU_F32)
        cfg.softmax = 1 if out_activation == "softmax" else 0
        self.filters = [int(64 * (2 ** l) * self.cf) for l in range(depth + 1)]      # unet.py:91,120
        for l, f in enumerate(self.filters):
            cfg.filters[l] = f
        lib = _lib.load()
        self._h = lib.mpu_unet_create(C.byref(cfg))
        if not self._h:
            raise ValueError("mpu_unet_create: " + (lib.mpu_last_error() or b"").decode())
        self._h = C.c_void_p(self._h)

        # tensor table
        self._tensors = {}
        self._order = []
        name = C.create_string_buffer(96)
        kind, off = C.c_int32(), C.c_int64()
        ps, ls = (C.c_int32 * 4)(), (C.c_int32 * 4)()
        for i in range(lib.mpu_unet_num_tensors(self._h)):
            _lib.check(lib.mpu_unet_tensor_info(self._h, i, name, 96, C.byref(kind), C.byref(off), ps, ls),
                       "mpu_unet_tensor_info")
            nm = name.value.decode()
            self._tensors[nm] = (kind.value, off.value, tuple(v for v in ps if v), tuple(v for v in ls if v))
            self._order.append(nm)
        n_par = lib.mpu_unet_param_floats(self._h)
        n_st = lib.mpu_unet_bn_state_floats(self._h)
        dev = self.device
        self.params = torch.zeros(n_par, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(n_par, dtype=torch.float32, device=dev)
        self.bn_state = torch.zeros(n_st, dtype=torch.float32, device=dev)
        self.packed = torch.zeros(lib.mpu_unet_packed_bytes(self._h), dtype=torch.uint8, device=dev)
        self._adam_m = self._adam_v = None
        self._ws = None
        self._ws_batch = 0
        self.optimizer_kwargs = dict(lr=5e-5, beta_1=0.9, beta_2=0.999, epsilon=1e-8)
        self.iterations = 0
        self._l2_ws = None
        self.reg_loss = None
        self._grad_hook = None         # e.g. an all-reduce over RCCL (multiplanarunet_amd.distributed)
        self._init_weights(seed)

        # receptive field of the contracting path (unet.py:104-109, utils/conv_arithmetics.py:57)
        rf, jump = 1, 1
        for _ in range(depth):
            rf += 2 * jump; rf += 2 * jump          # two 3x3 convs
            jump *= 2; rf += jump                   # 2x2 max-pool, stride 2
        rf += 2 * jump; rf += 2 * jump              # bottom convs
        self.receptive_field = np.array([rf, rf])
        self.layers = [_OutputLayerShim(self)]
        self.metrics_names = ["loss"]
        self.log()

    # ------------------------------------------------------------------ #
    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().mpu_unet_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _init_weights(self, seed):
        """Keras defaults: glorot_uniform kernels, zero biases, BN gamma=1 beta=0 mean=0 var=1."""
        rng = np.random.RandomState(seed)
        w = {}
        for nm in self._order:
            kind, off, ps, ls = self._tensors[nm]
            var = nm.split("/")[1]
            if var == "kernel":
                kh, kw, ci, co = ls
                lim = np.sqrt(6.0 / (kh * kw * ci + kh * kw * co))
                w[nm] = rng.uniform(-lim, lim, ls).astype(np.float32)
            elif var in ("gamma", "moving_variance"):
                w[nm] = np.ones(ls, np.float32)
            else:
                w[nm] = np.zeros(ls, np.float32)
        self.set_weights_dict(w)

    # ---- weights ------------------------------------------------------ #
    def _buf(self, kind):
        return self.params if kind == 0 else self.bn_state

    def get_weights_dict(self):
        """{'<layer>/<var>': ndarray} in Keras layouts (kernels HWIO), padding stripped."""
        out = {}
        host = {0: self.params.cpu().numpy(), 1: self.bn_state.cpu().numpy()}
        for nm in self._order:
            kind, off, ps, ls = self._tensors[nm]
            a = host[kind][off:off + int(np.prod(ps))].reshape(ps)
            out[nm] = self._from_stored(nm, a, ps, ls)
        return out

    @staticmethod
    def _is_concat_kernel(nm):
        return nm.startswith("upsample_L") and nm.endswith("_conv2/kernel")

    def _from_stored(self, nm, a, ps, ls):
        """Strip the channel padding. The conv after the skip-concat reads [skip(F_p), up(F_p)]:
        its input-channel padding sits after EACH half (unet.py:168-169 concat order: skip first)."""
        if self._is_concat_kernel(nm):
            F, Fp = ls[3], ps[3]
            return np.ascontiguousarray(np.concatenate([a[:, :, :F, :F], a[:, :, Fp:Fp + F, :F]], axis=2))
        return np.ascontiguousarray(a[tuple(slice(0, s) for s in ls)])

    def _to_stored(self, nm, val, ps, ls):
        a = np.zeros(ps, np.float32)
        if self._is_concat_kernel(nm):
            F, Fp = ls[3], ps[3]
            a[:, :, :F, :F] = val[:, :, :F, :]
            a[:, :, Fp:Fp + F, :F] = val[:, :, F:, :]
        else:
            a[tuple(slice(0, s) for s in ls)] = val
        return a

    def set_weights_dict(self, weights, strict=False):
        host = {0: self.params.cpu().numpy(), 1: self.bn_state.cpu().numpy()}
        for nm, val in weights.items():
            if nm not in self._tensors:
                if strict:
                    raise KeyError(nm)
                continue
            kind, off, ps, ls = self._tensors[nm]
            val = np.asarray(val, np.float32)
            if tuple(val.shape) != tuple(ls):
                raise ValueError("shape mismatch for %s: got %s expected %s" % (nm, val.shape, ls))
            a = self._to_stored(nm, val, ps, ls)
            host[kind][off:off + a.size] = a.ravel()
        self.params.copy_(torch.from_numpy(host[0]))
        self.bn_state.copy_(torch.from_numpy(host[1]))
        self._repack()

    _KERAS_VARS = {"kernel": 0, "bias": 1, "gamma": 0, "beta": 1, "moving_mean": 2, "moving_variance": 3}

    def _keras_order(self):
        """Layer creation order of unet.py (encoder, bottom, upsample, head)."""
        names = []
        for i in range(self.depth):
            names += ["encoder_L%d_conv1" % i, "encoder_L%d_conv2" % i, "encoder_L%d_BN" % i]
        names += ["bottom_conv1", "bottom_conv2", "bottom_BN"]
        for i in range(self.depth):
            p = "upsample_L%d" % i
            names += [p + "_conv1", p + "_BN1", p + "_conv2", p + "_conv3", p + "_BN2"]
        names += ["conv2d"]
        out = []
        for n in names:
            vs = ("gamma", "beta", "moving_mean", "moving_variance") if "BN" in n else ("kernel", "bias")
            out += [n + "/" + v for v in vs]
        return out

    def get_weights(self):
        d = self.get_weights_dict()
        return [d[n] for n in self._keras_order()]

    def set_weights(self, weights):
        names = self._keras_order()
        if len(weights) != len(names):
            raise ValueError("expected %d arrays" % len(names))
        self.set_weights_dict(dict(zip(names, weights)), strict=True)

    def save_weights(self, path):
        """Name-keyed weights: .npz (native; keys "<layer>__<var>") or, with h5py installed, a tf.keras
        `save_weights` .h5 file (mpunet/callbacks/mcp_clean.py:57, multiplanarunet_amd/formats.py)."""
        d = self.get_weights_dict()
        if str(path).endswith((".h5", ".hdf5")):
            from .formats import save_keras_h5
            save_keras_h5(path, d, depth=self.depth)
            return
        with open(path, "wb") as f:
            np.savez(f, **{k.replace("/", "__"): v for k, v in d.items()})

    def load_weights(self, path, by_name=True):
        """load_weights(by_name=True) (mpunet/models/model_init.py:31,56): .npz or a reference Keras .h5 checkpoint.
        The 1x1 head is unnamed in the reference (unet.py:211), so a checkpoint written by a process that had built
        other models calls it conv2d_<N>: a lone 1x1 `conv2d(_N)` layer of the file is mapped onto the head. Model
        tensors the file does not supply are reported (Keras by_name loading skips them silently; here that almost
        always means a naming problem) -- `self.missing_on_load` lists them."""
        import re
        if str(path).endswith((".h5", ".hdf5")):
            from .formats import load_keras_h5
            d = load_keras_h5(path)
        else:
            with np.load(path) as z:
                d = {k.replace("__", "/"): z[k] for k in z.files}
        if "conv2d/kernel" not in d:
            auto = sorted({k.split("/")[0] for k, v in d.items()
                           if re.fullmatch(r"conv2d_\d+/kernel", k) and tuple(np.shape(v)[:2]) == (1, 1)})
            if len(auto) == 1:
                for v in ("kernel", "bias"):
                    if auto[0] + "/" + v in d:
                        d["conv2d/" + v] = d.pop(auto[0] + "/" + v)
        self.missing_on_load = sorted(n for n in self._tensors if n not in d)
        if self.missing_on_load:
            msg = "load_weights(%s): %d model tensors not in the file (kept as they are): %s" % (
                path, len(self.missing_on_load), ", ".join(self.missing_on_load[:6]) +
                (" ..." if len(self.missing_on_load) > 6 else ""))
            if not by_name:
                raise KeyError(msg)
            self.logger(msg)
        self.set_weights_dict(d, strict=not by_name)

    def count_params(self):
        """Keras count_params(): trainable + BN moving statistics, unpadded."""
        return int(sum(int(np.prod(ls)) for (_, _, _, ls) in self._tensors.values()))

    def _repack(self):
        self._infer_dirty = True
        if self.device.type != "cuda":
            return          # host-only introspection (layout, weight I/O); nothing can run without a GPU
        _lib.call("mpu_unet_pack_weights", self._h, _lib.ptr(self.params), _lib.ptr(self.packed),
                  _lib.stream_ptr())

    # ---- execution ---------------------------------------------------- #
    def _workspace(self, batch):
        if self._ws is None or self._ws_batch < batch:
            n = _lib.load().mpu_unet_workspace_bytes(self._h, batch)
            self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._ws_batch = batch
        return self._ws

    def _as_input(self, X):
        if not torch.is_tensor(X):
            X = torch.from_numpy(np.ascontiguousarray(X))
        X = X.to(device=self.device, dtype=torch.float32).contiguous()
        if X.ndim != 4 or tuple(X.shape[1:]) != tuple(self.img_shape):
            raise ValueError("expected input [B,%d,%d,%d], got %s" % (self.img_shape + (tuple(X.shape),)))
        return X

    def _forward(self, X, training, out=None):
        if self.device.type != "cuda":
            raise _lib.MpuError("UNet needs a HIP device (MI355X); there is no CPU execution path")
        B = X.shape[0]
        # the workspace layout depends on the batch it was planned for
        ws = self._workspace_exact(B)
        if training:
            self._infer_dirty = True               # the moving statistics are about to change
        elif getattr(self, "_infer_dirty", True):
            _lib.call("mpu_unet_prepare_inference", self._h, _lib.ptr(self.params), _lib.ptr(self.bn_state),
                      _lib.ptr(self.packed), _lib.stream_ptr())
            self._infer_dirty = False
        shape = (B, self.img_shape[0], self.img_shape[1], self.n_classes)
        if training and out is None:
            # train-mode forward: the probabilities stay in the workspace (the backward pass reads them there);
            # the returned tensor is a VIEW of that region, valid until the next forward
            off = int(_lib.load().mpu_unet_workspace_probs_offset(self._h, B))
            n = 4 * B * shape[1] * shape[2] * shape[3]
            view = ws[off:off + n].view(torch.float32).reshape(shape)
            _lib.call("mpu_unet_forward", self._h, B, _lib.ptr(X), _lib.ptr(self.params), _lib.ptr(self.packed),
                      _lib.ptr(self.bn_state), _lib.ptr(ws), 1, None, _lib.stream_ptr())
            return view
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=self.device)
        _lib.call("mpu_unet_forward", self._h, B, _lib.ptr(X), _lib.ptr(self.params), _lib.ptr(self.packed),
                  _lib.ptr(self.bn_state), _lib.ptr(ws), 1 if training else 0, _lib.ptr(out), _lib.stream_ptr())
        return out

    def _workspace_exact(self, batch):
        # plans for smaller batches fit inside a larger allocation
        return self._workspace(batch)

    def predict_on_batch(self, X):
        out = self._forward(self._as_input(X), training=False)
        return out.reshape(out.shape[0], -1, self.n_classes) if self.flatten_output else out

    def max_batch(self):
        """Largest batch whose activations stay below the kernels' 2 GiB (32-bit offset) operand bound."""
        H, W = self.img_shape[:2]
        f0 = 8 * ((self.filters[0] + 7) // 8)                       # level-0 filters (padded): the largest activations
        per_image = H * W * f0 * self.params_esz()
        return max(1, int((2 ** 31 - 1) // per_image))

    def params_esz(self):
        return 2 if self.dtype == torch.bfloat16 else 4

    def auto_batch(self, n, cap=160):
        """Even chunks of n images, as large as the operand bound (and `cap`) allow: big batches fill the chip at the deep
        levels. Among the smallest chunk count and the next two, the one whose chunk size fills the rounds of the persistent
        predict kernel best (conv_halo16p: one workgroup per CU walks `tiles / workgroups` tiles -- 138 planes of 256x256
        leave its 512-channel level at 4.3 rounds, 92 planes at 2.9: 276 planes -> 3 x 92, measured -1.5 % per volume)."""
        bmax = max(1, min(cap, self.max_batch()))
        cmin = -(-n // bmax)
        if self.dtype != torch.bfloat16:
            return -(-n // cmin)
        best, best_score = cmin, -1.0
        for c in range(cmin, cmin + 3):
            if c > cmin and -(-n // c) < 64:           # small batches lose more at the deep levels than a full round wins
                break                                   # (69 planes per launch: +6 % per volume, measured)
            score = self._round_fill(-(-n // c)) - 0.002 * (c - cmin)
            if score > best_score + 1e-9:
                best, best_score = c, score
        return -(-n // best)

    def _round_fill(self, B):
        """Mean over the levels with 128-channel tiles of (rounds of 16 x 32-pixel tiles per persistent workgroup) /
        ceil(rounds) for a batch of B images; a level too small for that kernel (< 512 tiles, rows not a multiple of 16,
        narrower than 32 pixels) counts 0.9 (two workgroups per CU of the 4-wave kernel, measured flat)."""
        H, W = self.img_shape[:2]
        ncu = 256
        if self.device.type == "cuda":
            ncu = torch.cuda.get_device_properties(self.device).multi_processor_count or 256
        effs = []
        for l in range(1, self.depth + 1):
            f = self.filters[l]
            h, w = H >> l, W >> l
            if f <= 64 or f % 32:
                continue
            if h % 16 or w < 32:
                effs.append(0.9)
                continue
            nt = -(-f // 128)
            tiles = B * (h // 16) * (-(-w // 32))
            if tiles * nt < 512:
                effs.append(0.9)
                continue
            rounds = tiles / max(1, ncu // nt)
            effs.append(rounds / float(-(-tiles // max(1, ncu // nt))))
        return sum(effs) / len(effs) if effs else 1.0

    def predict(self, X, batch_size=8, verbose=0):
        """model.predict: inference-mode forward in chunks of batch_size (None: auto_batch); returns a device tensor."""
        numpy_in = not torch.is_tensor(X)
        n = X.shape[0]
        if batch_size is None:
            batch_size = self.auto_batch(n)
        out = torch.empty((n, self.img_shape[0], self.img_shape[1], self.n_classes),
                          dtype=torch.float32, device=self.device)
        for s in range(0, n, batch_size):
            xb = self._as_input(X[s:s + batch_size])
            self._forward(xb, training=False, out=out[s:s + xb.shape[0]])
        if self.flatten_output:
            out = out.reshape(n, -1, self.n_classes)
        return out.cpu().numpy() if numpy_in else out

    __call__ = predict_on_batch

    # ---- training ------------------------------------------------------ #
    def compile(self, optimizer="Adam", loss=None, metrics=None, optimizer_kwargs=None, **kwargs):
        """
        Trainer.compile_model (mpunet/train/trainer.py:51-97): only the reference
        default is on the path -- Adam + SparseCategoricalCrossentropy(reduction=NONE).
        """
        name = optimizer if isinstance(optimizer, str) else type(optimizer).__name__
        if name.lower() != "adam":
            raise NotImplementedError("only the Adam optimizer is supported")
        if loss is not None:
            lname = loss if isinstance(loss, str) else getattr(loss, "__name__", type(loss).__name__)
            if isinstance(loss, (list, tuple)):
                lname = loss[0] if isinstance(loss[0], str) else type(loss[0]).__name__
            if "sparsecategoricalcrossentropy" not in lname.lower().replace("_", ""):
                raise NotImplementedError("only SparseCategoricalCrossentropy is supported")
        if optimizer_kwargs:
            kw = dict(optimizer_kwargs)
            if "learning_rate" in kw:
                kw["lr"] = kw.pop("learning_rate")
            if kw.get("decay"):
                raise NotImplementedError("Adam decay != 0 is not supported")
            kw.pop("decay", None)
            self.optimizer_kwargs.update(kw)
        return self

    def _ensure_adam(self):
        if self._adam_m is None:
            self._adam_m = torch.zeros_like(self.params)
            self._adam_v = torch.zeros_like(self.params)

    def grad_ready_points(self):
        """Float offsets (descending) of the backward pass's gradient-ready points: after point k every gradient at
        offset >= offsets[k] of the flat buffer is final (head, up blocks, bottom, encoder levels)."""
        import ctypes as C
        buf = (C.c_int64 * 32)()
        n = _lib.load().mpu_unet_grad_ready_points(self._h, buf, 32)
        return [int(buf[i]) for i in range(n)]

    def forward_backward(self, x, y, sample_weight=None, want_loss=True, ready_events=None, adam=None):
        """Train-mode forward + backward; fills self.grads (sum-gradient). Returns (probs, loss[B,H*W] or None).
        ready_events: optional list (one entry per grad_ready_points(), None = skip) of torch.cuda.Event recorded
        on the current stream when that part of the gradient buffer is final (data-parallel overlap).
        adam: (t, step_dev) -- ALSO apply the optimizer inside the same library call (mpu_unet_backward_adam: the
        update of the deep levels' parameters runs beside the weight gradients of the high-resolution levels)."""
        X = self._as_input(x)
        B = X.shape[0]
        if not torch.is_tensor(y):
            y = torch.from_numpy(np.ascontiguousarray(y))
        y = y.to(device=self.device, dtype=torch.uint8).reshape(B, -1).contiguous()
        if y.shape[1] != self.img_shape[0] * self.img_shape[1]:
            raise ValueError("labels must have H*W entries per image")
        if sample_weight is None:
            sw = torch.ones(B, dtype=torch.float32, device=self.device)
        else:
            sw = torch.as_tensor(np.asarray(sample_weight, np.float32) if not torch.is_tensor(sample_weight)
                                 else sample_weight).to(device=self.device, dtype=torch.float32).contiguous()
        probs = self._forward(X, training=True)
        loss = torch.empty((B, y.shape[1]), dtype=torch.float32, device=self.device) if want_loss else None
        if adam is not None:
            if ready_events is not None:
                raise ValueError("forward_backward: adam= and ready_events= exclude each other")
            t, step_dev = adam
            k = self.optimizer_kwargs
            _lib.call("mpu_unet_backward_adam", self._h, B, _lib.ptr(y), _lib.ptr(sw), _lib.ptr(self.params),
                      _lib.ptr(self.packed), _lib.ptr(self.bn_state), _lib.ptr(self._ws), _lib.ptr(self.grads),
                      _lib.ptr(loss), _lib.ptr(self._adam_m), _lib.ptr(self._adam_v), int(t), _lib.ptr(step_dev),
                      float(k["lr"]), float(k["beta_1"]), float(k["beta_2"]), float(k["epsilon"]), _lib.stream_ptr())
            self._infer_dirty = True
        elif ready_events is None:
            _lib.call("mpu_unet_backward", self._h, B, _lib.ptr(y), _lib.ptr(sw), _lib.ptr(self.params),
                      _lib.ptr(self.packed), _lib.ptr(self.bn_state), _lib.ptr(self._ws), _lib.ptr(self.grads),
                      _lib.ptr(loss), _lib.stream_ptr())
        else:
            import ctypes as C
            arr = (C.c_void_p * len(ready_events))(*[None if e is None else e.cuda_event for e in ready_events])
            _lib.call("mpu_unet_backward_events", self._h, B, _lib.ptr(y), _lib.ptr(sw), _lib.ptr(self.params),
                      _lib.ptr(self.packed), _lib.ptr(self.bn_state), _lib.ptr(self._ws), _lib.ptr(self.grads),
                      _lib.ptr(loss), arr, len(ready_events), _lib.stream_ptr())
        self._ready_events_fresh = ready_events is not None      # the DP hook falls back to a full stream wait otherwise
        self._last_batch = B
        return probs, loss

    def loss_mean(self):
        """Device scalar (a [1] f32 VIEW into the workspace, valid until the next backward pass): the mean over all pixels of the
        weighted per-pixel loss of the last forward_backward / train_step -- what Keras logs per batch. Produced by the backward
        pass itself (mpu_unet_workspace_loss_mean_offset), so a training loop needs no reduction of the 1-MB loss tensor; inside a
        captured HIP graph that torch reduction (several blocks per output, semaphores) went stale for stretches of replays."""
        off = int(_lib.load().mpu_unet_workspace_loss_mean_offset(self._h, self._last_batch))
        return self._ws[off:off + 4].view(torch.float32)

    def _add_l2(self, want_loss=False):
        """kernel_regularizer=l2(self.l2_reg) (unet.py:189): grads += 2*l2*W on the 3x3 / 2x2 conv kernels; the
        term l2*sum(W^2) that Keras adds to the reported loss lands in self.reg_loss (device scalar)."""
        if not self.l2_reg:
            return
        if self._l2_ws is None:
            n = int(_lib.load().mpu_unet_l2_workspace_doubles())
            self._l2_ws = torch.empty(n, dtype=torch.float64, device=self.device)
            self.reg_loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        _lib.call("mpu_unet_l2_regularizer", self._h, _lib.ptr(self.params), _lib.ptr(self.grads),
                  float(self.l2_reg), _lib.ptr(self._l2_ws), _lib.ptr(self.reg_loss) if want_loss else None,
                  _lib.stream_ptr())

    def apply_gradients(self, fused=True):
        """Keras Adam on the flat parameter buffer + refresh of the packed MFMA operands: one launch
        (mpu_unet_adam_pack); fused=False runs the two separate passes (mpu_adam_step, mpu_unet_pack_weights;
        bit-identical, kept for the equality test)."""
        self._ensure_adam()
        self.iterations += 1
        k = self.optimizer_kwargs
        if fused:
            _lib.call("mpu_unet_adam_pack", self._h, _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self._adam_m),
                      _lib.ptr(self._adam_v), self.iterations, None, float(k["lr"]), float(k["beta_1"]),
                      float(k["beta_2"]), float(k["epsilon"]), _lib.ptr(self.packed), _lib.stream_ptr())
            self._infer_dirty = True
            return
        _lib.call("mpu_adam_step", _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self._adam_m),
                  _lib.ptr(self._adam_v), self.params.numel(), self.iterations, float(k["lr"]),
                  float(k["beta_1"]), float(k["beta_2"]), float(k["epsilon"]), _lib.stream_ptr())
        self._repack()

    # ---- HIP-graph replay of the whole step (launch-bound regime) ---------------------------------
    def make_graphed_train_step(self, x, y, sample_weight=None, loss_sum=None, warmup=True):
        """
        Capture forward + backward + Adam + repack (~130 kernel launches) into one HIP graph that reads the
        given DEVICE tensors x, y, sample_weight in place; returns replay() which runs one train step per
        call. Single-GPU only (the RCCL all-reduce stays eager). The Adam step count lives on the device.
        loss_sum: optional f64 device scalar; every step adds its mean weighted per-pixel loss (+ the l2 term) to it ON THE
        DEVICE (`mp train` reads it once per epoch, pipeline.TrainPipeline). warmup=False: capture only (a re-capture after
        a learning-rate change -- the rate is a kernel argument of the captured launches); `replay.warmup_ran` tells.
        """
        if self._grad_hook is not None:
            raise NotImplementedError("graphed train step is single-GPU (gradient all-reduce is eager)")
        for t in (x, y):
            if not torch.is_tensor(t) or t.device.type != "cuda":
                raise ValueError("graphed train step needs device tensors")
        self._ensure_adam()
        if not warmup and (self._ws is None or (self.l2_reg and self._l2_ws is None)):
            #  a capture without a warm-up step is a RE-capture: the lazily created buffers must exist already, or
            # their allocations would land in the graph's private pool
            raise RuntimeError("make_graphed_train_step(warmup=False) needs a model that has already run a train step")
        step_dev = torch.tensor([self.iterations], dtype=torch.int64, device=self.device)
        k = self.optimizer_kwargs

        fused_tail = not self.l2_reg and os.environ.get("MPU_FUSED_ADAM") != "0"

        def body():
            if fused_tail:               # backward + optimizer in one call: the optimizer beside the weight gradients
                self.forward_backward(x, y, sample_weight, want_loss=False, adam=(0, step_dev))
                if loss_sum is not None:
                    loss_sum.add_(self.loss_mean().double())     # (the pass's own mean: no torch reduction inside the graph)
                return
            self.forward_backward(x, y, sample_weight, want_loss=False)
            self._add_l2(want_loss=loss_sum is not None)
            if loss_sum is not None:
                loss_sum.add_(self.loss_mean().double())
                if self.l2_reg:
                    loss_sum.add_(self.reg_loss.double())
            if os.environ.get("MPU_FUSED_ADAM") == "0":         # A/B: the two separate passes
                _lib.call("mpu_adam_step_device_counter", _lib.ptr(self.params), _lib.ptr(self.grads),
                          _lib.ptr(self._adam_m), _lib.ptr(self._adam_v), self.params.numel(), _lib.ptr(step_dev),
                          float(k["lr"]), float(k["beta_1"]), float(k["beta_2"]), float(k["epsilon"]), _lib.stream_ptr())
                self._repack()
                return
            _lib.call("mpu_unet_adam_pack", self._h, _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self._adam_m),
                      _lib.ptr(self._adam_v), 0, _lib.ptr(step_dev), float(k["lr"]), float(k["beta_1"]),
                      float(k["beta_2"]), float(k["epsilon"]), _lib.ptr(self.packed), _lib.stream_ptr())
            self._infer_dirty = True

        if warmup:
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):               # warm-up outside capture (lazy inits, allocations): a REAL step
                body()
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            body()
        if warmup:
            self.iterations += 1                        # the warm-up step

        def replay():
            graph.replay()
            self.iterations += 1
            self._infer_dirty = True
        replay.graph = graph
        replay.warmup_ran = bool(warmup)
        # Everything the captured launches address must outlive the graph. The Adam step counter is created HERE: without this
        # reference it was freed when this function returned, the caching allocator handed its block to the next small tensor
        # on the stream, and the replayed Adam kernels then read that tensor's bytes as the step count -- a wrong bias correction
        # (up to NaN) from the first allocation after the capture on (found by tests/test_gpu_pipeline.py, round 5).
        # (The model's own buffers too: a later, larger eager batch REPLACES self._ws -- the graph keeps running on the one it
        # was captured with, which must therefore stay allocated.)
        replay.keep_alive = (step_dev, x, y, sample_weight, loss_sum, self._ws, self._l2_ws, getattr(self, "reg_loss", None),
                             self._adam_m, self._adam_v, self.params, self.grads, self.packed, self.bn_state)
        return replay

    def train_step(self, x, y, sample_weight=None, want_loss=True):
        """One Model.fit inner step. Returns the per-pixel loss [B,H*W] (device) or None."""
        hook = self._grad_hook
        if hook is None and not self.l2_reg:     # single GPU, no l2 term: backward + optimizer as one library call
            self._ensure_adam()
            self.iterations += 1
            _, loss = self.forward_backward(x, y, sample_weight, want_loss, adam=(self.iterations, None))
            return loss
        events = getattr(hook, "ready_events", None)
        _, loss = self.forward_backward(x, y, sample_weight, want_loss, ready_events=events)
        if hook is not None:
            hook(self.grads)                     # data-parallel: SUM of replica gradients
        self._add_l2(want_loss)                  # once, after the replica sum (Keras scales it 1/replicas per replica)
        self.apply_gradients()
        return loss

    def train_on_batch(self, x, y, sample_weight=None):
        """Scalar Keras reports for the step: mean of the weighted per-pixel loss (+ the l2 term when l2_reg is set)."""
        loss = float(self.train_step(x, y, sample_weight).mean().item())
        return loss + float(self.reg_loss.item()) if self.l2_reg else loss

    def fit(self, data, steps_per_epoch, epochs=1, callbacks=None, initial_epoch=0, verbose=0, **kwargs):
        """Minimal Model.fit over an iterator of (x, y, w) batches (trainer.py:246-257)."""
        it = iter(data)
        history = []
        for ep in range(initial_epoch, epochs):
            tot = 0.0
            for _ in range(steps_per_epoch):
                x, y, w = next(it)
                tot += self.train_on_batch(x, y, w)
            history.append(tot / steps_per_epoch)
            if verbose:
                self.logger("epoch %d: loss %.5f" % (ep + 1, history[-1]))
            if self.stop_training:
                break
        return history

    def log(self):
        self.logger("UNet Model Summary\n------------------")
        self.logger("Image rows:        %i" % self.img_shape[0])
        self.logger("Image cols:        %i" % self.img_shape[1])
        self.logger("Image channels:    %i" % self.img_shape[2])
        self.logger("N classes:         %i" % self.n_classes)
        self.logger("CF factor:         %.3f" % self.cf ** 2)
        self.logger("Depth:             %i" % self.depth)
        self.logger("l2 reg:            %s" % self.l2_reg)
        self.logger("Padding:           %s" % self.padding)
        self.logger("Conv activation:   %s" % self.activation)
        self.logger("Out activation:    %s" % self.out_activation)
        self.logger("Receptive field:   %s" % self.receptive_field)
        self.logger("N params:          %i" % self.count_params())
        self.logger("Compute dtype:     %s (MI355X / gfx950 HIP kernels)" % str(self.dtype))
        self.logger("Crop:              None")
