"""
mpunet.models.FusionModel on MI355X (mpunet/models/fusion_model.py:14-75):
softmax_k(sum_v W[v,k] x[n,v,k] + b[0,k]); W (V,K) init 1.0, b (1,K) init 0.0.
predict() accepts the reference's explicit [N,V,K] layout; the predict pipeline
uses the fused map+fuse kernel instead (multiplanarunet_amd.interpolation.map_and_fuse).
Training (mpunet/bin/train_fusion.py:327-362): compile() + fit()/train_on_batch() run the per-point
generalized Dice loss, its gradient and Keras Adam(1e-3) in mpu_fusion_train_step (csrc/fusion_train.hip).
Data parallel (one process per GPU under torch.distributed): every rank holds the points of ITS
images; a fit() step is the rank's share of the batch -> gradient / loss SUMS (mpu_fusion_grad_sums) -> one SUM all-reduce
of V*K + K + 2 doubles -> mpu_fusion_apply_sums on every rank (identical weights everywhere, no broadcast needed); the
validation Dice comes from all-reduced per-class counts.
"""
import numpy as np
import torch

from . import _lib


class _FusionLayerShim:
    def __init__(self, model):
        self._m = model
        self.name = "fusion_layer"

    def get_weights(self):
        return [self._m.W.cpu().numpy(), self._m.b.cpu().numpy()]

    def set_weights(self, weights):
        self._m.set_weights(weights)


class FusionModel:
    def __init__(self, n_inputs, n_classes, weight="Simple", logger=None, verbose=True, device="cuda"):
        self.n_inputs = n_inputs
        self.n_classes = n_classes
        self.logger = logger or (lambda *a, **k: print(*a))
        self.device = torch.device(device)
        self.W = torch.ones((n_inputs, n_classes), dtype=torch.float32, device=self.device)
        self.b = torch.zeros((1, n_classes), dtype=torch.float32, device=self.device)
        self.layers = [_FusionLayerShim(self)]
        self.weight = weight                       # GDL class-weight type: Simple | Square | Uniform (identical per point)
        if str(weight).lower() not in ("simple", "square", "uniform"):
            raise ValueError('The variable type_weight "%s" is not defined.' % weight)
        self.optimizer_kwargs = None
        self.iterations = 0
        self.stop_training = False
        if verbose:
            self._log()

    def _log(self):
        self.logger("Input:      (None, %d, %d)" % (self.n_inputs, self.n_classes))
        self.logger("Output:     (None, %d)" % self.n_classes)
        self.logger("N weights:  %s" % self.count_params())

    def count_params(self):
        return self.n_inputs * self.n_classes + self.n_classes

    def get_weights(self):
        return [self.W.cpu().numpy(), self.b.cpu().numpy()]

    def set_weights(self, weights):
        W, b = weights
        W = np.asarray(W, np.float32)
        b = np.asarray(b, np.float32).reshape(1, -1)
        if W.shape != (self.n_inputs, self.n_classes) or b.shape != (1, self.n_classes):
            raise ValueError("FusionModel weights must be W (%d,%d), b (1,%d)" %
                             (self.n_inputs, self.n_classes, self.n_classes))
        self.W = torch.tensor(W, device=self.device)
        self.b = torch.tensor(b, device=self.device)

    def save_weights(self, path):
        with open(path, "wb") as f:
            np.savez(f, W=self.W.cpu().numpy(), b=self.b.cpu().numpy())

    def load_weights(self, path, by_name=False):
        with np.load(path) as z:
            self.set_weights([z["W"], z["b"]])

    # ---- training ---------------------------------------------------------------------------------
    def compile(self, optimizer="Adam", loss=None, metrics=None, optimizer_kwargs=None, **kwargs):
        """train_fusion.py:343-345: Adam(lr=1e-3) (Keras defaults b1 .9, b2 .999, eps 1e-7), loss = the model's GDL."""
        if optimizer not in ("Adam", "adam"):
            raise NotImplementedError("FusionModel training supports the reference's optimizer (Adam) only")
        kw = dict(lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7)
        kw.update(optimizer_kwargs or {})
        if "learning_rate" in kw:
            kw["lr"] = kw.pop("learning_rate")
        self.optimizer_kwargs = kw
        n = self.count_params()
        self._adam_m = torch.zeros(n, dtype=torch.float32, device=self.device)
        self._adam_v = torch.zeros(n, dtype=torch.float32, device=self.device)
        self._ws = torch.empty(int(_lib.load().mpu_fusion_train_workspace_floats(self.n_inputs, self.n_classes)),
                               dtype=torch.float32, device=self.device)
        self.iterations = 0
        return self

    def _check_xy(self, x, y):
        xd = torch.as_tensor(x).to(device=self.device, dtype=torch.float32).contiguous()
        if xd.ndim != 3 or xd.shape[1] != self.n_inputs or xd.shape[2] != self.n_classes:
            raise ValueError("expected input [N,%d,%d]" % (self.n_inputs, self.n_classes))
        yd = torch.as_tensor(y).to(device=self.device).reshape(-1).to(torch.uint8).contiguous()
        if yd.shape[0] != xd.shape[0]:
            raise ValueError("x and y must have the same number of points")
        return xd, yd

    def _step(self, xd, yd, apply, want_grads=False):
        if self.optimizer_kwargs is None:
            self.compile()
        k = self.optimizer_kwargs
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        grads = torch.empty(self.count_params(), dtype=torch.float32, device=self.device) if want_grads else None
        t = 0
        if apply:
            self.iterations += 1
            t = self.iterations
        _lib.call("mpu_fusion_train_step", _lib.ptr(xd), _lib.ptr(yd), xd.shape[0], self.n_inputs, self.n_classes,
                  _lib.ptr(self.W), _lib.ptr(self.b), _lib.ptr(self._adam_m), _lib.ptr(self._adam_v), t,
                  float(k["lr"]), float(k["beta_1"]), float(k["beta_2"]), float(k["epsilon"]), _lib.ptr(self._ws),
                  _lib.ptr(grads), _lib.ptr(loss), _lib.stream_ptr())
        return loss, grads

    # ---- data-parallel halves of a step ------------------------------------------------------------
    @staticmethod
    def _world():
        import torch.distributed as dist
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def _local_sums(self, xd, yd):
        """This rank's share of a batch -> f64 [V*K + K + 2]: gradient sums, loss sum, point count (device)."""
        sums = torch.empty(self.count_params() + 2, dtype=torch.float64, device=self.device)
        n = int(xd.shape[0])
        _lib.call("mpu_fusion_grad_sums", _lib.ptr(xd) if n else None, _lib.ptr(yd) if n else None, n, self.n_inputs,
                  self.n_classes, _lib.ptr(self.W), _lib.ptr(self.b), _lib.ptr(self._ws), _lib.ptr(sums), _lib.stream_ptr())
        return sums

    def _apply_sums(self, sums, apply=True, want_grads=False):
        k = self.optimizer_kwargs
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        grads = torch.empty(self.count_params(), dtype=torch.float32, device=self.device) if want_grads else None
        t = 0
        if apply:
            self.iterations += 1
            t = self.iterations
        _lib.call("mpu_fusion_apply_sums", _lib.ptr(sums), self.n_inputs, self.n_classes, _lib.ptr(self.W), _lib.ptr(self.b),
                  _lib.ptr(self._adam_m), _lib.ptr(self._adam_v), t, float(k["lr"]), float(k["beta_1"]), float(k["beta_2"]),
                  float(k["epsilon"]), _lib.ptr(grads), _lib.ptr(loss), _lib.stream_ptr())
        return loss, grads

    def _step_dp(self, xd, yd, apply=True, want_grads=False):
        """One step over the union of all ranks' shares: (loss of the whole batch, its point count, gradients)."""
        import torch.distributed as dist
        if self.optimizer_kwargs is None:
            self.compile()
        sums = self._local_sums(xd, yd)
        if self._world() > 1:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        loss, grads = self._apply_sums(sums, apply, want_grads)
        return loss, sums[-1], grads

    def _val_dice_dp(self, xv, yv):
        """Mean foreground Dice of the argmax over ALL ranks' validation points, from SUM-all-reduced per-class counts
        (true positives, targets, selections): dice_all's formula (smooth 1, NaN where a class is in neither)."""
        import torch.distributed as dist
        K = self.n_classes
        cnt = torch.zeros(3 * K, dtype=torch.int64, device=self.device)
        if xv.shape[0]:
            pred = self.predict(xv).argmax(-1)
            yl = yv.long()
            cnt[:K] = torch.bincount(yl[yl == pred], minlength=K)[:K]
            cnt[K:2 * K] = torch.bincount(yl, minlength=K)[:K]
            cnt[2 * K:] = torch.bincount(pred, minlength=K)[:K]
        if self._world() > 1:
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        c = cnt.cpu().numpy().astype(np.float64)
        classes = np.arange(1, max(2, K))
        d = np.full(classes.shape, np.nan, dtype=np.float32)
        for i, k in enumerate(classes):
            tp, na, nb = (c[k], c[K + k], c[2 * K + k]) if k < K else (0.0, 0.0, 0.0)
            if na or nb:
                d[i] = (1.0 + 2.0 * tp) / (1.0 + na + nb)
        return d

    def loss_and_gradients(self, x, y):
        """Batch loss and d loss / d (W, b) without updating (parity tests)."""
        xd, yd = self._check_xy(x, y)
        loss, g = self._step(xd, yd, apply=False, want_grads=True)
        V, K = self.n_inputs, self.n_classes
        return float(loss.item()), g[:V * K].reshape(V, K).cpu().numpy(), g[V * K:].reshape(1, K).cpu().numpy()

    def train_on_batch(self, x, y):
        xd, yd = self._check_xy(x, y)
        loss, _ = self._step(xd, yd, apply=True)
        return float(loss.item())

    def fit(self, x, y, batch_size=2 ** 17, epochs=1, shuffle=True, validation_data=None, early_stopping=None,
            verbose=0, seed=None, callbacks=None, data_parallel=None, **kwargs):
        """
        Keras Model.fit on an in-memory point set (train_fusion.py:205-216): per epoch a fresh shuffle, batches of
        batch_size points (the last one may be short), epoch loss = point-weighted mean of the batch losses.
        validation_data=(X_val, y_val): logs val_dice = mean foreground Dice of the argmax (ValDiceScores,
        callbacks/validation.py:308-354); early_stopping=n: stop after n epochs without a val_dice improvement
        (EarlyStopping(monitor='val_dice', mode='max', min_delta=0)). Returns {"loss": [...], "val_dice": [...]}.
        data_parallel (default: torch.distributed has > 1 rank): x, y are THIS rank's points (possibly none); a step takes
        ceil(batch_size / world) of them on every rank (ranks that have run out contribute nothing), the gradient sums are
        all-reduced, every rank applies the same update and sees the same loss / val_dice / early-stopping decision.
        """
        from .interpolation import dice_all
        xd, yd = self._check_xy(x, y)
        N = xd.shape[0]
        world = self._world()
        dp = (world > 1) if data_parallel is None else bool(data_parallel)
        b_loc = -(-int(batch_size) // world) if dp else int(batch_size)
        steps, N_all = -(-N // b_loc), N
        if dp and world > 1:
            import torch.distributed as dist
            t = torch.tensor([N, steps], dtype=torch.int64, device=self.device)
            tmax = t.clone()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            N_all, steps = int(t[0].item()), int(tmax[1].item())
        if N_all == 0:
            raise ValueError("fit() needs at least one point")
        gen = torch.Generator(device="cpu")
        gen.manual_seed(int(seed) if seed is not None else int(np.random.randint(0, 2 ** 31 - 1)))
        if validation_data is not None:
            xv, yv = self._check_xy(*validation_data)
        hist = {"loss": [], "val_dice": []}
        best, wait = -np.inf, 0
        self.stop_training = False
        for ep in range(epochs):
            if shuffle:
                perm = torch.randperm(N, generator=gen).to(self.device)
                xe, ye = xd[perm], yd[perm]
            else:
                xe, ye = xd, yd
            tot = torch.zeros(1, dtype=torch.float64, device=self.device)
            for j in range(steps):
                s, e = min(N, j * b_loc), min(N, (j + 1) * b_loc)
                if dp:
                    loss, n_step, _ = self._step_dp(xe[s:e], ye[s:e])
                    tot += loss.double() * n_step
                else:
                    loss, _ = self._step(xe[s:e], ye[s:e], apply=True)
                    tot += loss.double() * (e - s)
            hist["loss"].append(float(tot.item()) / N_all)
            msg = "Epoch %d/%d - loss: %.6f" % (ep + 1, epochs, hist["loss"][-1])
            if validation_data is not None:
                if dp:
                    d = self._val_dice_dp(xv, yv)
                else:
                    d = dice_all(yv, self.predict(xv).argmax(-1), n_classes=self.n_classes, ignore_zero=True)
                vd = float(np.nanmean(d))
                hist["val_dice"].append(vd)
                msg += " - val_dice: %.4f (per class %s)" % (vd, np.round(d, 4))
                if early_stopping is not None:
                    if vd > best:
                        best, wait = vd, 0
                    else:
                        wait += 1
                        if wait >= early_stopping:
                            self.stop_training = True
            if verbose:
                self.logger(msg)
            if self.stop_training:
                break
        return hist

    def predict(self, x, batch_size=10 ** 4, verbose=0):
        """x [N,V,K] (numpy or device tensor) -> probabilities [N,K]."""
        numpy_in = not torch.is_tensor(x)
        xd = torch.as_tensor(x).to(device=self.device, dtype=torch.float32).contiguous()
        if xd.ndim != 3 or xd.shape[1] != self.n_inputs or xd.shape[2] != self.n_classes:
            raise ValueError("expected input [N,%d,%d]" % (self.n_inputs, self.n_classes))
        N = xd.shape[0]
        probs = torch.empty((N, self.n_classes), dtype=torch.float32, device=self.device)
        _lib.call("mpu_fusion_forward", _lib.ptr(xd), N, self.n_inputs, self.n_classes,
                  _lib.ptr(self.W), _lib.ptr(self.b), _lib.ptr(probs), None, _lib.stream_ptr())
        return probs.cpu().numpy() if numpy_in else probs
