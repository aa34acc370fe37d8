"""
mpunet.models.FusionModel on MI355X (mpunet/models/fusion_model.py:14-75):
softmax_k(sum_v W[v,k] x[n,v,k] + b[0,k]); W (V,K) init 1.0, b (1,K) init 0.0.
predict() accepts the reference's explicit [N,V,K] layout; the predict pipeline
uses the fused map+fuse kernel instead (multiplanarunet_amd.interpolation.map_and_fuse).
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
