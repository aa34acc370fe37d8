"""
On-the-fly augmentation of training batches on the GPU:
mpunet.augmentation.augmenters.Elastic2D (augmenters.py:13-131) over mpu_elastic_transform_2d.
Per batch element, with probability apply_prob: alpha and sigma are drawn uniformly from their ranges, two
uniform noise fields are drawn on the device, blurred (zero-padded Gaussian) and scaled into a displacement
field; the slice is re-sampled bilinearly (labels: nearest) and its sample weight replaced by aug_weight.
"""
import numpy as np
import torch

from . import _lib


def gaussian_kernel1d(sigma, truncate=4.0):
    """scipy.ndimage's order-0 kernel: radius int(truncate*sigma + .5), exp(-x^2 / 2 sigma^2) normalised (f64)."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (float(sigma) * float(sigma)) * x ** 2)
    return phi / phi.sum(), radius


def elastic_transform_2d(image, labels, alpha, sigma, bg_val=0.0, noise=None, generator=None):
    """
    elastic_deformation.py:6-69 on device tensors: image [H,W(,C)] f32, labels [H,W] u8 or None ->
    (image', labels'). noise: optional [2,H,W] f64 uniform fields (default: torch.rand on the device).
    """
    squeeze = image.ndim == 2
    img = (image[..., None] if squeeze else image).contiguous().float()
    H, W, C = (int(v) for v in img.shape)
    dev = img.device
    bg = bg_val if isinstance(bg_val, (list, tuple, np.ndarray)) else [bg_val] * C
    bgt = torch.tensor(np.asarray(bg, np.float64).astype(np.float32), device=dev)
    if noise is None:
        noise = torch.rand((2, H, W), dtype=torch.float64, device=dev, generator=generator)
    noise = torch.as_tensor(noise).to(device=dev, dtype=torch.float64).contiguous()
    w, radius = gaussian_kernel1d(sigma)
    wt = torch.tensor(w, dtype=torch.float64, device=dev)
    ws = torch.empty(int(_lib.load().mpu_elastic_workspace_doubles(H, W)), dtype=torch.float64, device=dev)
    out = torch.empty_like(img)
    lab = lab_out = None
    if labels is not None:
        lab = labels.to(device=dev, dtype=torch.uint8).contiguous()
        lab_out = torch.empty_like(lab)
    _lib.call("mpu_elastic_transform_2d", _lib.ptr(img), _lib.ptr(lab), H, W, C, _lib.ptr(noise), _lib.ptr(wt), radius,
              float(alpha), _lib.ptr(bgt), _lib.ptr(ws), _lib.ptr(out), _lib.ptr(lab_out), _lib.stream_ptr())
    return (out[..., 0] if squeeze else out), lab_out


class Elastic2D:
    """augmenters.py:13-131 (Elastic / Elastic2D). alpha, sigma: numbers or [lo, hi] ranges."""
    __name__ = "Elastic2D"

    def __init__(self, alpha, sigma, apply_prob, aug_weight=0.33, seed=None):
        for name, v in (("alpha", alpha), ("sigma", sigma)):
            if isinstance(v, (list, tuple)):
                if len(v) != 2:
                    raise ValueError("Invalid list of %ss specified '%s'. Should be 2 numbers." % (name, v))
                if v[1] <= v[0]:
                    raise ValueError("%s upper is smaller than %s lower (%s)" % (name, name, v))
        if apply_prob > 1 or apply_prob < 0:
            raise ValueError("Apply probability is invalid with value %3.f" % apply_prob)
        self._alpha, self._sigma, self.apply_prob, self.weight = alpha, sigma, apply_prob, aug_weight
        self.rng = np.random.RandomState(seed)
        self._gen = None
        self._seed = seed

    def _draw(self, v):
        return self.rng.uniform(v[0], v[1], 1)[0] if isinstance(v, (list, tuple)) else v

    @property
    def alpha(self):
        return self._draw(self._alpha)

    @property
    def sigma(self):
        return self._draw(self._sigma)

    def __call__(self, batch_x, batch_y, bg_values, batch_w=None):
        """batch_x [B,H,W,C] f32, batch_y [B,H,W] u8 (device); bg_values: per element list of C values."""
        if self._gen is None and self._seed is not None:
            self._gen = torch.Generator(device=batch_x.device)
            self._gen.manual_seed(int(self._seed))
        mask = self.rng.rand(len(batch_x)) <= self.apply_prob
        for i, aug in enumerate(mask):
            if not aug:
                continue
            x, y = elastic_transform_2d(batch_x[i], batch_y[i], self.alpha, self.sigma, bg_values[i], generator=self._gen)
            batch_x[i] = x
            batch_y[i] = y
            if batch_w is not None:
                batch_w[i] = self.weight
        return (batch_x, batch_y, batch_w) if batch_w is not None else (batch_x, batch_y)

    def __str__(self):
        return "%s(alpha=%s, sigma=%s, apply_prob=%.3f)" % (self.__name__, self._alpha, self._sigma, self.apply_prob)


def build_augmenters(spec, seed=None):
    """fit.augmenters of train_hparams.yaml: [{cls_name: "Elastic2D", kwargs: {...}}, ...]."""
    out = []
    for k, a in enumerate(spec or []):
        if a.get("cls_name") != "Elastic2D":
            raise NotImplementedError("augmenter %s is outside the 2-D path (available: Elastic2D)" % a.get("cls_name"))
        out.append(Elastic2D(seed=None if seed is None else seed + k, **a.get("kwargs", {})))
    return out
