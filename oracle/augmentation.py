"""
ORACLE (test infrastructure only) - NumPy restatement of the reference's on-the-fly augmentation
(SURVEY.md 8f row N1): Elastic2D = elastic_transform_2d (mpunet/augmentation/elastic_deformation.py:6-69)
applied per batch element with probability apply_prob, alpha/sigma drawn uniformly from their ranges and the
sample weight of augmented elements replaced by 0.33 (mpunet/augmentation/augmenters.py:13-107).
Pinned by tests/golden/elastic_golden.npz (outputs of the reference's own function, oracle/gen_golden_elastic.py).
"""
import numpy as np

from .geometry import rgi_linear, rgi_nearest


def gaussian_kernel1d(sigma, truncate=4.0):
    """scipy.ndimage._filters._gaussian_kernel1d(order=0): radius int(truncate*sigma + .5), normalised."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (float(sigma) * float(sigma)) * x ** 2)
    return phi / phi.sum(), radius


def correlate1d_symmetric_zero_pad(a, w, radius, axis):
    """
    scipy.ndimage.correlate1d (ni_filters.c NI_Correlate1D, symmetric branch) with mode='constant', cval=0:
    out[l] = in[l]*w[0] + sum_{j=-radius..-1} (in[l+j] + in[l-j]) * w[j]   (w centred), double precision.
    """
    a = np.moveaxis(np.asarray(a, np.float64), axis, -1)
    n = a.shape[-1]
    pad = np.zeros(a.shape[:-1] + (n + 2 * radius,), np.float64)
    pad[..., radius:radius + n] = a
    out = pad[..., radius:radius + n] * w[radius]
    for j in range(-radius, 0):
        out = out + (pad[..., radius + j:radius + j + n] + pad[..., radius - j:radius - j + n]) * w[radius + j]
    return np.moveaxis(out, -1, axis)


def gaussian_filter_zero_pad(a, sigma):
    """gaussian_filter(a, sigma, mode='constant', cval=0.) for a 2-D double array: axis 0 then axis 1."""
    w, r = gaussian_kernel1d(sigma)
    out = np.asarray(a, np.float64)
    for axis in range(out.ndim):
        out = correlate1d_symmetric_zero_pad(out, w, r, axis)
    return out


def elastic_transform_2d(image, labels, alpha, sigma, bg_val=0.0, noise=None):
    """
    elastic_deformation.py:6-69. image [H,W(,C)] f32, labels [H,W] or None. `noise` = the two uniform [0,1)
    fields the reference draws with np.random.rand (default: drawn here in the same order).
    """
    if image.ndim == 2:
        image = image[..., None]
    shape = image.shape[:2]
    C = image.shape[-1]
    bg = bg_val if isinstance(bg_val, (list, tuple, np.ndarray)) else [bg_val] * C
    coords = (np.arange(shape[0]), np.arange(shape[1]))
    if noise is None:
        noise = (np.random.rand(*shape), np.random.rand(*shape))
    dx = gaussian_filter_zero_pad(noise[0] * 2 - 1, sigma) * alpha
    dy = gaussian_filter_zero_pad(noise[1] * 2 - 1, sigma) * alpha
    x, y = np.mgrid[0:shape[0], 0:shape[1]]
    xi = np.stack([np.reshape(x + dx, -1), np.reshape(y + dy, -1)])
    out = np.empty(image.shape, image.dtype)
    for c in range(C):
        out[..., c] = rgi_linear(image[..., c], coords, xi, bg[c]).reshape(shape)
    lab = None
    if labels is not None:
        lab = rgi_nearest(labels, coords, xi, 0).reshape(shape).astype(labels.dtype)
    return out, lab
