"""
Host side of the predict-time geometry path: the few dozen fp64 numbers that
define a view (plane basis, axes, offsets) are computed here with NumPy exactly
as the reference computes them; every per-sample / per-voxel operation runs in
the HIP kernels of csrc/geometry.hip through the C ABI.

Mirrors (same names, argument meaning, return layout):
  sample_plane_at ............ mpunet/interpolation/sample_grid.py:192-244
  get_voxel_axes_real_space .. mpunet/interpolation/sample_grid.py:63-98
  get_view_from / sample_at .. mpunet/sequences/isotrophic_live_view_sequence_2d.py:29-117
  get_voxel_grid_real_space .. mpunet/interpolation/sample_grid.py:101-130
  map_real_space_pred ........ mpunet/utils/fusion/fuse_and_predict.py:92-137
  predict_volume ............. mpunet/utils/fusion/fuse_and_predict.py:81-89
  merge_multi_view_preds ..... mpunet/bin/predict.py:349-366

dtype note: the reference relies on NumPy promotion rules; the goldens were
produced under NumPy >= 2 (NEP 50). All dtypes are spelled out explicitly here
so the result does not depend on the NumPy version.
"""
import ctypes as C
import numpy as np
import torch

from . import _lib


# --------------------------------------------------------------------------- #
# view basis (host, fp64) -- sample_grid.py:194-224, linalg.py:33-51
# --------------------------------------------------------------------------- #
def _rotation_matrix(axis64, angle_deg):
    theta = np.deg2rad(np.float64(angle_deg))
    axis64 = np.asarray(axis64, np.float64)
    a = np.cos(theta / 2.0)
    b, c, d = -axis64 * np.sin(theta / 2.0)
    aa, bb, cc, dd = a * a, b * b, c * c, d * d
    bc, ad, ac, ab, bd, cd = b * c, a * d, a * c, a * b, b * d, c * d
    return np.array([[aa + bb - cc - dd, 2 * (bc + ad), 2 * (bd - ac)],
                     [2 * (bc - ad), aa + cc - bb - dd, 2 * (cd + ab)],
                     [2 * (bd + ac), 2 * (cd - ab), aa + dd - bb - cc]], np.float64)


def plane_basis(norm_vector, noise=None):
    """[u v n_hat] (columns) of the sampling plane with normal `norm_vector`."""
    n = np.array(norm_vector, dtype=np.float32)
    n = (n / np.linalg.norm(n).astype(np.float32)).astype(np.float32)
    if noise is not None:
        n = (n.astype(np.float64) + np.asarray(noise, np.float64)).astype(np.float32)
    n = (n / np.linalg.norm(n).astype(np.float32)).astype(np.float32)
    if np.all(n[:2] < 0.2):
        n[:2] = np.abs(n[:2])
    if np.all(np.isclose(n[:2], 0)):
        u = np.array([1.0, 0.0, 0.0])
        v = np.array([0.0, 1.0, 0.0])
    else:
        vs = n.copy()
        vs[2] = vs[2] + np.float32(1)
        vs = (vs / np.linalg.norm(vs).astype(np.float32)).astype(np.float32)
        ax = np.cross(n, vs).astype(np.float32)
        ax = (ax / np.linalg.norm(ax).astype(np.float32)).astype(np.float32)
        u = _rotation_matrix(ax, -90).dot(n.astype(np.float64))
        v = np.cross(n.astype(np.float64), u)
    return np.column_stack((u, v, n.astype(np.float64))).astype(np.float64)


class ViewGeometry:
    """Everything `get_view_from(image, view, 'same+<extra>')` derives from the view alone."""

    def __init__(self, view, sample_dim, real_space_span, n_planes="same+20", noise=None):
        self.view = np.asarray(view, np.float64)
        self.dim = int(sample_dim)
        self.span = real_space_span
        self.basis = plane_basis(view, noise)
        self.inv_basis = np.linalg.inv(self.basis)
        hd = real_space_span // 2                                  # floor, sample_grid.py:227
        self.real_axis = np.linspace(-hd, hd, self.dim)            # returned `g`
        self.g_start = float(-hd)                                  # np.mgrid axis = i*step+start
        self.g_step = float((hd - (-hd)) / float(self.dim - 1))
        sample_res = real_space_span / (self.dim - 1)
        extra = 0
        if n_planes == "same":
            P = self.dim
        elif isinstance(n_planes, str) and n_planes[:5] == "same+":
            extra = int(n_planes.split("+")[-1])
            P = self.dim + extra
        else:
            raise NotImplementedError("n_planes=%r (only 'same' / 'same+N')" % (n_planes,))
        bounds = (real_space_span + (extra * sample_res)) / 2
        self.offsets = np.linspace(-bounds, bounds, P)
        self.n_planes = P
        self._dev_axes = {}

    def device_axes(self, device):
        """(real_axis, offsets) as device f64 tensors: ONE upload per view and device, shared by the sampler and the
        back-mapping (small synchronous host-to-device copies are what the geometry kernels would otherwise wait on)."""
        key = str(device)
        hit = self._dev_axes.get(key)
        if hit is None or hit[0] is not self.offsets or hit[1] is not self.real_axis:      # (re)assigned arrays: upload again
            both = torch.tensor(np.concatenate([self.real_axis, self.offsets]), device=device)
            hit = (self.offsets, self.real_axis, both[:len(self.real_axis)], both[len(self.real_axis):])
            self._dev_axes[key] = hit
        return hit[2], hit[3]

    def struct(self, rot_mat, axes=None):
        g = _lib.ViewGeom()
        if axes is not None:
            for k in range(3):
                g.vol_axis[k] = _lib.make_axis(axes[k])
        g.basis[:] = self.basis.ravel().tolist()
        rot = np.eye(3) if rot_mat is None else rot_mat
        g.rot[:] = np.asarray(rot, np.float64).ravel().tolist()
        g.has_rot = 0 if rot_mat is None else 1
        g.dim, g.n_planes = self.dim, self.n_planes
        g.g_start, g.g_step = self.g_start, self.g_step
        return g


# --------------------------------------------------------------------------- #
# Volume -- the ImagePair duck type the hot path consumes
# --------------------------------------------------------------------------- #
class Volume:
    """
    image f32 [X,Y,Z,C], labels u8 [X,Y,Z] or None, affine 4x4. `bg_value` is a
    number per channel (the reference's '1pct' percentile is host prep, outside
    the path); `scaler` is None or (center[C], scale[C]) of a fitted sklearn
    scaler (image_pair.py:323-341, preprocessing/scaling.py:47-89).
    """

    def __init__(self, image, labels=None, affine=None, bg_value=0.0, scaler=None,
                 bg_class=0, identifier="volume", device="cuda"):
        image = torch.as_tensor(image)
        if image.ndim != 4:
            raise ValueError("Input img of dim %i must be dim 4." % image.ndim)
        self.image = image.to(device=device, dtype=torch.float32).contiguous()
        self.labels = None if labels is None else \
            torch.as_tensor(labels).to(device=device, dtype=torch.uint8).contiguous()
        self.affine = np.eye(4) if affine is None else np.asarray(affine, np.float64)
        C_ = self.image.shape[-1]
        if not isinstance(bg_value, (list, tuple, np.ndarray)):
            bg_value = [bg_value] * C_
        if len(bg_value) != C_:
            raise ValueError("'bg_value' should be a list of length 'n_channels'. "
                             "Got {} for n_channels={}".format(bg_value, C_))
        self.bg_value = [float(b) for b in bg_value]
        self.bg_class = int(bg_class)
        self.identifier = identifier
        self.device = self.image.device
        self._bg = torch.tensor(self.bg_value, dtype=torch.float32, device=self.device)
        self._center = self._scale = None
        self.scaler = scaler
        if scaler is not None:
            c, s = scaler
            self._center = torch.tensor(np.asarray(c, np.float64), device=self.device)
            self._scale = torch.tensor(np.asarray(s, np.float64), device=self.device)
        # voxel axes centred in real space + rot_mat (sample_grid.py:63-98)
        basis = self.affine[:3, :3]
        pixdims = np.linalg.norm(basis, axis=0)
        transform = np.diag(pixdims)
        if np.any(np.sign(np.diagonal(transform)) == -1):      # view_interpolator.py:113-114
            raise AssertionError("negative axis")
        self.rot_mat = transform.dot(np.linalg.inv(basis)) \
            if np.any(~np.isclose(transform, basis)) else None
        axes = []
        for n, pd in zip(self.image.shape[:3], pixdims):
            a32 = np.arange(n, dtype=np.float32) - np.float32((n - 1) / 2)
            axes.append(a32.astype(np.float64) * np.float64(pd))
        self.axes = axes
        self._axes_dev = [torch.tensor(a, device=self.device) for a in axes]

    @property
    def shape(self):
        return np.array(self.image.shape)

    @property
    def n_channels(self):
        return int(self.image.shape[-1])

    @property
    def predict_mode(self):
        return self.labels is None

    @staticmethod
    def fit_robust_scaler(image_np):
        """sklearn RobustScaler per channel: center_=median, scale_=IQR (host prep)."""
        c, s = [], []
        for ch in range(image_np.shape[-1]):
            q = np.nanpercentile(image_np[..., ch].astype(np.float64).ravel(), (25.0, 50.0, 75.0))
            c.append(q[1])
            sc = q[2] - q[0]
            s.append(sc if sc != 0 else 1.0)
        return np.array(c), np.array(s)

    def voxel_grid(self):
        """get_voxel_grid_real_space as numbers: p = A(i,j,k) - mean = A(ijk - (dims-1)/2)."""
        g = _lib.VoxelGrid()
        A = self.affine[:3, :3]
        g.A[:] = A.ravel().tolist()
        X, Y, Z = (int(v) for v in self.image.shape[:3])
        g.center[:] = A.dot(np.array([(X - 1) / 2.0, (Y - 1) / 2.0, (Z - 1) / 2.0])).tolist()
        g.shape[:] = [X, Y, Z]
        return g


# --------------------------------------------------------------------------- #
# sampling
# --------------------------------------------------------------------------- #
def sample_view(volume, geom, want_labels=True, out=None):
    """Planes of one view, model order: X f32 [P,dim,dim,C], y u8 [P,dim,dim] or None."""
    P, d, Cn = geom.n_planes, geom.dim, volume.n_channels
    dev = volume.device
    X = out if out is not None else torch.empty((P, d, d, Cn), dtype=torch.float32, device=dev)
    y = None
    if want_labels and volume.labels is not None:
        y = torch.empty((P, d, d), dtype=torch.uint8, device=dev)
    offs = geom.device_axes(dev)[1] if hasattr(geom, "device_axes") else torch.tensor(geom.offsets, device=dev)
    shape = (C.c_int32 * 4)(*[int(v) for v in volume.image.shape])
    gs = geom.struct(volume.rot_mat, volume.axes)
    _lib.call("mpu_sample_view_planes", _lib.ptr(volume.image),
              _lib.ptr(volume.labels if y is not None else None), shape,
              _lib.ptr(volume._axes_dev[0]), _lib.ptr(volume._axes_dev[1]),
              _lib.ptr(volume._axes_dev[2]), C.byref(gs), _lib.ptr(offs),
              _lib.ptr(volume._bg), volume.bg_class, _lib.ptr(volume._center),
              _lib.ptr(volume._scale), _lib.ptr(X), _lib.ptr(y), _lib.stream_ptr())
    return X, y


class ViewSampler:
    """
    The predict-time half of IsotrophicLiveViewSequence2D: holds views, sample
    dim and real-space span and serves get_view_from().
    """

    def __init__(self, views, dim, real_space_span, n_classes=None, batch_size=8, **kwargs):
        self.views = np.asarray(views, np.float64)
        self.sample_dim = int(dim)
        self.real_space_span = real_space_span
        self.n_classes = n_classes
        self.batch_size = batch_size

    def geometry(self, view, n_planes="same+20"):
        return ViewGeometry(view, self.sample_dim, self.real_space_span, n_planes)

    def get_view_from(self, image, view, n_planes="same+20"):
        """
        Returns (Xs [dim,dim,P,C], ys [dim,dim,P] or None, (real_axis, real_axis,
        offsets), inv_basis) as the reference does; Xs/ys are permuted views of
        plane-major device tensors, so predict_volume's moveaxis is free.
        """
        geom = self.geometry(view, n_planes)
        X, y = sample_view(image, geom, want_labels=not image.predict_mode)
        Xs = X.permute(1, 2, 0, 3)
        ys = None if y is None else y.permute(1, 2, 0)
        return Xs, ys, (geom.real_axis, geom.real_axis, geom.offsets), geom.inv_basis


# --------------------------------------------------------------------------- #
# back-mapping + fusion
# --------------------------------------------------------------------------- #
class _ViewPredHolder:
    """Keeps the device buffers a mpu_view_pred points at alive."""

    def __init__(self, pred, grid, inv_basis, device, dev_axes=None):
        if pred.ndim != 4:
            raise ValueError("pred must be [P,dim,dim,K]")
        self.pred = pred.to(dtype=torch.float32).contiguous()
        if dev_axes is not None:            # ViewGeometry.device_axes: already resident
            self.g, self.offs = dev_axes
        else:
            self.g = torch.tensor(np.asarray(grid[0], np.float64), device=device)
            self.offs = torch.tensor(np.asarray(grid[2], np.float64), device=device)
        s = _lib.ViewPred()
        s.inv_basis[:] = np.asarray(inv_basis, np.float64).ravel().tolist()
        s.d_pred = self.pred.data_ptr()
        s.d_g = self.g.data_ptr()
        s.d_offsets = self.offs.data_ptr()
        s.dim = int(self.g.shape[0])
        s.n_planes = int(self.offs.shape[0])
        s.g_axis = _lib.make_axis(np.asarray(grid[0], np.float64))
        s.o_axis = _lib.make_axis(np.asarray(grid[2], np.float64))
        self.struct = s


def predict_volume(model, X, batch_size=8, axis=0):
    """fuse_and_predict.py:81-89: move `axis` first, model.predict, move back."""
    X = torch.movedim(X, axis, 0) if torch.is_tensor(X) else np.moveaxis(X, axis, 0)
    pred = model.predict(X, batch_size=batch_size, verbose=0)
    return torch.movedim(pred, 0, axis) if torch.is_tensor(pred) else np.moveaxis(pred, 0, axis)


def map_real_space_pred(pred, grid, inv_basis, volume, method="nearest"):
    """
    pred in the reference layout [dim,dim,P,K] (torch, device) on axes
    grid=(g,g,offsets) -> mapped f32 [X,Y,Z,K]; OOB voxels -> [1,0,..,0].
    `volume` stands in for voxel_grid_real_space (computed on the fly).
    """
    if method != "nearest":
        raise NotImplementedError("only method='nearest' is on the path (predict.py:329-331)")
    pm = pred.permute(2, 0, 1, 3)
    h = _ViewPredHolder(pm, grid, inv_basis, volume.device)
    K = int(pm.shape[-1])
    X, Y, Z = (int(v) for v in volume.image.shape[:3])
    mapped = torch.empty((X, Y, Z, K), dtype=torch.float32, device=volume.device)
    vg = volume.voxel_grid()
    _lib.call("mpu_map_view_nearest", C.byref(vg), C.byref(h.struct), K, _lib.ptr(mapped),
              _lib.stream_ptr())
    return mapped


def map_and_fuse(volume, view_preds, W=None, b=None, sum_fusion=False,
                 want_probs=True, want_labels=True):
    """
    Fused _multi_view_predict_on + merge_multi_view_preds. view_preds: list of
    (pred [P,dim,dim,K] device f32, grid=(g,g,offsets), inv_basis[, ViewGeometry.device_axes]). Returns
    (merged f32 [X,Y,Z,K] or None, merged_map u8 [X,Y,Z] or None).
    """
    dev = volume.device
    holders = [_ViewPredHolder(vp[0], vp[1], vp[2], dev, vp[3] if len(vp) > 3 else None) for vp in view_preds]
    V = len(holders)
    K = int(holders[0].pred.shape[-1])
    arr = (_lib.ViewPred * V)(*[h.struct for h in holders])
    X, Y, Z = (int(v) for v in volume.image.shape[:3])
    probs = torch.empty((X, Y, Z, K), dtype=torch.float32, device=dev) if want_probs else None
    labels = torch.empty((X, Y, Z), dtype=torch.uint8, device=dev) if want_labels else None
    Wd = bd = None
    if not sum_fusion:
        Wd = torch.as_tensor(W, dtype=torch.float32).to(dev).reshape(V, K).contiguous()
        bd = torch.as_tensor(b, dtype=torch.float32).to(dev).reshape(K).contiguous()
    vg = volume.voxel_grid()
    _lib.call("mpu_map_fuse_views", C.byref(vg), arr, V, K, _lib.ptr(Wd), _lib.ptr(bd),
              1 if sum_fusion else 0, _lib.ptr(probs), _lib.ptr(labels), _lib.stream_ptr())
    return probs, labels


def map_accumulate(volume, pred_chunk, grid, inv_basis, Wv, p_lo, p_hi, owns_oob, z):
    """Multi-GPU predict: z += Wv * nearest(x_v) for planes [p_lo,p_hi) (pred_chunk holds only those)."""
    dev = volume.device
    h = _ViewPredHolder(pred_chunk, grid, inv_basis, dev)
    K = int(pred_chunk.shape[-1])
    Wd = torch.as_tensor(Wv, dtype=torch.float32).to(dev).reshape(K).contiguous()
    vg = volume.voxel_grid()
    _lib.call("mpu_map_accumulate_view", C.byref(vg), C.byref(h.struct), K, _lib.ptr(Wd),
              int(p_lo), int(p_hi), 1 if owns_oob else 0, _lib.ptr(z), _lib.stream_ptr())
    return z


def fusion_finalize(z, b=None, sum_fusion=False, want_probs=True):
    K = int(z.shape[-1])
    n = z.numel() // K
    probs = torch.empty_like(z) if want_probs else None
    labels = torch.empty(z.shape[:-1], dtype=torch.uint8, device=z.device)
    bd = None if sum_fusion else torch.as_tensor(b, dtype=torch.float32).to(z.device).reshape(K).contiguous()
    _lib.call("mpu_fusion_finalize", _lib.ptr(z), n, K, _lib.ptr(bd), 1 if sum_fusion else 0,
              _lib.ptr(probs), _lib.ptr(labels), _lib.stream_ptr())
    return probs, labels


def pred_to_class(tensor, img_dims=3, threshold=0.5, has_batch_dim=False):
    """mpunet/utils/utils.py:311-328: integer maps pass through, single-channel scores are thresholded,
    multi-class scores -> argmax(-1) as uint8. Accepts device tensors (stays on the device) or ndarrays."""
    is_t = torch.is_tensor(tensor)
    is_int = (not tensor.dtype.is_floating_point and tensor.dtype != torch.bool) if is_t \
        else np.issubdtype(tensor.dtype, np.integer)
    if len(tensor.shape) == img_dims + int(has_batch_dim):
        return tensor if is_int else tensor >= threshold
    if tensor.shape[-1] == 1:
        if is_int:
            return tensor.squeeze() if is_t else np.squeeze(tensor)
        return tensor >= threshold
    return tensor.argmax(-1).to(torch.uint8) if is_t else tensor.argmax(-1).astype(np.uint8)


def dice(y_true, y_pred, smooth=1.0):
    """mpunet/evaluate/metrics.py:13-23 (binary sets)."""
    s1 = np.asarray(y_true).ravel().astype(bool)
    s2 = np.asarray(y_pred).ravel().astype(bool)
    return (smooth + 2 * np.logical_and(s1, s2).sum()) / (smooth + s1.sum() + s2.sum())


def dice_all(y_true, y_pred, smooth=1.0, n_classes=None, ignore_zero=True, skip_if_no_y=False):
    """mpunet/evaluate/metrics.py:26-52 -- same positional order as the reference: per class
    (smooth + 2 n(A&B)) / (smooth + n(A) + n(B)), NaN where the class is in neither volume (or not in y_true
    with skip_if_no_y); classes = unique(y_true) when n_classes is None. Host integers (the parity metric)."""
    y_true = y_true.cpu().numpy() if torch.is_tensor(y_true) else np.asarray(y_true)
    y_pred = y_pred.cpu().numpy() if torch.is_tensor(y_pred) else np.asarray(y_pred)
    classes = np.unique(y_true) if n_classes is None else np.arange(max(2, n_classes))
    if ignore_zero:
        classes = classes[np.where(classes != 0)]
    out = np.full(classes.shape, np.nan, dtype=np.float32)
    for i, c in enumerate(classes):
        s1 = y_true == c
        if skip_if_no_y and not np.any(s1):
            continue
        s2 = y_pred == c
        if np.any(s1) or np.any(s2):
            out[i] = dice(s1, s2, smooth=smooth)
    return out
