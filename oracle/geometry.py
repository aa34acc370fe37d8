"""
ORACLE -- TEST INFRASTRUCTURE ONLY. Never imported by the product path
(multiplanarunet_amd/), only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.

NumPy restatement of the reference's (mpunet 0.2.12) view geometry, plane
resampling, nearest back-mapping and multi-view fusion. Every function cites
the reference file:line it restates. Pinned against golden vectors produced by
the reference's own unmodified NumPy code (oracle/gen_golden.py ->
tests/golden/geometry_golden.npz); see tests/test_oracle_golden.py.

dtype conventions follow what the reference does when run under NumPy >= 2
(NEP 50 promotion), which is how the goldens were produced.
"""
import itertools
import numpy as np


# --------------------------------------------------------------------------- #
# linalg helpers -- mpunet/interpolation/linalg.py:5-51
# --------------------------------------------------------------------------- #
def rotation_matrix(axis, angle_deg):
    """mpunet/interpolation/linalg.py:33-51 (get_rotation_matrix)."""
    theta = np.deg2rad(angle_deg)
    axis = np.asarray(axis).ravel()
    axis = axis / np.linalg.norm(axis)
    a = np.cos(theta / 2.0)
    b, c, d = -axis * np.sin(theta / 2.0)
    aa, bb, cc, dd = a * a, b * b, c * c, d * d
    bc, ad, ac, ab, bd, cd = b * c, a * d, a * c, a * b, b * d, c * d
    return np.array([[aa + bb - cc - dd, 2 * (bc + ad), 2 * (bd - ac)],
                     [2 * (bc - ad), aa + cc - bb - dd, 2 * (cd + ab)],
                     [2 * (bd + ac), 2 * (cd - ab), aa + dd - bb - cc]])


# --------------------------------------------------------------------------- #
# sample_plane_at -- mpunet/interpolation/sample_grid.py:192-244
# --------------------------------------------------------------------------- #
def plane_basis(norm_vector, noise=None):
    """Basis [u v n_hat] of the sampling plane; sample_grid.py:194-221."""
    n_hat = np.array(norm_vector, np.float32)
    n_hat /= np.linalg.norm(n_hat)
    if noise is None:
        noise = np.zeros(3)                     # np.random.normal(scale=0.)
    n_hat += noise
    n_hat /= np.linalg.norm(n_hat)
    if np.all(n_hat[:-1] < 0.2):
        n_hat[:-1] = np.abs(n_hat[:-1])
    if np.all(np.isclose(n_hat[:-1], 0)):
        u = np.array([1, 0, 0])
        v = np.array([0, 1, 0])
    else:
        nhat_vs = n_hat.copy()
        nhat_vs[-1] = nhat_vs[-1] + 1
        nhat_vs /= np.linalg.norm(nhat_vs)
        u = rotation_matrix(np.cross(n_hat, nhat_vs), -90).dot(n_hat)
        v = np.cross(n_hat, u)
    return np.column_stack((u, v, n_hat))


def sample_plane_at(norm_vector, sample_dim, real_space_span,
                    offset_from_center, noise=None):
    """
    sample_grid.py:192-244 with test_mode=True.
    Returns real_grid [3, dim, dim, 1] f64, g [dim] f64, inv(basis) [3,3].
    """
    basis = plane_basis(norm_vector, noise)
    hd = real_space_span // 2                   # floor, sample_grid.py:227
    g = np.linspace(-hd, hd, sample_dim)
    # np.mgrid[-hd:hd:dim*1j] == arange(dim) * step + start
    step = (hd - (-hd)) / float(sample_dim - 1)
    ax = np.arange(sample_dim, dtype=np.float64) * step + (-hd)
    gx, gy = np.meshgrid(ax, ax, indexing="ij")
    pts = np.stack([gx.ravel(), gy.ravel(),
                    np.full(gx.size, float(offset_from_center))], axis=1)
    real = basis.dot(pts.T).T
    real_grid = np.empty((3, sample_dim, sample_dim, 1))
    for i in range(3):
        real_grid[i] = real[:, i].reshape(sample_dim, sample_dim, 1)
    return real_grid, g, np.linalg.inv(basis)


# --------------------------------------------------------------------------- #
# voxel axes -- mpunet/interpolation/sample_grid.py:63-98
# --------------------------------------------------------------------------- #
def voxel_axes_real_space(shape3, affine):
    """get_voxel_axes_real_space(return_basis=True); sample_grid.py:63-98."""
    x, y, z = shape3
    axes = [np.arange(n, dtype=np.float32) - (n - 1) / 2 for n in (x, y, z)]
    basis = np.asarray(affine)[:-1, :-1]
    pixdims = np.linalg.norm(basis, axis=0)
    transform = np.diag(pixdims)
    if np.any(~np.isclose(transform, basis)):
        rot_mat = transform.dot(np.linalg.inv(basis))
    else:
        rot_mat = None
    axes = [axes[i] * transform[i, i] for i in range(3)]
    return tuple(axes), rot_mat


# --------------------------------------------------------------------------- #
# RegularGridInterpolator -- mpunet/interpolation/regular_grid_interpolator.py
# --------------------------------------------------------------------------- #
def rgi_find_indices(xi, grid):
    """_find_indices, regular_grid_interpolator.py:252-270. xi: [3, N]."""
    indices, norm_distances = [], []
    oob = np.zeros(xi.shape[1], dtype=bool)
    for x, g in zip(xi, grid):
        i = np.searchsorted(g, x) - 1
        i[i < 0] = 0
        i[i > g.size - 2] = g.size - 2
        indices.append(i)
        norm_distances.append((x - g[i]) / (g[i + 1] - g[i]))
        oob |= x < g[0]
        oob |= x > g[-1]
    return indices, norm_distances, oob


def rgi_linear(values, grid, xi, fill_value):
    """
    method="linear", regular_grid_interpolator.py:152-217. values [X,Y,Z];
    xi [3, N]; returns float64 [N] (f32 values x f64 weights), OOB -> fill.
    """
    idx, nd, oob = rgi_find_indices(xi, grid)
    out = 0.
    for edge in itertools.product(*[[i, i + 1] for i in idx]):
        w = 1.
        for ei, i, yi in zip(edge, idx, nd):
            w = w * np.where(ei == i, 1 - yi, yi)
        out = out + np.asarray(values[edge]) * w
    out[oob] = np.array(fill_value).astype(np.float32)
    return out


def rgi_nearest(values, grid, xi, fill_value):
    """
    method="nearest", regular_grid_interpolator.py:219-223. values [X,Y,Z,...]
    (trailing dims broadcast); OOB rows overwritten with fill (:199-200).
    """
    idx, nd, oob = rgi_find_indices(xi, grid)
    sel = tuple(np.where(yi <= .5, i, i + 1) for i, yi in zip(idx, nd))
    out = values[sel]
    out[oob] = fill_value
    return out


# --------------------------------------------------------------------------- #
# ViewInterpolator -- mpunet/interpolation/view_interpolator.py:17-147
# --------------------------------------------------------------------------- #
def view_interpolate(image, labels, affine, bg_value, bg_class, real_grid):
    """
    ViewInterpolator(image, labels, affine, bg_value, bg_class)(real_grid);
    view_interpolator.py:54-133. image [X,Y,Z,C] f32, real_grid [3,d,d,1].
    Returns (im [d,d,C] image dtype, lab [d,d] u8 or None).
    """
    C = image.shape[-1]
    if not isinstance(bg_value, (list, tuple, np.ndarray)):
        bg_value = [bg_value] * C
    axes, rot_mat = voxel_axes_real_space(image.shape[:3], affine)
    shp = real_grid[0].shape
    pts = np.stack([real_grid[i].ravel() for i in range(3)], axis=1)
    if rot_mat is not None:
        pts = rot_mat.dot(pts.T).T              # apply_rotation, :54-60
    xi = pts.T
    out_shape = real_grid[0].squeeze().shape
    im = np.zeros(out_shape + (C,), dtype=image.dtype)
    for c in range(C):
        im[..., c] = rgi_linear(image[..., c], axes, xi,
                                bg_value[c]).reshape(shp).squeeze()
    lab = None
    if labels is not None:
        lab = rgi_nearest(labels, axes, xi,
                          np.array(bg_class).astype(np.uint8))
        lab = lab.reshape(shp).squeeze().astype(np.uint8)
    return im, lab


def scaler_transform(im, center, scale):
    """
    MultiChannelScaler.transform with sklearn RobustScaler semantics
    (mpunet/preprocessing/scaling.py:75-89): per channel, in place on the f32
    plane: X -= center_ (f64 op, f32 store); X /= scale_ (f64 op, f32 store).
    center/scale None -> identity.
    """
    if center is None:
        return im
    out = np.empty_like(im)
    for c in range(im.shape[-1]):
        x = im[..., c].astype(np.float32)
        x = (x.astype(np.float64) - center[c]).astype(np.float32)
        x = (x.astype(np.float64) / scale[c]).astype(np.float32)
        out[..., c] = x
    return out


# --------------------------------------------------------------------------- #
# get_view_from -- mpunet/sequences/isotrophic_live_view_sequence_2d.py:29-117
# --------------------------------------------------------------------------- #
def view_offsets(sample_dim, real_space_span, extra=20):
    """n_planes='same+<extra>' branch, isotrophic_live_view_sequence_2d.py:47-62."""
    sample_res = real_space_span / (sample_dim - 1)
    n_planes = sample_dim + extra
    bounds = (real_space_span + (extra * sample_res)) / 2
    return np.linspace(-bounds, bounds, n_planes)


def get_view_from(image, labels, affine, view, sample_dim, real_space_span,
                  bg_value=0.0, bg_class=0, center=None, scale=None, extra=20):
    """
    get_view_from(image, view, 'same+20'); returns Xs [d,d,P,C], ys [d,d,P] or
    None, (g, g, offsets), inv_basis.
    """
    offsets = view_offsets(sample_dim, real_space_span, extra)
    P = offsets.shape[0]
    C = image.shape[-1]
    Xs = np.empty((sample_dim, sample_dim, P, C), dtype=image.dtype)
    ys = None if labels is None else np.empty((sample_dim, sample_dim, P),
                                              dtype=labels.dtype)
    g = inv_basis = None
    for p, off in enumerate(offsets):
        grid, g, inv_basis = sample_plane_at(view, sample_dim,
                                             real_space_span, off)
        im, lab = view_interpolate(image, labels, affine, bg_value, bg_class,
                                   grid)
        Xs[:, :, p, :] = scaler_transform(im, center, scale)
        if ys is not None:
            ys[:, :, p] = lab
    return Xs, ys, (g, g, offsets), inv_basis


# --------------------------------------------------------------------------- #
# voxel grid + back-mapping -- sample_grid.py:101-130, fuse_and_predict.py:92-137
# --------------------------------------------------------------------------- #
def voxel_grid_real_space(shape3, affine):
    """get_voxel_grid_real_space; sample_grid.py:101-130. -> [3,X,Y,Z] f64."""
    X, Y, Z = shape3
    A = np.asarray(affine)[:-1, :-1]
    gi, gj, gk = np.meshgrid(np.arange(X), np.arange(Y), np.arange(Z),
                             indexing="ij")
    pts = np.stack([gi.ravel(), gj.ravel(), gk.ravel()], axis=1)
    real = A.dot(pts.T).T
    real = real - np.mean(real, axis=0)
    out = np.empty((3, X, Y, Z))
    for i in range(3):
        out[i] = real[:, i].reshape(X, Y, Z)
    return out


def map_real_space_pred(pred, grid, inv_basis, voxel_grid):
    """
    map_real_space_pred(method='nearest'); fuse_and_predict.py:92-137.
    pred [d,d,P,K] f32 on axes grid=(g,g,offsets); -> mapped [X,Y,Z,K] f32,
    OOB voxels -> [1,0,...,0] (:99-100).
    """
    K = pred.shape[-1]
    fill = np.zeros(K, dtype=np.float32)
    fill[0] = 1.0
    shp = voxel_grid[0].shape
    pts = np.stack([voxel_grid[i].ravel() for i in range(3)], axis=1)
    pts = inv_basis.dot(pts.T).T
    out = rgi_nearest(pred, grid, pts.T, fill)
    return out.reshape(shp + (K,)).astype(pred.dtype)


# --------------------------------------------------------------------------- #
# fusion -- mpunet/models/fusion_model.py:38-39, mpunet/bin/predict.py:349-366
# --------------------------------------------------------------------------- #
def fusion_layer(x, W, b):
    """FusionLayer.call: softmax(sum_v W[v,k]*x[n,v,k] + b[0,k]); f32."""
    z = np.sum(W[None].astype(np.float32) * x.astype(np.float32), axis=1)
    z = z + np.asarray(b, np.float32).reshape(1, -1)
    z = z - z.max(axis=-1, keepdims=True)
    e = np.exp(z)
    return (e / e.sum(axis=-1, keepdims=True)).astype(np.float32)


def merge_multi_view_preds(combined, W, b, sum_fusion=False):
    """
    merge_multi_view_preds; predict.py:349-366. combined [V,X,Y,Z,K] ->
    (merged [X,Y,Z,K] f32, merged_map [X,Y,Z] u8).
    """
    d = combined.shape
    if not sum_fusion:
        x = np.moveaxis(combined, 0, -2).reshape((-1, d[0], d[-1]))
        merged = fusion_layer(x, W, b).reshape(d[1], d[2], d[3], d[4])
    else:
        merged = np.sum(combined, axis=0)
    return merged, pred_to_class(merged)


def pred_to_class(t):
    """utils.pred_to_class multi-class branch; mpunet/utils/utils.py:326-328."""
    return t.argmax(-1).astype(np.uint8)


def dice_all(y_true, y_pred, smooth=1.0, n_classes=None, ignore_zero=True, skip_if_no_y=False):
    """mpunet/evaluate/metrics.py:13-52 (dice, dice_all), same argument order."""
    classes = np.unique(y_true) if n_classes is None else np.arange(max(2, n_classes))
    if ignore_zero:
        classes = classes[classes != 0]
    out = np.full(classes.shape, np.nan, dtype=np.float32)
    for i, c in enumerate(classes):
        s1 = (y_true == c).ravel()
        if skip_if_no_y and not s1.any():
            continue
        s2 = (y_pred == c).ravel()
        if s1.any() or s2.any():
            out[i] = (smooth + 2 * np.logical_and(s1, s2).sum()) / \
                     (smooth + s1.sum() + s2.sum())
    return out


def multi_view_predict(image, affine, views, sample_dim, real_space_span,
                       predict_fn, W, b, bg_value=0.0, center=None, scale=None,
                       sum_fusion=False):
    """
    _multi_view_predict_on + merge_multi_view_preds (predict.py:294-366) for an
    arbitrary per-view predictor predict_fn(X[P,d,d,C]) -> [P,d,d,K].
    """
    vg = voxel_grid_real_space(image.shape[:3], affine)
    combined = []
    for view in views:
        Xs, _, grid, inv_basis = get_view_from(
            image, None, affine, view, sample_dim, real_space_span,
            bg_value=bg_value, center=center, scale=scale)
        pred = predict_fn(np.moveaxis(Xs, 2, 0))          # predict_volume :81-89
        pred = np.moveaxis(pred, 0, 2)
        combined.append(map_real_space_pred(pred, grid, inv_basis, vg))
    combined = np.stack(combined, 0)
    return merge_multi_view_preds(combined, W, b, sum_fusion) + (combined,)
