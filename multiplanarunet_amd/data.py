"""
Callers on either side of the hot path: volume files, random
views, and the train-time plane sampler that feeds UNet.train_step. Volumes live on the GPU and
every plane is cut by the HIP sampling kernel; only the few random numbers per slice are host work.

Reference behaviour restated here:
  training batch sampler .... mpunet/sequences/isotrophic_live_view_sequence_2d.py:119-216
  fg balancing .............. mpunet/sequences/isotrophic_live_view_sequence.py:98-128
  random views .............. mpunet/interpolation/sample_grid.py:133-173
  views.npz / dim / span .... mpunet/preprocessing/data_preparation_funcs.py:116-154, mpunet/image/auditor.py:108-112,199-209
NIfTI I/O (nibabel) is outside the path and not available here: volumes are .npz files with keys
`image` [X,Y,Z(,C)], optional `labels` [X,Y,Z], optional `affine` [4,4].
"""
import os
import numpy as np
import torch

from .interpolation import Volume, ViewGeometry, sample_view


def load_volume_file(path):
    if path.endswith((".nii", ".nii.gz")):
        raise NotImplementedError("NIfTI input needs nibabel, which is outside the accelerated path and "
                                  "not installed here; convert to .npz (image, labels, affine)")
    with np.load(path) as z:
        d = {k: z[k] for k in z.files}
    img = d.get("image", d.get("arr_0"))
    if img is None:
        raise ValueError("%s: no 'image' array" % path)
    if img.ndim == 3:
        img = img[..., None]
    return img.astype(np.float32), d.get("labels"), d.get("affine", np.eye(4))


def list_volume_files(base_dir, img_subdir="images"):
    d = os.path.join(base_dir, img_subdir)
    if not os.path.isdir(d):
        return []
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".npz"))


def make_toy_volume(size=64, seed=0):
    """A 3-class synthetic volume (background, ellipsoid, box) in the spirit of `mp toy_data`."""
    rng = np.random.RandomState(seed)
    g = np.mgrid[:size, :size, :size].astype(np.float32)
    c1 = size * (0.35 + 0.3 * rng.rand(3)); r1 = size * (0.12 + 0.1 * rng.rand(3))
    ell = (((g[0] - c1[0]) / r1[0]) ** 2 + ((g[1] - c1[1]) / r1[1]) ** 2 + ((g[2] - c1[2]) / r1[2]) ** 2) <= 1
    c2 = size * (0.3 + 0.4 * rng.rand(3)); h2 = size * (0.08 + 0.08 * rng.rand(3))
    box = (abs(g[0] - c2[0]) < h2[0]) & (abs(g[1] - c2[1]) < h2[1]) & (abs(g[2] - c2[2]) < h2[2])
    lab = np.zeros((size,) * 3, np.uint8)
    lab[ell] = 1
    lab[box] = 2
    img = 0.3 * np.sin(g[0] / size * 3) + 0.2 * np.cos(g[1] / size * 5) + 0.05 * rng.randn(size, size, size)
    img = img + 0.8 * (lab == 1) + 1.5 * (lab == 2)
    return img[..., None].astype(np.float32), lab, np.eye(4)


def as_volume(image, labels, affine, bg_value="1pct", scaler="RobustScaler", device="cuda", identifier="volume"):
    """Host prep the reference does lazily per ImagePair (image_pair.py:300-341,469-484), then upload."""
    C = image.shape[-1]
    if isinstance(bg_value, str) and bg_value.endswith("pct"):
        pct = int(bg_value[:-3])
        bg = [float(np.percentile(image[..., c], pct)) for c in range(C)]
    elif isinstance(bg_value, (list, tuple, np.ndarray)):
        bg = [float(b) for b in bg_value]
    else:
        bg = [float(bg_value or 0.0)] * C
    sc = None
    if scaler:
        if scaler != "RobustScaler":
            raise NotImplementedError("only scaler: RobustScaler (YAML default) or Null")
        sc = Volume.fit_robust_scaler(image)
    return Volume(image, labels, affine, bg_value=bg, scaler=sc, device=device, identifier=identifier)


def random_views(n, min_angle_deg=60.0, seed=None):
    """n unit vectors with z >= 0 and pairwise angles above a threshold that relaxes by 1 degree per failure."""
    rng = np.random.RandomState(seed)
    thr = float(min_angle_deg)
    while True:
        v = rng.normal(size=(n, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        v[:, 2] = np.abs(v[:, 2])
        cosines = np.clip(v @ v.T, -1, 1)
        ang = np.rad2deg(np.arccos(cosines))[np.triu_indices(n, 1)]
        if n < 2 or np.all(ang > thr):
            return v
        thr -= 1.0


def audit_dim_and_span(volumes, min_dim=128, max_dim=512):
    """real_space_span = 75th percentile of the real-space extents; sample dim at the median resolution,
    rounded to a multiple of 16 in [min_dim, max_dim]."""
    ext, res = [], []
    for v in volumes:
        pix = np.linalg.norm(v.affine[:3, :3], axis=0)
        ext += list(np.array(v.image.shape[:3]) * pix)
        res += list(pix)
    span = float(np.percentile(ext, 75))
    dim = int(np.clip(int(np.ceil(span / np.median(res) / 16.0) * 16), min_dim, max_dim))
    return dim, span


class TrainSampler:
    """Random-plane batch sampler (training half of IsotrophicLiveViewSequence2D)."""

    def __init__(self, volumes, views, dim, real_space_span, batch_size, n_classes, noise_sd=0.1,
                 fg_batch_fraction=0.5, force_all_fg="auto", sample_weights=None, seed=None, max_tries=10,
                 augmenters=None):
        self.volumes = list(volumes)
        self.views = np.asarray(views, float)
        self.dim, self.span = int(dim), real_space_span
        self.batch_size, self.n_classes = int(batch_size), int(n_classes)
        self.noise_sd = float(noise_sd)
        self.fg_classes = np.arange(1, n_classes) if n_classes > 1 else np.array([1])
        self.n_fg_slices = int(np.ceil(batch_size * fg_batch_fraction))
        self.force_all_fg = (batch_size > len(self.fg_classes)) if force_all_fg == "auto" else bool(force_all_fg)
        self.sample_weights = sample_weights or [1.0] * len(self.volumes)
        self.rng = np.random.RandomState(seed)
        self.max_tries = max_tries
        self.augmenters = list(augmenters or [])      # applied after scaling (isotrophic_live_view_sequence_2d.py:203-208)

    def _one_plane(self, vol):
        view = self.views[self.rng.randint(0, len(self.views))]
        half = self.span // 2
        off = self.rng.uniform(-half, half)
        noise = self.rng.normal(scale=self.noise_sd, size=3) if self.noise_sd else None
        g = ViewGeometry(view, self.dim, self.span, "same", noise=noise)
        g.offsets = np.array([off])
        g.n_planes = 1
        X, y = sample_view(vol, g, want_labels=True)
        return X[0], y[0]

    def __call__(self):
        xs, ys, ws, bgs = [], [], [], []
        has_fg, fg_vec = 0, np.zeros(len(self.fg_classes), bool)
        B = self.batch_size
        for _ in range(B):
            vi = self.rng.randint(0, len(self.volumes))
            vol = self.volumes[vi]
            for t in range(1, self.max_tries + 1):
                x, y = self._one_plane(vol)
                present = np.isin(self.fg_classes, torch.unique(y).cpu().numpy())
                last = t == self.max_tries
                if self.force_all_fg and not last:
                    new = fg_vec | present
                    if not (new.all() or (~new).sum() < (B - len(ys))):
                        continue
                    fg_vec_try = new
                else:
                    fg_vec_try = fg_vec
                if present.any():
                    ok, inc = True, 1
                elif (self.n_fg_slices - has_fg) < (B - len(ys)):
                    ok, inc = True, 0
                else:
                    ok, inc = False, 0
                if ok or last:
                    bg = torch.tensor(vol.bg_value, device=x.device)
                    if vol.scaler is not None:           # compare against the scaled background value
                        c, s = vol.scaler
                        bg = ((bg.double() - torch.tensor(c, device=x.device)) / torch.tensor(s, device=x.device)).float()
                    if last or bool((~torch.isclose(x, bg.expand_as(x))).any()):
                        has_fg += inc
                        fg_vec = fg_vec_try
                        break
            xs.append(x); ys.append(y); ws.append(self.sample_weights[vi]); bgs.append(list(vol.bg_value))
        x = torch.stack(xs)
        y = torch.stack(ys)
        w = torch.tensor(ws, dtype=torch.float32, device=x.device)
        for aug in self.augmenters:
            x, y, w = aug(x, y, bgs, w)
        return x, y.reshape(B, -1, 1), w

    def __iter__(self):
        while True:
            yield self()
