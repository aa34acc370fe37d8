"""
Callers on either side of the hot path: volume files, random
views, and the train-time plane sampler that feeds UNet.train_step. Volumes live on the GPU and
every plane is cut by the HIP sampling kernel; only the few random numbers per slice are host work.

Reference behaviour restated here:
  training batch sampler .... mpunet/sequences/isotrophic_live_view_sequence_2d.py:119-216
  fg balancing .............. mpunet/sequences/isotrophic_live_view_sequence.py:98-128
  random views .............. mpunet/interpolation/sample_grid.py:133-173
  views.npz / dim / span .... mpunet/preprocessing/data_preparation_funcs.py:116-154, mpunet/image/auditor.py:108-112,199-209
Volume files: NIfTI-1 (.nii / .nii.gz, read natively -- nifti.py) as in a reference project folder, or .npz files with
keys `image` [X,Y,Z(,C)], optional `labels` [X,Y,Z], optional `affine` [4,4].
"""
import os
import numpy as np
import torch

from .interpolation import Volume, ViewGeometry, sample_view


def load_volume_file(path):
    if path.endswith((".nii", ".nii.gz")):
        from .formats import load_nifti
        img, aff = load_nifti(path)
        return img, None, aff
    with np.load(path) as z:
        d = {k: z[k] for k in z.files}
    img = d.get("image", d.get("arr_0"))
    if img is None:
        raise ValueError("%s: no 'image' array" % path)
    if img.ndim == 3:
        img = img[..., None]
    return img.astype(np.float32), d.get("labels"), d.get("affine", np.eye(4))


def load_label_file(path):
    """Label volume [X,Y,Z] u8 of a NIfTI or .npz file (key `labels`, else the first array)."""
    if path.endswith((".nii", ".nii.gz")):
        from .formats import load_nifti_labels
        return load_nifti_labels(path)
    with np.load(path) as z:
        return z["labels"] if "labels" in z.files else z[z.files[0]]


def list_volume_files(base_dir, img_subdir="images"):
    d = os.path.join(base_dir, img_subdir)
    if not os.path.isdir(d):
        return []
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith((".npz", ".nii", ".nii.gz")))


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


def plane_basis_fast(view, noise=None):
    """
    [u v n] (row-major 3x3, columns u, v, n) of interpolation.plane_basis in plain Python floats: the same
    construction (normalise, add noise, normalise, |.| rule for small x/y, u = R(axis, -90 deg) n, v = n x u)
    without the reference's float32 round trips - agrees to ~1e-7, which is far below the random `noise_sd`
    the training planes carry anyway. ~10x cheaper than the NumPy version (the sampler's host cost).
    """
    import math
    n0, n1, n2 = float(view[0]), float(view[1]), float(view[2])
    r = math.sqrt(n0 * n0 + n1 * n1 + n2 * n2)
    n0, n1, n2 = n0 / r, n1 / r, n2 / r
    if noise is not None:
        n0 += float(noise[0]); n1 += float(noise[1]); n2 += float(noise[2])
        r = math.sqrt(n0 * n0 + n1 * n1 + n2 * n2)
        n0, n1, n2 = n0 / r, n1 / r, n2 / r
    if n0 < 0.2 and n1 < 0.2:
        n0, n1 = abs(n0), abs(n1)
    if abs(n0) <= 1e-8 and abs(n1) <= 1e-8:
        u = (1.0, 0.0, 0.0); v = (0.0, 1.0, 0.0)
    else:
        s0, s1, s2 = n0, n1, n2 + 1.0
        r = math.sqrt(s0 * s0 + s1 * s1 + s2 * s2)
        s0, s1, s2 = s0 / r, s1 / r, s2 / r
        a0, a1, a2 = n1 * s2 - n2 * s1, n2 * s0 - n0 * s2, n0 * s1 - n1 * s0
        r = math.sqrt(a0 * a0 + a1 * a1 + a2 * a2)
        a0, a1, a2 = a0 / r, a1 / r, a2 / r
        th = math.radians(-90.0)
        qa = math.cos(th / 2.0); sn = math.sin(th / 2.0)
        b, c, d = -a0 * sn, -a1 * sn, -a2 * sn
        aa, bb, cc, dd = qa * qa, b * b, c * c, d * d
        bc, ad, ac, ab, bd, cd = b * c, qa * d, qa * c, qa * b, b * d, c * d
        R = ((aa + bb - cc - dd, 2 * (bc + ad), 2 * (bd - ac)),
             (2 * (bc - ad), aa + cc - bb - dd, 2 * (cd + ab)),
             (2 * (bd + ac), 2 * (cd - ab), aa + dd - bb - cc))
        u = tuple(R[i][0] * n0 + R[i][1] * n1 + R[i][2] * n2 for i in range(3))
        v = (n1 * u[2] - n2 * u[1], n2 * u[0] - n0 * u[2], n0 * u[1] - n1 * u[0])
    return [u[0], v[0], n0, u[1], v[1], n1, u[2], v[2], n2]


class _VolCache:
    """Per-volume constants of the sampling call (built once): ctypes pieces and device-side background values."""

    def __init__(self, vol, dim, span):
        import ctypes as C
        from . import _lib
        self.shape = (C.c_int32 * 4)(*[int(v) for v in vol.image.shape])
        g = _lib.ViewGeom()
        for k in range(3):
            g.vol_axis[k] = _lib.make_axis(vol.axes[k])
        rot = np.eye(3) if vol.rot_mat is None else np.asarray(vol.rot_mat, np.float64)
        g.rot[:] = rot.ravel().tolist()
        g.has_rot = 0 if vol.rot_mat is None else 1
        hd = span // 2
        g.dim, g.n_planes = int(dim), 1
        g.g_start, g.g_step = float(-hd), float((hd - (-hd)) / float(dim - 1))
        self.geom = g
        bg = torch.tensor(vol.bg_value, dtype=torch.float64, device=vol.device)
        if vol.scaler is not None:                           # planes come out scaled: compare with the scaled value
            c, s_ = vol.scaler
            bg = (bg - torch.tensor(np.asarray(c, np.float64), device=vol.device)) / \
                torch.tensor(np.asarray(s_, np.float64), device=vol.device)
        self.bg_scaled = bg.float().contiguous()
        self.bg_scaled_ptr = self.bg_scaled.data_ptr()
        self.ptrs = [_lib.ptr(vol.image), _lib.ptr(vol.labels), _lib.ptr(vol._axes_dev[0]), _lib.ptr(vol._axes_dev[1]),
                     _lib.ptr(vol._axes_dev[2]), _lib.ptr(vol._bg), _lib.ptr(vol._center), _lib.ptr(vol._scale)]


class TrainSampler:
    """
    Random-plane batch sampler (training half of IsotrophicLiveViewSequence2D). Every plane is cut by the HIP
    sampling kernel straight into the batch tensors; the accept / reject statistics of a candidate (classes
    present, not-all-background) come back in one 8-byte read (mpu_plane_stats), which is the only host
    synchronisation per candidate.
    """

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
        self._cache = {}
        self._fg_mask = 0
        for c in self.fg_classes:
            self._fg_mask |= 1 << int(c)
        self._views_l = [tuple(float(a) for a in v) for v in self.views]

    def _vc(self, vi):
        c = self._cache.get(vi)
        if c is None:
            c = self._cache[vi] = _VolCache(self.volumes[vi], self.dim, self.span)
        return c

    def _issue_cut(self, vi, X, Y, slot, off_dev, stats):
        """Enqueue one candidate plane of volume vi into X[slot], Y[slot] and its statistics into stats[slot] (class mask,
        not-all-background flag): two launches, NO host synchronisation. off_dev: f64 [>= slot + 1] scratch, stats: i32 [., 2]."""
        import ctypes as C
        from . import _lib
        vol, vc = self.volumes[vi], self._vc(vi)
        view = self._views_l[self.rng.randint(0, len(self._views_l))]
        half = self.span // 2
        off = self.rng.uniform(-half, half)
        noise = self.rng.normal(scale=self.noise_sd, size=3) if self.noise_sd else None
        g = vc.geom
        g.basis[:] = plane_basis_fast(view, noise)
        od = off_dev[slot:slot + 1]
        od.fill_(off)                                         # scalar fill: no host->device copy of a tensor
        st = _lib.stream_ptr()
        p = vc.ptrs
        xs, ys = X[slot], Y[slot]
        _lib.call("mpu_sample_view_planes", p[0], p[1], vc.shape, p[2], p[3], p[4], C.byref(g), _lib.ptr(od),
                  p[5], vol.bg_class, p[6], p[7], _lib.ptr(xs), _lib.ptr(ys), st)
        _lib.call("mpu_plane_stats", _lib.ptr(ys), _lib.ptr(xs), self.dim * self.dim, vol.n_channels,
                  _lib.ptr(vc.bg_scaled), _lib.ptr(stats[slot]), st)

    def _cut(self, vi, X, Y, slot, off_dev, stats):
        """One candidate plane of volume vi into X[slot], Y[slot]; returns (class mask, not-all-bg flag) -- synchronises."""
        if stats.ndim == 1:                                   # (the one-candidate form: off_dev [1], stats [2])
            self._issue_cut(vi, X[slot:slot + 1], Y[slot:slot + 1], 0, off_dev, stats.reshape(1, 2))
            m, nb = stats.tolist()
        else:
            self._issue_cut(vi, X, Y, slot, off_dev, stats)
            m, nb = stats[slot].tolist()
        return m, bool(nb)

    def __call__(self):
        """One batch. Round 5: the candidates of ALL undecided slots are cut before the host reads their statistics -- one
        read (one stream synchronisation) per ROUND instead of one per candidate: a batch whose first candidates are all
        accepted costs a single round trip, which is what lets the sampler run on a side stream beside a train step without
        queueing behind its kernels 2 x B times (pipeline.TrainPipeline). The accept / reject rule and its sequential state
        (foreground balance, classes seen) are the reference's, applied slot by slot in order; candidates are i.i.d. draws, so
        deciding a slot on a candidate drawn before the earlier slots were settled changes nothing statistically. With no
        rejection the random stream is consumed in the same order as the one-candidate-at-a-time form."""
        B, d = self.batch_size, self.dim
        vol0 = self.volumes[0]
        dev = vol0.device
        X = torch.empty((B, d, d, vol0.n_channels), dtype=torch.float32, device=dev)
        Y = torch.empty((B, d, d), dtype=torch.uint8, device=dev)
        off_dev = torch.empty(B, dtype=torch.float64, device=dev)
        stats = torch.empty((B, 2), dtype=torch.int32, device=dev)
        has_fg, fg_vec = 0, 0                                 # fg_vec: bit mask of the fg classes seen so far
        nfg = len(self.fg_classes)
        vis = [None] * B
        tries = [0] * B
        fresh = [False] * B                                   # slot holds a candidate not yet judged
        first = 0                                             # first undecided slot
        self.rounds = 0                                       # (diagnostic: host reads of this batch)
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        off_host = getattr(self, "_off_host", None)
        if off_host is None or off_host.numel() != B:            # pinned: the round's offsets go up as ONE asynchronous copy
            off_host = self._off_host = torch.empty(B, dtype=torch.float64).pin_memory()
        xp, yp, op, sp = X.data_ptr(), Y.data_ptr(), off_dev.data_ptr(), stats.data_ptr()
        xs, ys = X.stride(0) * 4, Y.stride(0)
        half = self.span // 2
        nviews = len(self._views_l)
        stp = _lib.stream_ptr()
        while first < B:
            # (re)cut what is missing. Round 6: host work first -- the draws of every missing slot in slot order (the same random
            # stream as one candidate at a time), their plane bases, the offsets into a pinned buffer -- then ONE copy of the
            # offsets and one library call per candidate (mpu_sample_plane_stats: sample + statistics): about half the host time
            # per candidate, which is what bounded the mp-train loop once the train step had reached 2.4 ms
            todo = []
            for slot in range(first, B):
                if not fresh[slot]:
                    if vis[slot] is None:
                        vis[slot] = self.rng.randint(0, len(self.volumes))
                    tries[slot] += 1
                    view = self._views_l[self.rng.randint(0, nviews)]
                    off = self.rng.uniform(-half, half)
                    noise = self.rng.normal(scale=self.noise_sd, size=3) if self.noise_sd else None
                    off_host[slot] = off
                    todo.append((slot, vis[slot], plane_basis_fast(view, noise)))
                    fresh[slot] = True
            off_dev.copy_(off_host, non_blocking=True)
            for slot, vi, basis in todo:
                vol, vc = self.volumes[vi], self._vc(vi)
                g = vc.geom
                g.basis[:] = basis
                p = vc.ptrs
                _lib.check(lib.mpu_sample_plane_stats(p[0], p[1], vc.shape, p[2], p[3], p[4], C.byref(g), op + 8 * slot, p[5],
                                                      vol.bg_class, p[6], p[7], xp + slot * xs, yp + slot * ys, vc.bg_scaled_ptr,
                                                      sp + 8 * slot, stp), "mpu_sample_plane_stats")
            st = stats.tolist()                               # the round's one synchronisation
            self.rounds += 1
            while first < B:                                  # judge in slot order until a candidate is rejected
                slot = first
                m, nonbg = st[slot][0], bool(st[slot][1])
                fresh[slot] = False
                present = m & self._fg_mask
                last = tries[slot] >= self.max_tries
                fg_try = fg_vec
                if self.force_all_fg and not last:
                    new = fg_vec | present
                    missing = nfg - bin(new).count("1")
                    if not (missing == 0 or missing < (B - slot)):
                        break                                 # rejected: this slot gets a new candidate next round
                    fg_try = new
                if present:
                    ok, inc = True, 1
                elif (self.n_fg_slices - has_fg) < (B - slot):
                    ok, inc = True, 0
                else:
                    ok, inc = False, 0
                if (ok or last) and (last or nonbg):
                    has_fg += inc
                    fg_vec = fg_try
                    first += 1
                else:
                    break
        ws = [self.sample_weights[vi] for vi in vis]
        bgs = [list(self.volumes[vi].bg_value) for vi in vis]
        x, y = X, Y
        w = torch.tensor(ws, dtype=torch.float32, device=dev)
        for aug in self.augmenters:
            x, y, w = aug(x, y, bgs, w)
        return x, y.reshape(B, -1, 1), w

    def __iter__(self):
        while True:
            yield self()
