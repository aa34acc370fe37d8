"""
GPU parity: HIP resampling / back-mapping / fusion kernels (through the C ABI)
against the reference goldens and the oracle. Integer / index work is held to EXACT equality (round 5; rounds 1-4
allowed 1e-3 of the voxels to differ and observed 0): sampled image values bit for bit (fp64 accumulate -> f32 as the
reference computes them), nearest labels and back-mapped vectors identical, ties as NumPy breaks them. Floating point:
fusion probabilities atol 2e-6; a fused LABEL (argmax of f32 sums whose summation order differs between NumPy and the
kernel) may differ only where the oracle's two largest values are closer than that float band -- those voxels are
counted, everything outside the band must be identical.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

AFFS = ("ident", "aniso", "rot")


def assert_labels_equal_outside_float_ties(got, ref_labels, ref_scores, band=8e-6):
    """argmax of float scores: identical wherever the reference's top-2 margin exceeds `band` (relative to the largest
    score); returns the number of differing voxels inside the band."""
    srt = np.sort(np.asarray(ref_scores, np.float64), axis=-1)
    if srt.shape[-1] < 2:                                            # one class: nothing to tie with
        assert np.array_equal(np.asarray(got), np.asarray(ref_labels))
        return 0
    margin = srt[..., -1] - srt[..., -2]
    scale = np.maximum(1.0, np.abs(srt[..., -1]))
    diff = np.asarray(got) != np.asarray(ref_labels)
    outside = diff & (margin > band * scale)
    assert not outside.any(), "%d labels differ outside the float tie band (of %d differing)" % (outside.sum(), diff.sum())
    return int(diff.sum())


def _vol(golden, an):
    from multiplanarunet_amd.interpolation import Volume
    return Volume(golden["g3_vol"], golden["g3_lab"], golden["aff_" + an], bg_value=[12.5],
                  scaler=(golden["g3_center"], golden["g3_scale"]))


@pytest.mark.parametrize("an", AFFS)
@pytest.mark.parametrize("dim", (16, 32))
def test_get_view_from_vs_reference_golden(golden, an, dim):
    from multiplanarunet_amd.interpolation import ViewSampler
    span = {16: 30.0, 32: 33.0}[dim]
    vol = _vol(golden, an)
    seq = ViewSampler(golden["views"], dim, span, n_classes=3)
    for v in golden["g3_views"]:
        key = "%s_%d_%d" % (an, dim, v)
        Xs, ys, grid, ib = seq.get_view_from(vol, golden["views"][v], "same+20")
        assert tuple(Xs.shape) == golden["g3_X_" + key].shape
        X = Xs.cpu().numpy()
        y = ys.cpu().numpy()
        np.testing.assert_array_equal(X, golden["g3_X_" + key])            # bit for bit
        np.testing.assert_array_equal(y, golden["g3_y_" + key])
        np.testing.assert_array_equal(ib, golden["g3_invb_" + key])


@pytest.mark.parametrize("an", AFFS)
def test_two_channel_planes_vs_reference_golden(golden, an):
    from multiplanarunet_amd.interpolation import Volume, ViewGeometry, sample_view
    vol = Volume(golden["g2_vol"], golden["g2_lab"], golden["aff_" + an], bg_value=list(golden["g2_bg"]))
    for pi, (v, dim, span, off) in enumerate(golden["g2_planes"]):
        g = ViewGeometry(golden["views"][int(v)], int(dim), span, "same")
        g.offsets = np.array([off]); g.n_planes = 1
        X, y = sample_view(vol, g)
        np.testing.assert_array_equal(X[0].cpu().numpy(), golden["g2_im_%s_%d" % (an, pi)])
        np.testing.assert_array_equal(y[0].cpu().numpy(), golden["g2_lab_%s_%d" % (an, pi)])


@pytest.mark.parametrize("an", AFFS)
def test_map_real_space_pred_vs_reference_golden(golden, an):
    from multiplanarunet_amd.interpolation import map_real_space_pred
    vol = _vol(golden, an)
    for v in golden["g3_views"]:
        key = "%s_16_%d" % (an, v)
        grid = (golden["g3_g_" + key], golden["g3_g_" + key], golden["g3_off_" + key])
        for K in (1, 3, 5):
            pr = torch.tensor(golden["g5_pred_%s_%d_%d" % (an, v, K)], device="cuda")
            ref = golden["g5_map_%s_%d_%d" % (an, v, K)]
            mp = map_real_space_pred(pr, grid, golden["g3_invb_" + key], vol).cpu().numpy()
            np.testing.assert_array_equal(mp, ref)                          # every voxel carries the reference's vector


@pytest.mark.parametrize("an", AFFS)
@pytest.mark.parametrize("sum_fusion", (False, True))
def test_fused_map_fuse_vs_oracle(golden, an, sum_fusion):
    """combined[V,...] from the reference goldens -> oracle merge vs the fused HIP kernel."""
    from multiplanarunet_amd.interpolation import map_and_fuse, map_accumulate, fusion_finalize
    from oracle import geometry as G
    vol = _vol(golden, an)
    K = 3
    rng = np.random.RandomState(5)
    W = rng.uniform(0.5, 1.5, (len(golden["g3_views"]), K)).astype(np.float32)
    b = rng.uniform(-0.2, 0.2, (K,)).astype(np.float32)
    combined, vps = [], []
    for v in golden["g3_views"]:
        key = "%s_16_%d" % (an, v)
        grid = (golden["g3_g_" + key], golden["g3_g_" + key], golden["g3_off_" + key])
        pr = golden["g5_pred_%s_%d_%d" % (an, v, K)]
        combined.append(golden["g5_map_%s_%d_%d" % (an, v, K)])
        vps.append((torch.tensor(np.moveaxis(pr, 2, 0).copy(), device="cuda"), grid,
                    golden["g3_invb_" + key]))
    merged_ref, map_ref = G.merge_multi_view_preds(np.stack(combined), W, b, sum_fusion)
    probs, labels = map_and_fuse(vol, vps, W, b, sum_fusion=sum_fusion)
    torch.cuda.synchronize()
    p = probs.cpu().numpy()
    assert np.abs(p - merged_ref).max() <= 2e-6                          # every voxel: no wrong back-mapping index anywhere
    n_ties = assert_labels_equal_outside_float_ties(labels.cpu().numpy(), map_ref, merged_ref)
    assert n_ties <= 4, n_ties
    # plane-sharded accumulate path (multi-GPU predict) == fused path
    z = torch.zeros_like(probs)
    for vi, (pr, grid, ib) in enumerate(vps):
        P = pr.shape[0]
        cuts = [0, 7, 20, P]
        Wv = np.ones(K, np.float32) if sum_fusion else W[vi]
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            map_accumulate(vol, pr[lo:hi].contiguous(), grid, ib, Wv, lo, hi, lo == 0, z)
    p2, l2 = fusion_finalize(z, b, sum_fusion=sum_fusion)
    np.testing.assert_allclose(p2.cpu().numpy(), p, rtol=0, atol=2e-6)
    assert assert_labels_equal_outside_float_ties(l2.cpu().numpy(), labels.cpu().numpy(), p) <= 4


@pytest.mark.parametrize("an", ("rot", "aniso"))
def test_sample_map_fuse_at_128_cubed_equals_the_oracle_exactly(golden, an):
    """the whole geometry chain at D = 128 against oracle/geometry.py (the NumPy restatement that is
    bit-exact on the reference goldens): a 128^3 x 2-channel volume under the goldens' rotated / anisotropic affine, six
    views, 148 planes of 128 x 128 each, K = 5. Sampled planes bit for bit, nearest labels identical, every back-mapped
    vector identical, fused probabilities to 2e-6 and fused labels identical outside the f32 tie band."""
    from multiplanarunet_amd.interpolation import Volume, ViewSampler, map_real_space_pred, map_and_fuse
    from oracle import geometry as G
    D, C_, K = 128, 2, 5
    rng = np.random.RandomState(128)
    g = np.indices((D, D, D)).astype(np.float32) / D
    image = np.stack([60 * np.sin(5 * g[0]) * np.cos(3 * g[1]) + 40 * g[2] + 4 * rng.randn(D, D, D),
                      50 * np.cos(4 * g[2] + g[0]) + 30 * g[1] * g[1] + 4 * rng.randn(D, D, D)], -1).astype(np.float32)
    lab = ((g[0] > .3).astype(np.uint8) + (g[1] > .5) + (g[2] > .6) + ((g[0] - .5) ** 2 + (g[1] - .5) ** 2 < .04)).astype(np.uint8)
    aff = golden["aff_" + an]
    bg = [float(np.percentile(image[..., c], 1)) for c in range(C_)]
    center, scale = Volume.fit_robust_scaler(image)
    vol = Volume(image, lab, aff, bg_value=bg, scaler=(center, scale))
    views = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [0.3, 0.5, 0.8], [0.7, -0.2, 0.6], [-0.4, 0.6, 0.55]], float)
    views /= np.linalg.norm(views, axis=1, keepdims=True)
    span = float(D) * 1.05
    seq = ViewSampler(views, D, span, n_classes=K)
    A = (rng.randn(C_, K) * 1.5).astype(np.float32)
    vg = G.voxel_grid_real_space(image.shape[:3], aff)
    W = rng.uniform(0.5, 1.5, (len(views), K)).astype(np.float32)
    b = rng.uniform(-0.2, 0.2, (K,)).astype(np.float32)
    combined, vps = [], []
    n_oob = 0
    for view in views:
        Xr, yr, grid, ib = G.get_view_from(image, lab, aff, view, D, span, bg_value=bg, center=center, scale=scale)
        Xs, ys, grid_h, ib_h = seq.get_view_from(vol, view, "same+20")
        np.testing.assert_array_equal(Xs.cpu().numpy(), Xr)               # 2.4 M trilinear samples x 2 channels, bit for bit
        np.testing.assert_array_equal(ys.cpu().numpy(), yr)
        np.testing.assert_array_equal(ib_h, ib)
        for a_h, a_r in zip(grid_h, grid):
            np.testing.assert_array_equal(np.asarray(a_h), a_r)
        z = Xr @ A + 0.02 * np.arange(K, dtype=np.float32)                # a per-pixel "prediction" both sides share
        e = np.exp(z - z.max(-1, keepdims=True))
        pred = (e / e.sum(-1, keepdims=True)).astype(np.float32)          # [d,d,P,K]
        ref = G.map_real_space_pred(pred, grid, ib, vg)
        got = map_real_space_pred(torch.tensor(pred, device="cuda"), grid, ib, vol).cpu().numpy()
        np.testing.assert_array_equal(got, ref)                           # 2.1 M voxels x 5 floats: the same vector everywhere
        n_oob += int((ref[..., 0] == 1.0).sum())
        combined.append(ref)
        vps.append((torch.tensor(np.moveaxis(pred, 2, 0).copy(), device="cuda"), grid, ib))
    assert n_oob > 0                                                      # (the OOB fill rule was exercised)
    merged_ref, map_ref = G.merge_multi_view_preds(np.stack(combined), W, b, False)
    probs, labels = map_and_fuse(vol, vps, W, b, sum_fusion=False)
    assert np.abs(probs.cpu().numpy() - merged_ref).max() <= 2e-6
    n_ties = assert_labels_equal_outside_float_ties(labels.cpu().numpy(), map_ref, merged_ref)
    assert n_ties <= 8, n_ties
    assert np.bincount(map_ref.ravel(), minlength=K).min() > 0.01 * D ** 3   # (all five classes are present)


def test_fusion_forward_vs_oracle():
    import ctypes as C
    from multiplanarunet_amd import _lib
    from oracle import geometry as G
    rng = np.random.RandomState(0)
    for (N, V, K) in ((1000, 6, 3), (777, 3, 5), (64, 1, 1), (4096, 6, 16)):
        x = rng.rand(N, V, K).astype(np.float32)
        x /= x.sum(-1, keepdims=True)
        W = rng.uniform(.5, 1.5, (V, K)).astype(np.float32)
        b = rng.uniform(-.2, .2, (K,)).astype(np.float32)
        ref = G.fusion_layer(x, W, b)
        xd, Wd, bd = (torch.tensor(t, device="cuda") for t in (x, W, b))
        probs = torch.empty((N, K), device="cuda")
        lab = torch.empty((N,), dtype=torch.uint8, device="cuda")
        _lib.call("mpu_fusion_forward", _lib.ptr(xd), N, V, K, _lib.ptr(Wd), _lib.ptr(bd),
                  _lib.ptr(probs), _lib.ptr(lab), _lib.stream_ptr())
        np.testing.assert_allclose(probs.cpu().numpy(), ref, rtol=0, atol=1e-6)
        assert assert_labels_equal_outside_float_ties(lab.cpu().numpy(), ref.argmax(-1), ref, band=4e-6) <= 2


def test_error_behaviour():
    import ctypes as C
    from multiplanarunet_amd import _lib
    x = torch.zeros(4, device="cuda")
    with pytest.raises(_lib.MpuError):
        _lib.call("mpu_fusion_forward", _lib.ptr(x), 1, 1, 17, _lib.ptr(x), _lib.ptr(x),
                  _lib.ptr(x), None, _lib.stream_ptr())
    with pytest.raises(_lib.MpuError):
        _lib.call("mpu_fusion_forward", None, 1, 1, 3, _lib.ptr(x), _lib.ptr(x),
                  _lib.ptr(x), None, _lib.stream_ptr())


def test_train_sampler_planes_match_view_sampling_and_fg_balance():
    """TrainSampler's fast path cuts the same planes as sample_view(ViewGeometry(view, noise), offset) and keeps the
    reference's foreground balancing (>= ceil(B * fg_batch_fraction) slices with foreground when available)."""
    from multiplanarunet_amd.data import make_toy_volume, as_volume, random_views, TrainSampler
    from multiplanarunet_amd.interpolation import ViewGeometry, sample_view
    vols = []
    for i in range(2):
        img, lab, aff = make_toy_volume(48, i)
        aff = aff.copy(); aff[:3, :3] *= 1.0 + 0.25 * i           # different voxel sizes
        vols.append(as_volume(img, lab, aff, "1pct", "RobustScaler", "cuda", "toy%d" % i))
    views = random_views(3, 60.0, 0)
    s = TrainSampler(vols, views, 48, 48.0, 8, 3, noise_sd=0.1, seed=5)
    # replay the sampler's random stream by hand for the first candidate
    rng = np.random.RandomState(5)
    vi = rng.randint(0, len(vols))
    view = views[rng.randint(0, len(views))]
    off = rng.uniform(-(48.0 // 2), 48.0 // 2)
    noise = rng.normal(scale=0.1, size=3)
    g = ViewGeometry(view, 48, 48.0, "same", noise=noise)
    g.offsets = np.array([off]); g.n_planes = 1
    Xr, yr = sample_view(vols[vi], g, want_labels=True)
    X = torch.empty((8, 48, 48, 1), device="cuda"); Y = torch.empty((8, 48, 48), dtype=torch.uint8, device="cuda")
    s2 = TrainSampler(vols, views, 48, 48.0, 8, 3, noise_sd=0.1, seed=5)
    assert s2.rng.randint(0, len(vols)) == vi
    m, nonbg = s2._cut(vi, X, Y, 0, torch.empty(1, dtype=torch.float64, device="cuda"),
                       torch.empty(2, dtype=torch.int32, device="cuda"))
    assert (Y[0] != yr[0]).float().mean().item() < 0.01             # basis differs by ~1e-7: a few edge pixels at most
    assert (X[0] - Xr[0]).abs().max().item() < 1e-3
    present = set(torch.unique(Y[0]).cpu().tolist())
    assert {c for c in range(32) if m >> c & 1} == present and nonbg == bool((X[0] != X[0, 0, 0]).any().item())
    # fg balancing over full batches
    for _ in range(5):
        x, y, w = s()
        assert x.shape == (8, 48, 48, 1) and y.shape == (8, 48 * 48, 1) and w.shape == (8,)
        has_fg = (y.reshape(8, -1) > 0).any(1).sum().item()
        assert has_fg >= 4


def _schedule_log(lib, fn):
    import ctypes as C
    lib.mpu_schedule_log_enable(1)
    try:
        out = fn()
        n = lib.mpu_schedule_log_read(None, 0)
        buf = C.create_string_buffer(int(n) + 1)
        lib.mpu_schedule_log_read(buf, n + 1)
    finally:
        lib.mpu_schedule_log_enable(0)
    return out, buf.value.decode().splitlines()


@pytest.mark.parametrize("aff", ("ident", "rot", "aniso", "aniso1", "odd"))
def test_fast_paths_equal_exact_search_at_scale(aff):
    """The closed-form fast paths (uniform axes, samples farther than 1e-6 index units from a decision boundary)
    against the exact NumPy-order search forced for every sample: sampled planes, labels, mapped volumes and fused
    labels must be IDENTICAL, on volumes large enough to hit node / tie / out-of-bounds neighbourhoods (integer
    spans put many samples exactly on nodes and half-way points)."""
    from multiplanarunet_amd import _lib
    from multiplanarunet_amd.interpolation import Volume, ViewGeometry, sample_view, map_and_fuse, map_real_space_pred, map_accumulate
    lib = _lib.load()
    rng = np.random.RandomState(3)
    D, K = 96, 3
    A = np.eye(4)
    if aff == "rot":
        th = np.deg2rad(25.0)
        R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
        A[:3, :3] = R.dot(np.diag([1.0, 0.8, 1.5])); A[:3, 3] = [3.0, -2.0, 5.0]
    if aff.startswith("aniso"):       # pixdims whose voxel axes are NOT reproduced by the linspace form: closed-form kind 2
        A[:3, :3] = np.diag([0.8, 0.7, 1.3]); A[:3, 3] = [1.0, 2.0, -3.0]
    nch = 1 if aff == "aniso1" else 2
    shape = (D, D - 8, D + 4)
    if aff == "odd":                  # odd extents: partial bricks, Z % 4 != 0 (unpacked label stores, scalar accumulate), 5 classes
        A[:3, :3] = np.diag([1.7, 1.9, 1.1]); shape = (50, 47, 61); K = 5
    vol_np = rng.randn(*shape, nch).astype(np.float32)
    lab_np = rng.randint(0, K, shape).astype(np.uint8)
    views = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0], [0.3, 0.5, 0.8], [-0.6, 0.64, 0.48]], float)
    W = torch.tensor(rng.uniform(.5, 1.5, (len(views), K)).astype(np.float32))
    b = torch.tensor(rng.uniform(-.1, .1, (K,)).astype(np.float32))
    outs = {}
    for fast in (1, 0):
        _lib.check(lib.mpu_geometry_set_fast_path(fast), "mpu_geometry_set_fast_path")
        try:
            vol = Volume(vol_np, lab_np, A, bg_value=[0.5, -1.0][:nch], scaler=(np.array([0.1, 0.2][:nch]), np.array([1.3, 0.7][:nch])))

            def run():
                res, preds = [], []
                for vi, v in enumerate(views):
                    for dim, span in ((D, float(D)), (64, 70.0)):        # integer step (samples on nodes) and a ragged one
                        g = ViewGeometry(v, dim, span, "same+20")
                        X, y = sample_view(vol, g, want_labels=(vi % 2 == 0))
                        res += [X.clone()] + ([y.clone()] if y is not None else [])
                        if dim == D:
                            pr = torch.tensor(np.random.RandomState(10 + vi).rand(g.n_planes, dim, dim, K).astype(np.float32), device="cuda")
                            preds.append((pr, (g.real_axis, g.real_axis, g.offsets), g.inv_basis))
                probs, labels = map_and_fuse(vol, preds, W, b)
                # single-view kernels: map_real_space_pred of one view, and the sharded accumulate in two plane chunks
                pr, grid, ib = preds[3]
                mapped = map_real_space_pred(pr.permute(1, 2, 0, 3), grid, ib, vol)
                z = torch.zeros(tuple(vol.image.shape[:3]) + (K,), dtype=torch.float32, device="cuda")
                P = pr.shape[0]
                for lo, hi in ((0, P // 3), (P // 3, P)):
                    map_accumulate(vol, pr[lo:hi].contiguous(), grid, ib, W[3].cuda(), lo, hi, owns_oob=(lo == 0), z=z)
                return res + [probs.clone(), labels.clone(), mapped.clone(), z.clone()]

            outs[fast], log = _schedule_log(lib, run)
            if fast:                      # the straight-line kernels are the ones under test
                kinds = {"ident": "kind=1", "aniso": "kind=2", "aniso1": "kind=2"}
                if aff in kinds:
                    assert sum(1 for l in log if l.startswith("sample fast " + kinds[aff])) == 2 * len(views), log
                if aff == "odd":
                    assert any(l.startswith("sample fast") for l in log), log
                assert any(l.startswith("map_fuse fast") for l in log), log
                assert sum(1 for l in log if l.startswith("map_view fast")) == 3, log
            else:
                assert not any(" fast" in l for l in log), log
        finally:
            _lib.check(lib.mpu_geometry_set_fast_path(1), "mpu_geometry_set_fast_path")
    for a, b_ in zip(outs[1], outs[0]):
        assert torch.equal(a, b_)


def test_cell_division_is_ieee():
    """The straight-line sampler divides by a cell width through a Newton-refined reciprocal of the axis step
    (csrc/geometry.hip cell_div); it must give the IEEE quotient the reference's NumPy division gives. 2^28
    pseudo-random quotients per axis, on linspace axes (kind 1) and voxel axes (kind 2) of several spacings."""
    import ctypes as C
    from multiplanarunet_amd import _lib
    lib = _lib.load()
    axes = [np.linspace(-128, 128, 256), np.linspace(-35.0, 35.0, 64), np.linspace(-140.04, 140.04, 276)]
    for n, pd in ((256, 1.0), (96, 0.8), (88, 0.7), (100, 1.3), (512, 0.123456789), (600, 2.9999)):
        a32 = np.arange(n, dtype=np.float32) - np.float32((n - 1) / 2)
        axes.append(a32.astype(np.float64) * np.float64(pd))
    kinds = set()
    for i, ax in enumerate(axes):
        m = _lib.make_axis(ax)
        assert m.kind in (1, 2)
        kinds.add(m.kind)
        bad = C.c_uint64(123)
        _lib.check(lib.mpu_geometry_check_cell_division(C.byref(m), 1 << 28, 1000 + i, C.byref(bad)),
                   "mpu_geometry_check_cell_division")
        assert bad.value == 0, (i, m.kind, m.step, bad.value)
    assert kinds == {1, 2}


@pytest.mark.parametrize("an", ("ident", "rot"))
def test_per_view_evaluation_equals_the_oracle_exactly(golden, an):
    """_per_view_evaluation (mpunet/bin/predict.py:248-275, inside the loop at :334-346). Per view the Dice of the
    argmax of the view-space prediction against the SAMPLED labels and of the back-mapped prediction against the volume's labels
    (dice_all, ignore_zero=False) and their mean without class 0 -- on the GPU through mpu_map_view_nearest + mpu_validation_count,
    integer counts -> the oracle's float32 values exactly."""
    from multiplanarunet_amd.interpolation import Volume, ViewSampler
    from multiplanarunet_amd.predict import multi_view_predict
    from oracle import geometry as G
    D, K = 48, 4
    rng = np.random.RandomState(48)
    g = np.indices((D, D, D)).astype(np.float32) / D
    image = (60 * np.sin(5 * g[0]) * np.cos(3 * g[1]) + 40 * g[2] + 4 * rng.randn(D, D, D)).astype(np.float32)[..., None]
    lab = ((g[0] > .3).astype(np.uint8) + (g[1] > .5) + (g[2] > .6)).astype(np.uint8)
    aff = golden["aff_" + an]
    bg = [float(np.percentile(image[..., 0], 1))]
    center, scale = Volume.fit_robust_scaler(image)
    vol = Volume(image, lab, aff, bg_value=bg, scaler=(center, scale))
    views = np.array([[1, 0, 0], [0.3, 0.5, 0.8], [-0.4, 0.6, 0.55]], float)
    views /= np.linalg.norm(views, axis=1, keepdims=True)
    span = float(D) * 1.05
    A = (rng.randn(1, K) * 2.5).astype(np.float32)
    vg = G.voxel_grid_real_space(image.shape[:3], aff)

    def probs_of(X):                                                      # a per-pixel "U-Net" both sides share (f32 NumPy)
        z = X @ A + 0.3 * np.arange(K, dtype=np.float32)
        e = np.exp(z - z.max(-1, keepdims=True))
        return (e / e.sum(-1, keepdims=True)).astype(np.float32)

    class Net:                                                            # model.predict on [P,dim,dim,C] device planes
        def predict(self, X, batch_size=None):
            return torch.tensor(probs_of(X.cpu().numpy()), device=X.device)

    want = []
    for view in views:
        Xr, yr, grid, ib = G.get_view_from(image, lab, aff, view, D, span, bg_value=bg, center=center, scale=scale)
        pred = probs_of(Xr)                                               # [d,d,P,K]
        mapped = G.map_real_space_pred(pred, grid, ib, vg)
        vd = G.dice_all(yr, pred.argmax(-1), n_classes=K, ignore_zero=False)
        md = G.dice_all(lab, mapped.argmax(-1), n_classes=K, ignore_zero=False)
        want.append((vd, md, md[~np.isnan(md)][1:].mean()))
    got = {}
    np.random.seed(0)
    multi_view_predict(Net(), vol, views, D, span, sum_fusion=True, want_probs=False,
                       per_view_eval=dict(eval_prob=1.0, n_classes=K, report=lambda i, view, vd, md, mean: got.__setitem__(i, (vd, md, mean))))
    assert sorted(got) == [0, 1, 2]
    for i, (vd, md, mean) in enumerate(want):
        np.testing.assert_array_equal(got[i][0], vd)
        np.testing.assert_array_equal(got[i][1], md)
        assert got[i][2] == mean and 0.0 < mean < 1.0
    # eval_prob: the same draw as the reference (np.random.rand() > eval_prob skips the view)
    np.random.seed(3)
    draws = np.random.rand(3)
    seen = []
    np.random.seed(3)
    multi_view_predict(Net(), vol, views, D, span, sum_fusion=True, want_probs=False,
                       per_view_eval=dict(eval_prob=0.5, n_classes=K, report=lambda i, *a: seen.append(i)))
    assert seen == [i for i in range(3) if not draws[i] > 0.5]
    # no labels: nothing to evaluate, the prediction itself is unchanged
    vol2 = Volume(image, None, aff, bg_value=bg, scaler=(center, scale))
    l0 = multi_view_predict(Net(), vol, views, D, span, sum_fusion=True, want_probs=False)[1]
    l1 = multi_view_predict(Net(), vol2, views, D, span, sum_fusion=True, want_probs=False,
                            per_view_eval=dict(eval_prob=1.0, n_classes=K, report=lambda *a: seen.append("x")))[1]
    assert torch.equal(l0, l1) and "x" not in seen
