"""Host-side view geometry (multiplanarunet_amd.interpolation) against the reference goldens. CPU only."""
import re
import os
import ctypes
import numpy as np
import pytest
from multiplanarunet_amd import interpolation as I
from multiplanarunet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plane_basis_and_axes_match_reference(golden):
    for vi, v in enumerate(golden["views"]):
        for ci, (dim, span, off) in enumerate(golden["g1_cfg"]):
            g = I.ViewGeometry(v, int(dim), span, "same+20")
            np.testing.assert_array_equal(g.inv_basis, golden["g1_invb_%d_%d" % (vi, ci)])
            np.testing.assert_array_equal(g.real_axis, golden["g1_g_%d_%d" % (vi, ci)])
            # in-plane coordinates: basis @ (i*step+start, j*step+start, off)
            ax = np.arange(int(dim), dtype=np.float64) * g.g_step + g.g_start
            gx, gy = np.meshgrid(ax, ax, indexing="ij")
            pts = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, off)], 1)
            real = g.basis.dot(pts.T).T
            ref = golden["g1_grid_%d_%d" % (vi, ci)]
            for k in range(3):
                np.testing.assert_array_equal(real[:, k].reshape(ref[k].shape), ref[k])


def test_offsets_match_reference(golden):
    for an in ("ident", "rot"):
        for dim, span in ((16, 30.0), (32, 33.0)):
            for v in golden["g3_views"]:
                key = "%s_%d_%d" % (an, dim, v)
                g = I.ViewGeometry(golden["views"][v], dim, span, "same+20")
                np.testing.assert_array_equal(g.offsets, golden["g3_off_" + key])
                np.testing.assert_array_equal(g.real_axis, golden["g3_g_" + key])
                np.testing.assert_array_equal(g.inv_basis, golden["g3_invb_" + key])
                assert g.n_planes == dim + 20


def test_volume_axes_and_voxel_grid(golden):
    from oracle import geometry as G
    for an in ("ident", "aniso", "rot"):
        aff = golden["aff_" + an]
        vol = I.Volume(golden["g3_vol"], golden["g3_lab"], aff, bg_value=[12.5], device="cpu")
        axes, rot = G.voxel_axes_real_space(golden["g3_vol"].shape[:3], aff)
        for a, b in zip(vol.axes, axes):
            np.testing.assert_array_equal(a, b)
        assert (rot is None) == (vol.rot_mat is None)
        if rot is not None:
            np.testing.assert_array_equal(rot, vol.rot_mat)
        vg = vol.voxel_grid()
        A = np.array(vg.A[:]).reshape(3, 3)
        X, Y, Z = vg.shape[:]
        ref = golden["g4_vgrid_" + an]
        for (i, j, k) in ((0, 0, 0), (X - 1, 2, 5), (3, Y - 1, Z - 1)):
            p = A.dot(np.array([i, j, k], float)) - np.array(vg.center[:])
            np.testing.assert_allclose(p, ref[:, i, j, k], rtol=0, atol=1e-11)


def test_volume_argument_errors():
    with pytest.raises(ValueError):
        I.Volume(np.zeros((4, 4, 4), np.float32), device="cpu")
    with pytest.raises(ValueError):
        I.Volume(np.zeros((4, 4, 4, 2), np.float32), bg_value=[0, 1, 2], device="cpu")


def test_capi_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mpunet_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(mpu_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == _lib.declared_symbols()
    lib = _lib.load()                      # raises if the .so or a symbol is missing
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.mpu_abi_version() >= 1
    assert ctypes.sizeof(_lib.Axis) == 32
    assert ctypes.sizeof(_lib.ViewGeom) == 9 * 8 * 2 + 4 * 4 + 16 + 3 * 32
    assert ctypes.sizeof(_lib.ViewPred) == 72 + 24 + 8 + 2 * 32
    assert ctypes.sizeof(_lib.VoxelGrid) == 72 + 24 + 16


def test_library_carries_the_hash_of_its_sources_and_a_stale_one_is_refused(monkeypatch):
    """build.py compiles sha256(sources, headers, flags) into the library; _lib.load() refuses a library built from other
    sources (round 4: build() compared mtimes only and could ship a stale binary)."""
    from multiplanarunet_amd import build as B
    want = B.expected_hash()
    assert len(want) == 16 and _lib.build_hash() == want

    class Stale:
        @staticmethod
        def mpu_build_hash():
            return b"0123456789abcdef"
    monkeypatch.delenv("MPU_LIB_PATH", raising=False)
    with pytest.raises(_lib.MpuError, match="built from other sources"):
        _lib._check_build_hash(Stale)
    # the hash moves with a flag and with a header byte
    from multiplanarunet_amd import srchash
    assert srchash.build_sha16(lambda s: B.flags_of(s) + ["-DX"]) != want
    assert srchash.unit_sha16("env.hip", B.flags_of("env.hip"), b"other headers") != srchash.unit_sha16("env.hip", B.flags_of("env.hip"))


def test_axis_closed_forms_reproduce_the_arrays_bitwise(golden):
    """make_axis only claims a closed form when it reproduces the reference's axis array exactly."""
    def rebuild(a):
        i = np.arange(a.n, dtype=np.float64)
        if a.kind == 1:
            v = i * a.step + a.start
            v[-1] = a.last
            return v
        if a.kind == 2:
            return (i - a.start) * a.step
        return None
    n_closed = 0
    for an in ("ident", "aniso", "rot"):
        vol = I.Volume(golden["g3_vol"], None, golden["aff_" + an], bg_value=[0.0], device="cpu")
        for ax in vol.axes:
            a = _lib.make_axis(ax)
            assert a.kind in (1, 2)
            np.testing.assert_array_equal(rebuild(a), ax)
            n_closed += 1
    for dim, span in ((16, 30.0), (32, 33.0), (256, 256.0), (512, 512.0), (128, 101.3)):
        g = I.ViewGeometry([0.3, 0.5, 0.8], dim, span, "same+20")
        for arr in (g.real_axis, g.offsets):
            a = _lib.make_axis(arr)
            if a.kind:
                np.testing.assert_array_equal(rebuild(a), arr)
                n_closed += 1
    assert n_closed >= 15
    odd = np.array([0.0, 1.0, 2.5, 7.0])                 # non-uniform axis: no closed form claimed
    assert _lib.make_axis(odd).kind == 0


def test_product_dice_all_and_pred_to_class_match_reference_goldens(golden):
    """G6 on the PRODUCT functions (reference argument order: y_true, y_pred, smooth, n_classes, ...;
    mpunet/evaluate/metrics.py:26-52, mpunet/utils/utils.py:311-328)."""
    import inspect
    import torch
    a, b = golden["g6_a"], golden["g6_b"]
    assert list(inspect.signature(I.dice_all).parameters)[:6] == \
        ["y_true", "y_pred", "smooth", "n_classes", "ignore_zero", "skip_if_no_y"]
    np.testing.assert_array_equal(I.dice_all(a, b, 1.0, 5), golden["g6_dice_5"])
    np.testing.assert_array_equal(I.dice_all(a, b, n_classes=5, ignore_zero=True), golden["g6_dice_5"])
    np.testing.assert_array_equal(I.dice_all(a, b, n_classes=4, ignore_zero=False), golden["g6_dice_4_with0"])
    np.testing.assert_array_equal(I.dice_all(torch.from_numpy(a), torch.from_numpy(b), n_classes=5), golden["g6_dice_5"])
    # n_classes=None: the classes present in y_true; skip_if_no_y leaves NaN for classes absent from y_true
    d = I.dice_all(a, b, ignore_zero=False)
    assert d.shape == np.unique(a).shape and d.dtype == np.float32
    only_pred = np.where(b == b.max(), 7, b)
    assert np.isnan(I.dice_all(a, only_pred, n_classes=8, skip_if_no_y=True)[-1])
    assert not np.isnan(I.dice_all(a, only_pred, n_classes=8)[-1])
    p = golden["g6_probs"]
    got = I.pred_to_class(p, img_dims=3)
    assert got.dtype == np.uint8
    np.testing.assert_array_equal(got, golden["g6_cls"])
    np.testing.assert_array_equal(I.pred_to_class(torch.from_numpy(p)).numpy(), golden["g6_cls"])
    ints = golden["g6_cls"]
    assert I.pred_to_class(ints) is ints                                   # integer maps pass through
    np.testing.assert_array_equal(I.pred_to_class(p[..., :1]), p[..., :1] >= 0.5)


def test_wgrad_scratch_region_covers_the_grouped_and_the_standalone_plan():
    """a layer's weight-gradient scratch region was sized with the stand-alone plan and used with the
    grouped one (other thresholds: taps <-> glds, other split counts), which overflowed into the next layer's region on
    shapes outside the BASELINE configs. Host arithmetic only: sweep shapes and require region >= both plans."""
    lib = _lib.load()
    BF16, CONV3, UPCONV2 = 1, 0, 1
    # the four overflowing shapes the advisor measured on the round-3 library
    for B, H, Cin, Cout in [(2, 40, 128, 128), (1, 80, 128, 128), (4, 80, 64, 64), (1, 32, 512, 512)]:
        g = lib.mpu_conv2d_wgrad_job_floats(BF16, CONV3, B, H, H, Cin, 0, Cout, 1)
        s = lib.mpu_conv2d_wgrad_job_floats(BF16, CONV3, B, H, H, Cin, 0, Cout, 0)
        assert lib.mpu_conv2d_wgrad_scratch_floats(BF16, CONV3, B, H, H, Cin, 0, Cout) >= max(g, s)
    n = 0
    for dtype in (0, 1):
        for mode in (CONV3, UPCONV2):
            for B in (1, 2, 3, 4, 8, 16, 32):
                for H in (8, 16, 24, 32, 40, 48, 64, 80, 96, 128, 256):
                    for W in {H, 32, 2 * H}:
                        for Cin, Cout in [(8, 64), (64, 64), (64, 128), (96, 96), (128, 64), (128, 128), (184, 184), (256, 128),
                                          (256, 256), (368, 368), (512, 256), (512, 512), (1024, 512), (728, 728)]:
                            for C1 in (0, Cin // 2):
                                if C1 and (mode != CONV3 or (Cin // 2) % 8):
                                    continue
                                C0 = Cin - C1
                                need = lib.mpu_conv2d_wgrad_scratch_floats(dtype, mode, B, H, W, C0, C1, Cout)
                                for grouped in (0, 1):
                                    job = lib.mpu_conv2d_wgrad_job_floats(dtype, mode, B, H, W, C0, C1, Cout, grouped)
                                    assert 0 < job <= need, (dtype, mode, B, H, W, C0, C1, Cout, grouped, job, need)
                                n += 1
    assert n > 10000
