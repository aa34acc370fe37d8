"""
Pins the oracle (oracle/geometry.py, a NumPy restatement) against golden
vectors produced by the reference's own unmodified NumPy code
(oracle/gen_golden.py -> tests/golden/geometry_golden.npz).
Every comparison is EXACT (round 6): the restatement reproduces the reference's
arrays bit for bit, and tests/test_gpu_geometry.py leans on exactly that.
CPU only.
"""
import numpy as np
import pytest
from oracle import geometry as G

AFFS = ("ident", "aniso", "rot")


def test_g1_sample_plane_at(golden):
    views = golden["views"]
    for vi, v in enumerate(views):
        for ci, (dim, span, off) in enumerate(golden["g1_cfg"]):
            rg, g, ib = G.sample_plane_at(v, int(dim), span, off)
            np.testing.assert_array_equal(rg, golden["g1_grid_%d_%d" % (vi, ci)])
            np.testing.assert_array_equal(g, golden["g1_g_%d_%d" % (vi, ci)])
            np.testing.assert_array_equal(ib, golden["g1_invb_%d_%d" % (vi, ci)])


@pytest.mark.parametrize("an", AFFS)
def test_g2_view_interpolator(golden, an):
    vol, lab, bg = golden["g2_vol"], golden["g2_lab"], list(golden["g2_bg"])
    for pi, (v, dim, span, off) in enumerate(golden["g2_planes"]):
        rg, _, _ = G.sample_plane_at(golden["views"][int(v)], int(dim), span, off)
        im, lb = G.view_interpolate(vol, lab, golden["aff_" + an], bg, 0, rg)
        ref_im = golden["g2_im_%s_%d" % (an, pi)]
        ref_lb = golden["g2_lab_%s_%d" % (an, pi)]
        assert im.dtype == ref_im.dtype and lb.dtype == ref_lb.dtype
        np.testing.assert_array_equal(im, ref_im)          # bit for bit (round 6: the allowances of rounds 1-5 were never needed)
        np.testing.assert_array_equal(lb, ref_lb)


@pytest.mark.parametrize("an", AFFS)
@pytest.mark.parametrize("dim", (16, 32))
def test_g3_get_view_from(golden, an, dim):
    span = {16: 30.0, 32: 33.0}[dim]
    for v in golden["g3_views"]:
        key = "%s_%d_%d" % (an, dim, v)
        Xs, ys, grid, ib = G.get_view_from(
            golden["g3_vol"], golden["g3_lab"], golden["aff_" + an],
            golden["views"][v], dim, span, bg_value=[12.5],
            center=golden["g3_center"], scale=golden["g3_scale"])
        np.testing.assert_array_equal(Xs, golden["g3_X_" + key])
        np.testing.assert_array_equal(ys, golden["g3_y_" + key])
        np.testing.assert_array_equal(grid[0], golden["g3_g_" + key])
        np.testing.assert_array_equal(grid[2], golden["g3_off_" + key])
        np.testing.assert_array_equal(ib, golden["g3_invb_" + key])


@pytest.mark.parametrize("an", AFFS)
def test_g4_voxel_grid(golden, an):
    vg = G.voxel_grid_real_space(golden["g3_vol"].shape[:3], golden["aff_" + an])
    np.testing.assert_array_equal(vg, golden["g4_vgrid_" + an])


@pytest.mark.parametrize("an", AFFS)
def test_g5_map_real_space_pred(golden, an):
    vg = golden["g4_vgrid_" + an]
    for v in golden["g3_views"]:
        key = "%s_16_%d" % (an, v)
        grid = (golden["g3_g_" + key], golden["g3_g_" + key],
                golden["g3_off_" + key])
        for K in (1, 3, 5):
            pr = golden["g5_pred_%s_%d_%d" % (an, v, K)]
            ref = golden["g5_map_%s_%d_%d" % (an, v, K)]
            mp = G.map_real_space_pred(pr, grid, golden["g3_invb_" + key], vg)
            assert mp.dtype == ref.dtype and mp.shape == ref.shape
            np.testing.assert_array_equal(mp, ref)          # every voxel's vector is the reference's


def test_g6_dice_and_class(golden):
    a, b = golden["g6_a"], golden["g6_b"]
    np.testing.assert_array_equal(G.dice_all(a, b, n_classes=5), golden["g6_dice_5"])
    np.testing.assert_array_equal(G.dice_all(a, b, n_classes=4, ignore_zero=False),
                                  golden["g6_dice_4_with0"])
    np.testing.assert_array_equal(G.pred_to_class(golden["g6_probs"]),
                                  golden["g6_cls"])


@pytest.mark.parametrize("an", AFFS)
def test_g7_round_trip(golden, an):
    """one_hot(ys) -> map -> argmax equals the reference's mapped label volume."""
    vg = golden["g4_vgrid_" + an]
    lab = golden["g3_lab"]
    for dim in (16, 32):
        for v in golden["g3_views"]:
            key = "%s_%d_%d" % (an, dim, v)
            ys = golden["g3_y_" + key]
            onehot = np.eye(3, dtype=np.float32)[ys]
            grid = (golden["g3_g_" + key], golden["g3_g_" + key],
                    golden["g3_off_" + key])
            mp = G.map_real_space_pred(onehot, grid, golden["g3_invb_" + key], vg)
            got = mp.argmax(-1).astype(np.uint8)
            ref = golden["g7_map_" + key]
            np.testing.assert_array_equal(got, ref)
            if dim == 32 and an == "ident" and v in (0, 1):
                d = G.dice_all(lab, got, n_classes=3)
                assert np.all(d > 0.9), d
