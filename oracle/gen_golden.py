"""
TEST INFRASTRUCTURE ONLY -- generates tests/golden/geometry_golden.npz by
running the reference's own, unmodified NumPy functions (mpunet 0.2.12 under
/root/reference) through oracle/ref_shim.py. Run by hand in the build
container (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

The file holds only data: seeded inputs and the reference's outputs
(SURVEY.md section 8c, G1..G7).
"""
import io
import os
import sys
import contextlib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()

from mpunet.interpolation.view_interpolator import ViewInterpolator  # noqa
from mpunet.sequences import IsotrophicLiveViewSequence2D  # noqa
from mpunet.utils.fusion.fuse_and_predict import map_real_space_pred  # noqa
from mpunet.interpolation.sample_grid import (sample_plane_at,  # noqa
                                              get_voxel_grid_real_space)
from mpunet.evaluate.metrics import dice_all  # noqa
from mpunet.utils.utils import pred_to_class  # noqa
from mpunet.preprocessing.scaling import get_scaler  # noqa
from mpunet.interpolation.linalg import get_rotation_matrix  # noqa


class Img:
    """Duck-typed ImagePair (fields of SURVEY.md section 8b)."""
    def __init__(self, image, labels, affine, bg_value=0.0, scaler=None):
        self.image = image
        self.labels = labels
        self.affine = affine
        self.shape = np.array(image.shape)
        self.n_channels = image.shape[-1]
        self.predict_mode = labels is None
        self.interpolator = ViewInterpolator(image, labels, affine=affine,
                                             bg_value=bg_value)
        self.scaler = scaler
        self.identifier = "golden"


class IdentityScaler:
    def transform(self, x):
        return x


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def affines():
    ident = np.eye(4)
    aniso = np.diag([1.0, 0.5, 2.0, 1.0])
    R = get_rotation_matrix(np.array([.2, .3, 1.]), 25)
    rot = np.eye(4)
    rot[:3, :3] = R.dot(np.diag([1.0, 0.8, 1.5]))
    rot[:3, 3] = [3.0, -2.0, 7.0]
    return {"ident": ident, "aniso": aniso, "rot": rot}


def blob_labels(shape, rng):
    X, Y, Z = shape
    gi, gj, gk = np.meshgrid(np.arange(X), np.arange(Y), np.arange(Z),
                             indexing="ij")
    lab = np.zeros(shape, np.uint8)
    c = np.array(shape) / 2.0
    r = np.sqrt(((gi - c[0]) / (X * .30)) ** 2 + ((gj - c[1]) / (Y * .25)) ** 2 +
                ((gk - c[2]) / (Z * .35)) ** 2)
    lab[r < 1.0] = 1
    lab[(abs(gi - X * .35) < X * .12) & (abs(gj - Y * .6) < Y * .15) &
        (abs(gk - Z * .5) < Z * .2)] = 2
    return lab


def main():
    out = {}
    rng = np.random.RandomState(0)
    views = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0],
                      [0.1, 0.15, 0.98],           # the "<0.2" branch
                      [-0.1, 0.05, 0.99],          # "<0.2" with abs()
                      [0.3, 0.5, 0.8],
                      [-0.6, 0.64, 0.48],
                      [0.7, -0.5, 0.5]], dtype=np.float64)
    rv = rng.normal(size=(3, 3))
    rv /= np.linalg.norm(rv, axis=1, keepdims=True)
    rv[:, -1] = np.abs(rv[:, -1])
    views = np.concatenate([views, rv], 0)
    out["views"] = views

    # ---- G1: sample_plane_at ------------------------------------------------
    g1_cfg = [(16, 31.0, -3.25), (16, 40.0, 0.0), (32, 33.0, 7.5),
              (32, 32.0, -12.0)]
    out["g1_cfg"] = np.array(g1_cfg)
    for vi, v in enumerate(views):
        for ci, (dim, span, off) in enumerate(g1_cfg):
            rg, g, ib = sample_plane_at(v, int(dim), span, off, 0., test_mode=True)
            out["g1_grid_%d_%d" % (vi, ci)] = rg
            out["g1_g_%d_%d" % (vi, ci)] = g
            out["g1_invb_%d_%d" % (vi, ci)] = ib

    # ---- G2: ViewInterpolator ----------------------------------------------
    vol = rng.randn(24, 20, 16, 2).astype(np.float32)
    lab = rng.randint(0, 4, size=(24, 20, 16)).astype(np.uint8)
    out["g2_vol"] = vol
    out["g2_lab"] = lab
    bg = [-1.5, 0.25]
    out["g2_bg"] = np.array(bg)
    g2_planes = [(0, 16, 31.0, -3.25), (1, 16, 40.0, 2.0), (3, 32, 33.0, 1.5),
                 (5, 32, 33.0, -4.0), (6, 16, 31.0, 6.0), (8, 32, 40.0, 0.5)]
    out["g2_planes"] = np.array(g2_planes)
    for an, aff in affines().items():
        out["aff_" + an] = aff
        vi_ = ViewInterpolator(vol, lab, affine=aff, bg_value=bg)
        for pi, (v, dim, span, off) in enumerate(g2_planes):
            rg, _, _ = sample_plane_at(views[int(v)], int(dim), span, off, 0.,
                                       test_mode=True)
            im, lb = vi_(rg)
            out["g2_im_%s_%d" % (an, pi)] = im
            out["g2_lab_%s_%d" % (an, pi)] = lb

    # ---- G3/G4/G5/G7: get_view_from, voxel grid, map, round trip -----------
    D = (32, 28, 24)
    vol3 = (rng.randn(*D, 1) * 50 + 100).astype(np.float32)
    lab3 = blob_labels(D, rng)
    out["g3_vol"] = vol3
    out["g3_lab"] = lab3
    scaler = get_scaler("RobustScaler").fit(vol3)
    out["g3_center"] = np.array([s.center_[0] for s in scaler.scalers])
    out["g3_scale"] = np.array([s.scale_[0] for s in scaler.scalers])
    g3_views = [0, 1, 5, 6]
    out["g3_views"] = np.array(g3_views)
    for an, aff in affines().items():
        img = Img(vol3, lab3, aff, bg_value=[12.5], scaler=scaler)
        for dim, span in ((16, 30.0), (32, 33.0)):
            seq = IsotrophicLiveViewSequence2D(
                None, views=views, dim=dim, batch_size=4, n_classes=3,
                real_space_span=span, no_log=True, logger=lambda *a, **k: None)
            vg = get_voxel_grid_real_space(img)
            if dim == 16:
                out["g4_vgrid_%s" % an] = vg
            for v in g3_views:
                Xs, ys, grid, ib = quiet(seq.get_view_from, img, views[v],
                                         "same+20")
                key = "%s_%d_%d" % (an, dim, v)
                out["g3_X_" + key] = Xs
                out["g3_y_" + key] = ys
                out["g3_g_" + key] = grid[0]
                out["g3_off_" + key] = grid[2]
                out["g3_invb_" + key] = ib
                # G7: model-free round trip: one-hot(ys) -> map -> argmax
                onehot = np.eye(3, dtype=np.float32)[ys]
                mapped = quiet(map_real_space_pred, onehot, grid, ib, vg)
                out["g7_map_" + key] = mapped.argmax(-1).astype(np.uint8)
                if dim == 16:
                    # G5: random predictions, K in {1,3,5}
                    for K in (1, 3, 5):
                        pr = np.random.RandomState(K * 7 + v).rand(
                            dim, dim, dim + 20, K).astype(np.float32)
                        mp = quiet(map_real_space_pred, pr, grid, ib, vg)
                        out["g5_pred_%s_%d_%d" % (an, v, K)] = pr
                        out["g5_map_%s_%d_%d" % (an, v, K)] = mp

    # ---- G6: dice_all / pred_to_class ---------------------------------------
    a = rng.randint(0, 4, size=(12, 10, 8)).astype(np.uint8)
    b = rng.randint(0, 4, size=(12, 10, 8)).astype(np.uint8)
    b[a == 3] = 0                                   # class 3 absent from pred
    out["g6_a"], out["g6_b"] = a, b
    out["g6_dice_5"] = dice_all(a, b, n_classes=5, ignore_zero=True)
    out["g6_dice_4_with0"] = dice_all(a, b, n_classes=4, ignore_zero=False)
    p = rng.rand(6, 5, 4, 3).astype(np.float32)
    out["g6_probs"] = p
    out["g6_cls"] = pred_to_class(p, img_dims=3)

    dst = os.path.join(HERE, "..", "tests", "golden", "geometry_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", os.path.abspath(dst), "%.2f MB" % (os.path.getsize(dst) / 1e6),
          len(out), "arrays")


if __name__ == "__main__":
    main()
