"""On-disk formats: the Keras name list of Appendix B, checkpoint selection, and the optional
h5py adapter (its round trip runs only where h5py exists); NIfTI is native -- tests/test_nifti_host.py."""
import os
import numpy as np
import pytest
from multiplanarunet_amd import formats as F

# SURVEY.md Appendix B, complexity_factor=1, n_channels=1, n_classes=3: layer -> kernel shape / vector length
APPENDIX_B = [
    ("encoder_L0_conv1", (3, 3, 1, 64)), ("encoder_L0_conv2", (3, 3, 64, 64)), ("encoder_L0_BN", 64),
    ("encoder_L1_conv1", (3, 3, 64, 128)), ("encoder_L1_conv2", (3, 3, 128, 128)), ("encoder_L1_BN", 128),
    ("encoder_L2_conv1", (3, 3, 128, 256)), ("encoder_L2_conv2", (3, 3, 256, 256)), ("encoder_L2_BN", 256),
    ("encoder_L3_conv1", (3, 3, 256, 512)), ("encoder_L3_conv2", (3, 3, 512, 512)), ("encoder_L3_BN", 512),
    ("bottom_conv1", (3, 3, 512, 1024)), ("bottom_conv2", (3, 3, 1024, 1024)), ("bottom_BN", 1024),
    ("upsample_L0_conv1", (2, 2, 1024, 512)), ("upsample_L0_BN1", 512), ("upsample_L0_conv2", (3, 3, 1024, 512)),
    ("upsample_L0_conv3", (3, 3, 512, 512)), ("upsample_L0_BN2", 512),
    ("upsample_L1_conv1", (2, 2, 512, 256)), ("upsample_L1_BN1", 256), ("upsample_L1_conv2", (3, 3, 512, 256)),
    ("upsample_L1_conv3", (3, 3, 256, 256)), ("upsample_L1_BN2", 256),
    ("upsample_L2_conv1", (2, 2, 256, 128)), ("upsample_L2_BN1", 128), ("upsample_L2_conv2", (3, 3, 256, 128)),
    ("upsample_L2_conv3", (3, 3, 128, 128)), ("upsample_L2_BN2", 128),
    ("upsample_L3_conv1", (2, 2, 128, 64)), ("upsample_L3_BN1", 64), ("upsample_L3_conv2", (3, 3, 128, 64)),
    ("upsample_L3_conv3", (3, 3, 64, 64)), ("upsample_L3_BN2", 64),
    ("conv2d", (1, 1, 64, 3)),
]


def _model():
    from multiplanarunet_amd.unet import UNet
    return UNet(n_classes=3, dim=128, depth=4, complexity_factor=1, device="cpu", logger=lambda *a, **k: None)


def test_keras_name_list_and_shapes_equal_survey_appendix_b():
    m = _model()
    assert F.keras_layer_names(4) == [n for n, _ in APPENDIX_B]
    d = m.get_weights_dict()
    expected = []
    for layer, shp in APPENDIX_B:
        if isinstance(shp, tuple):
            assert d[layer + "/kernel"].shape == shp and d[layer + "/bias"].shape == (shp[3],)
            expected += [layer + "/kernel", layer + "/bias"]
        else:
            for v in ("gamma", "beta", "moving_mean", "moving_variance"):
                assert d[layer + "/" + v].shape == (shp,)
                expected.append(layer + "/" + v)
    assert m._keras_order() == expected == [n for l in F.keras_layer_names(4) for n in F.layer_weight_names(l)]
    assert sum(int(np.prod(v.shape)) for v in d.values()) == m.count_params() == 31046339
    # the HDF5 layout a tf.keras 2.3 load_weights(by_name=True) walks: layer group -> "<layer>/<var>:0" datasets
    ent = F.h5_entries(d, 4)
    assert [l for l, _ in ent] == F.keras_layer_names(4)
    assert [n for n, _ in ent[2][1]] == ["encoder_L0_BN/gamma:0", "encoder_L0_BN/beta:0", "encoder_L0_BN/moving_mean:0",
                                          "encoder_L0_BN/moving_variance:0"]
    assert [n for n, _ in ent[-1][1]] == ["conv2d/kernel:0", "conv2d/bias:0"]


def test_get_best_model_patterns(tmp_path):
    d = tmp_path / "model"; d.mkdir()
    with pytest.raises(OSError):
        F.get_best_model(str(d))
    (d / "model_weights.npz").write_bytes(b"")
    assert F.get_best_model(str(d)).endswith("model_weights.npz")
    for n in ("@epoch_03_val_dice_0.71234.h5", "@epoch_07_val_dice_0.80011.npz", "@epoch_09_val_dice_0.79000.h5"):
        (d / n).write_bytes(b"")
    assert F.get_best_model(str(d)).endswith("@epoch_07_val_dice_0.80011.npz")
    d2 = tmp_path / "m2"; d2.mkdir()
    for n in ("@epoch_01_val_loss_0.50000.h5", "@epoch_02_val_loss_0.30000.h5"):
        (d2 / n).write_bytes(b"")
    assert F.get_best_model(str(d2)).endswith("@epoch_02_val_loss_0.30000.h5")


def test_missing_optional_packages_fail_loudly(tmp_path):
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="h5py"):
            F.load_keras_h5(str(tmp_path / "x.h5"))
        with pytest.raises(ImportError, match="h5py"):
            _model().save_weights(str(tmp_path / "x.h5"))


def test_keras_h5_round_trip(tmp_path):
    pytest.importorskip("h5py")
    m = _model()
    p = str(tmp_path / "@epoch_01_val_dice_0.50000.h5")
    m.save_weights(p)
    w = F.load_keras_h5(p)
    d = m.get_weights_dict()
    assert sorted(w) == sorted(d)
    for k in d:
        np.testing.assert_array_equal(w[k], d[k])
    m2 = _model(); m2.load_weights(p, by_name=True)
    for k, v in m2.get_weights_dict().items():
        np.testing.assert_array_equal(v, d[k])


def test_nifti_round_trip(tmp_path):
    lab = (np.arange(4 * 5 * 6).reshape(4, 5, 6) % 3).astype(np.uint8)
    aff = np.diag([1.0, 0.8, 1.5, 1.0])
    p = str(tmp_path / "x_PRED.nii.gz")
    F.save_nifti(p, lab, aff)
    img, a2 = F.load_nifti(p)
    np.testing.assert_array_equal(img[..., 0].astype(np.uint8), lab)
    np.testing.assert_allclose(a2, aff)
