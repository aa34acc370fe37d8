"""On-disk formats: the Keras name list of Appendix B, checkpoint selection, and the optional
Keras .h5 adapter -- through the HDF5 C library (hdf5.py) against files the REAL h5py wrote (tests/golden/keras_unet_d1*.h5)
and back through the real h5py where an interpreter has it; NIfTI is native -- tests/test_nifti_host.py."""
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


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CONDA_PY = "/opt/conda/bin/python3.9"            # the build image's interpreter that HAS h5py (real-h5py cross checks)


def _have_libhdf5():
    from multiplanarunet_amd import hdf5
    return hdf5.available()


needs_libhdf5 = pytest.mark.skipif(not _have_libhdf5(), reason="no libhdf5 on this host")


def test_no_h5_backend_fails_loudly(tmp_path, monkeypatch):
    from multiplanarunet_amd import hdf5
    monkeypatch.setenv("MPU_H5_BACKEND", "libhdf5")
    monkeypatch.setattr(hdf5, "_lib", None)
    monkeypatch.setattr(hdf5, "_err", "no usable libhdf5 found")
    with pytest.raises(ImportError, match="libhdf5"):
        F.load_keras_h5(str(tmp_path / "x.h5"))
    with pytest.raises(ImportError, match="h5py"):
        _model().save_weights(str(tmp_path / "x.h5"))
    monkeypatch.setenv("MPU_H5_BACKEND", "nope")
    with pytest.raises(ValueError):
        F.load_keras_h5(str(tmp_path / "x.h5"))


@needs_libhdf5
@pytest.mark.parametrize("name", ["keras_unet_d1.h5", "keras_unet_d1.full.h5"])
def test_libhdf5_backend_reads_files_written_by_real_h5py(name, monkeypatch):
    """tests/golden/keras_unet_d1*.h5 were written by h5py 3.3 / HDF5 1.10.6 following Keras' save_weights step by step
    (tests/golden/make_keras_h5_fixture.py): weightless layers with empty weight_names, groups in sorted order, bytes
    (fixed-length) and str (variable-length) attributes, the weights under /model_weights in the full-model file."""
    monkeypatch.setenv("MPU_H5_BACKEND", "libhdf5")
    w = F.load_keras_h5(os.path.join(GOLDEN, name))
    with np.load(os.path.join(GOLDEN, "keras_unet_d1.npz")) as z:
        want = {k: z[k] for k in z.files}
    assert sorted(w) == sorted(want) and len(w) == 32
    for k in want:
        assert w[k].dtype == np.float32 and w[k].shape == want[k].shape, k
        np.testing.assert_array_equal(w[k], want[k], err_msg=k)


@needs_libhdf5
def test_model_loads_a_reference_style_checkpoint_by_name(monkeypatch):
    from multiplanarunet_amd.unet import UNet
    monkeypatch.setenv("MPU_H5_BACKEND", "libhdf5")
    m = UNet(n_classes=3, dim=16, depth=1, complexity_factor=1.0 / 256, device="cpu", logger=lambda *a, **k: None)
    m.load_weights(os.path.join(GOLDEN, "keras_unet_d1.h5"), by_name=True)
    assert m.missing_on_load == []
    d = m.get_weights_dict()
    with np.load(os.path.join(GOLDEN, "keras_unet_d1.npz")) as z:
        for k in z.files:
            mine = k.replace("conv2d_7/", "conv2d/")                       # the auto-named 1x1 head lands on the model's head
            np.testing.assert_array_equal(d[mine], z[k], err_msg=k)
    assert len(d) == 32


@needs_libhdf5
def test_file_written_through_libhdf5_is_read_by_real_h5py_like_keras_does(tmp_path, monkeypatch):
    import subprocess
    monkeypatch.setenv("MPU_H5_BACKEND", "libhdf5")
    m = _model()
    p = str(tmp_path / "@epoch_01_val_dice_0.50000.h5")
    m.save_weights(p)
    d = m.get_weights_dict()
    w = F.load_keras_h5(p)
    assert sorted(w) == sorted(d)
    for k in d:
        np.testing.assert_array_equal(w[k], d[k])
    if not os.path.exists(CONDA_PY):
        pytest.skip("no interpreter with h5py on this host")
    # keras/saving/hdf5_format.py load_weights_from_hdf5_group_by_name, reduced to its reads, under the real h5py
    reader = (
        "import sys, h5py, numpy as np\n"
        "dec = lambda n: n.decode('utf8') if hasattr(n, 'decode') else n\n"
        "out = {}\n"
        "with h5py.File(sys.argv[1], 'r') as f:\n"
        "    assert dec(f.attrs['backend']) == 'tensorflow' and dec(f.attrs['keras_version'])\n"
        "    for name in [dec(n) for n in f.attrs['layer_names']]:\n"
        "        g = f[name]\n"
        "        for wn in [dec(n) for n in g.attrs['weight_names']]:\n"
        "            out[wn.rsplit(':', 1)[0]] = np.asarray(g[wn])\n"
        "np.savez(sys.argv[2], **out)\n")
    r = subprocess.run([CONDA_PY, "-c", reader, p, str(tmp_path / "back.npz")], capture_output=True, text=True,
                       env={"PATH": os.environ.get("PATH", "")})
    if r.returncode != 0 and "No module named" in r.stderr:
        pytest.skip("interpreter without h5py: " + r.stderr.strip().splitlines()[-1])
    assert r.returncode == 0, r.stderr
    with np.load(str(tmp_path / "back.npz")) as z:
        assert sorted(z.files) == sorted(d)
        for k in d:
            assert z[k].dtype == np.float32
            np.testing.assert_array_equal(z[k], d[k], err_msg=k)


def test_long_name_lists_are_split_like_keras_does():
    names = ["layer_%04d_with_a_rather_long_name_to_fill_the_header" % i for i in range(3000)]
    chunks = F._name_chunks("layer_names", names)
    assert [n for n, _ in chunks][:3] == ["layer_names0", "layer_names1", "layer_names2"]
    width = max(len(n) for n in names)
    assert all(width * len(v) <= F.KERAS_ATTR_LIMIT for _, v in chunks)
    store = {n: [x.decode() for x in v] for n, v in chunks}
    assert F._chunked_names(store.__getitem__, store.__contains__, "layer_names") == names
    assert F._name_chunks("weight_names", ["a/kernel:0", "a/bias:0"]) == [("weight_names", [b"a/kernel:0", b"a/bias:0"])]


def test_keras_h5_round_trip_h5py_backend(tmp_path, monkeypatch):
    pytest.importorskip("h5py")
    monkeypatch.setenv("MPU_H5_BACKEND", "h5py")
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


@needs_libhdf5
def test_convert_keeps_layers_outside_the_canonical_name_list(tmp_path, monkeypatch):
    """A reference checkpoint whose 1x1 head is auto-named conv2d_7: .h5 -> dict -> .h5 keeps that layer (h5_entries
    appends layers it does not know, variables in Keras' order)."""
    monkeypatch.setenv("MPU_H5_BACKEND", "libhdf5")
    w = F.load_keras_h5(os.path.join(GOLDEN, "keras_unet_d1.h5"))
    p = str(tmp_path / "again.h5")
    F.save_keras_h5(p, w, depth=1)
    w2 = F.load_keras_h5(p)
    assert sorted(w2) == sorted(w) and "conv2d_7/kernel" in w2
    for k in w:
        np.testing.assert_array_equal(w2[k], w[k])
    ent = F.h5_entries(w, depth=1)
    assert ent[-1][0] == "conv2d_7" and [n for n, _ in ent[-1][1]] == ["conv2d_7/kernel:0", "conv2d_7/bias:0"]
