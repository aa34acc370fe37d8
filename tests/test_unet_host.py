"""Host-side logic of the UNet drop-in (no GPU): constructor contract, layout, weight I/O."""
import numpy as np
import pytest
from multiplanarunet_amd.unet import UNet
from oracle import unet_ref as U

quiet = lambda *a, **k: None


def test_constructor_contract():
    with pytest.raises(ValueError):
        UNet(n_classes=3, logger=quiet, device="cpu")
    m = UNet(n_classes=3, dim=32, n_channels=1, depth=2, logger=quiet, device="cpu",
             model_class_name="UNet", l1_reg=False, biased_output_layer=True)   # extra YAML keys ignored
    assert m.img_shape == (32, 32, 1) and m.n_classes == 3 and m.depth == 2
    assert m.label_crop.shape == (2, 2) and not m.label_crop.any()
    assert m.out_activation == "softmax" and m.padding == "same" and not m.flatten_output
    for bad in (dict(padding="valid"), dict(activation="elu"), dict(kernel_size=5), dict(dim=34)):
        kw = dict(n_classes=3, dim=32, depth=2, logger=quiet, device="cpu")
        kw.update(bad)
        with pytest.raises(NotImplementedError):
            UNet(**kw)


def test_param_count_and_receptive_field_match_reference_numbers():
    m = UNet(n_classes=3, dim=128, n_channels=1, depth=4, complexity_factor=1, logger=quiet, device="cpu")
    assert m.count_params() == 31030723 + 15616
    # receptive field of the contracting path per utils/conv_arithmetics.py:57-112
    rf, jump = 1, 1
    for _ in range(4):
        rf += 2 * jump; rf += 2 * jump; jump *= 2; rf += jump
    rf += 4 * jump
    assert list(m.receptive_field) == [rf, rf]
    m2 = UNet(n_classes=3, dim=64, depth=4, complexity_factor=2, logger=quiet, device="cpu")
    assert m2.filters == [90, 181, 362, 724, 1448]          # default YAML complexity_factor=2


def test_weight_names_shapes_and_roundtrip(tmp_path):
    m = UNet(n_classes=4, dim=32, n_channels=2, depth=2, complexity_factor=2, logger=quiet, device="cpu", seed=1)
    ref = U.init_weights(4, 2, 2, 2, seed=0)
    d = m.get_weights_dict()
    assert set(d) == set(ref)
    for k in ref:
        assert d[k].shape == ref[k].shape, k
    # Keras defaults at init
    assert np.all(d["encoder_L0_BN/gamma"] == 1) and np.all(d["encoder_L0_BN/moving_variance"] == 1)
    assert np.all(d["bottom_conv1/bias"] == 0)
    lim = np.sqrt(6.0 / (9 * 2 + 9 * 90))
    assert np.abs(d["encoder_L0_conv1/kernel"]).max() <= lim
    m.set_weights_dict(ref)
    d2 = m.get_weights_dict()
    for k in ref:
        np.testing.assert_array_equal(d2[k], ref[k])
    # ordered Keras-style list + file round trip
    lst = m.get_weights()
    assert len(lst) == 2 * 13 + 4 * 7
    m.set_weights(lst)
    p = tmp_path / "w.npz"
    m.save_weights(str(p))
    m3 = UNet(n_classes=4, dim=32, n_channels=2, depth=2, complexity_factor=2, logger=quiet, device="cpu", seed=7)
    m3.load_weights(str(p), by_name=True)
    for k, v in m3.get_weights_dict().items():
        np.testing.assert_array_equal(v, ref[k])
    with pytest.raises(ValueError):
        m.set_weights_dict({"conv2d/bias": np.zeros(5, np.float32)})


def test_no_cpu_execution_path():
    from multiplanarunet_amd._lib import MpuError
    m = UNet(n_classes=3, dim=16, depth=1, logger=quiet, device="cpu")
    with pytest.raises(MpuError):
        m.predict_on_batch(np.zeros((1, 16, 16, 1), np.float32))
