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


def test_flat_layout_follows_keras_creation_order_and_ready_points():
    """Each block's BatchNormalization gamma/beta sit next to that block's convs in the flat parameter buffer, so
    every tensor lies at or above the gradient-ready point of its block and below the previous point
    (mpu_unet_grad_ready_points: head, up blocks last->first, bottom, encoder levels last->first)."""
    from multiplanarunet_amd.unet import UNet
    for depth, cf in ((4, 1), (2, 2), (1, 1)):
        m = UNet(n_classes=3, dim=32 * 2 ** max(0, depth - 2), depth=depth, complexity_factor=cf, device="cpu",
                 logger=lambda *a, **k: None)
        pts = m.grad_ready_points()
        assert len(pts) == 2 * depth + 2 and pts == sorted(pts, reverse=True) and pts[-1] == 0
        n = m.params.numel()
        blocks = ["conv2d"] + ["upsample_L%d" % j for j in range(depth - 1, -1, -1)] + ["bottom"] + \
                 ["encoder_L%d" % i for i in range(depth - 1, -1, -1)]
        off_sorted = []
        for nm in m._order:
            kind, off, ps, ls = m._tensors[nm]
            if kind != 0:
                continue
            off_sorted.append((off, nm))
            k = [i for i, b in enumerate(blocks) if nm.startswith(b + "_") or nm.startswith(b + "/")]
            assert len(k) == 1, nm
            k = k[0]
            hi = n if k == 0 else pts[k - 1]
            assert pts[k] <= off and off + int(np.prod(ps)) <= hi, (nm, off, pts, k)
        # trainable tensors appear in the buffer in Keras creation order
        order = [nm for _, nm in sorted(off_sorted)]
        keras = [nm for nm in m._keras_order() if not nm.endswith(("moving_mean", "moving_variance"))]
        assert order == keras


def test_load_weights_reports_missing_tensors_and_maps_auto_named_head(tmp_path):
    """Keras names the unnamed 1x1 head conv2d_<N> when the process built other models first
    (mpunet/models/unet.py:211); such a file must still load the head, and tensors a file does not supply are
    reported instead of silently staying at their initial values."""
    from multiplanarunet_amd.unet import UNet
    msgs = []
    m = UNet(n_classes=3, dim=32, depth=2, complexity_factor=0.25, device="cpu", logger=msgs.append, seed=0)
    d = m.get_weights_dict()
    d["conv2d/kernel"] = d["conv2d/kernel"] + 1.5
    f = {k.replace("conv2d/", "conv2d_7/").replace("/", "__"): v for k, v in d.items() if not k.startswith("bottom_BN")}
    np.savez(tmp_path / "w.npz", **f)
    m2 = UNet(n_classes=3, dim=32, depth=2, complexity_factor=0.25, device="cpu", logger=msgs.append, seed=1)
    m2.load_weights(str(tmp_path / "w.npz"))
    np.testing.assert_array_equal(m2.get_weights_dict()["conv2d/kernel"], d["conv2d/kernel"])
    assert m2.missing_on_load == sorted("bottom_BN/" + v for v in ("gamma", "beta", "moving_mean", "moving_variance"))
    assert any("bottom_BN" in s and "not in the file" in s for s in msgs)
    with pytest.raises(KeyError):
        m2.load_weights(str(tmp_path / "w.npz"), by_name=False)


def test_environment_switch_table_is_the_only_getenv_and_is_documented():
    """csrc/env.h holds every environment switch of the library (name, kind, default, description): no other getenv in
    csrc/, mpu_env_describe lists exactly the table with the defaults in force here, and DESIGN.md names every switch."""
    import ctypes as C, os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "multiplanarunet_amd", "csrc")
    offenders = []
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")) and f != "env.hip":
            src = open(os.path.join(csrc, f)).read()
            if re.search(r"\bgetenv\s*\(", src):
                offenders.append(f)
    assert offenders == [], offenders
    table = re.findall(r'X\((\w+), "(MPU_\w+)", (ENV_\w+), (-?\d+),', open(os.path.join(csrc, "env.h")).read())
    assert len(table) >= 25 and len({n for _, n, _, _ in table}) == len(table)
    from multiplanarunet_amd import _lib
    lib = _lib.load()
    n = lib.mpu_env_describe(None, 0)
    buf = C.create_string_buffer(int(n) + 1)
    lib.mpu_env_describe(buf, n + 1)
    rows = [l.split("\t") for l in buf.value.decode().splitlines()]
    assert [r[0] for r in rows] == [name for _, name, _, _ in table]
    design = open(os.path.join(root, "DESIGN.md")).read()
    for (ident, name, kind, dflt), row in zip(table, rows):
        assert len(row) == 4 and row[3].strip(), row
        assert int(row[2]) == int(dflt), row
        if name not in os.environ:
            assert int(row[1]) == int(dflt), row                 # unset: the default is in force
        assert name in design, "DESIGN.md does not mention " + name


def test_auto_batch_fills_the_rounds_of_the_persistent_predict_kernel():
    """UNet.auto_batch (predict with batch_size=None): even chunks under the 2-GiB operand bound; for bf16 the chunk count
    among the smallest three whose chunk size leaves the fewest idle rounds to conv_halo16p (256 workgroups) at the levels
    with 128-channel tiles -- never chunks under 64 images (measured: 276 planes of 256 x 256 as 3 x 92 beat 2 x 138 by
    1.5 %, 4 x 69 lose 6 %)."""
    from multiplanarunet_amd.unet import UNet
    q = lambda *a, **k: None
    m = UNet(n_classes=3, dim=256, depth=4, complexity_factor=1, dtype="bf16", device="cpu", logger=q)
    assert m.max_batch() == 255
    assert [m.auto_batch(n) for n in (276, 184, 92, 138, 69, 35, 8, 1)] == [92, 92, 92, 138, 69, 35, 8, 1]
    # level 3 (512 channels, 32 x 32): 138 planes = 276 pixel tiles on 64 workgroups per n-tile = 4.3 rounds, 92 planes 2.9
    assert m._round_fill(92) > m._round_fill(138) > m._round_fill(69)
    m5 = UNet(n_classes=5, dim=512, n_channels=2, depth=4, complexity_factor=1, dtype="bf16", device="cpu", logger=q)
    assert m5.max_batch() == 63 and m5.auto_batch(532) == 60            # the operand bound decides: 9 chunks
    mf = UNet(n_classes=3, dim=256, depth=4, complexity_factor=1, dtype="f32", device="cpu", logger=q)
    assert mf.auto_batch(276) == 92                                      # f32: 3 chunks by the operand bound alone
