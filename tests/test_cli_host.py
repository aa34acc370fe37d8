"""CLI shims: argument surface and validation rules of mp train / mp predict (no GPU)."""
import os
import numpy as np
import pytest
from multiplanarunet_amd.cli import train as T, predict as P, common as C, mp as MP


def test_train_flags_and_validation(tmp_path):
    a = T.get_argparser().parse_args([])
    assert a.train_images_per_epoch == 2500 and a.val_images_per_epoch == 3500 and a.num_GPUs == 1
    for flag in ("--continue_training", "--overwrite", "--just_one", "--no_val", "--no_images", "--debug"):
        assert getattr(T.get_argparser().parse_args([flag]), flag[2:])
    with pytest.raises(ValueError):
        T.validate_args(T.get_argparser().parse_args(["--continue_training", "--overwrite"]))
    with pytest.raises(ValueError):
        T.validate_args(T.get_argparser().parse_args(["--train_images_per_epoch", "0"]))
    with pytest.raises(ValueError):
        T.validate_args(T.get_argparser().parse_args(["--force_GPU", "0", "--num_GPUs", "2"]))
    with pytest.raises(NotImplementedError):
        T.validate_args(T.get_argparser().parse_args(["--num_GPUs", "0"]))     # reference: CPU mode
    with pytest.raises(RuntimeError):
        C.validate_project_dir(str(tmp_path))


def test_predict_flags():
    a = P.get_argparser().parse_args(["--sum_fusion", "--no_eval", "--continue", "--on_val", "-f", "x.npz"])
    assert a.sum_fusion and a.no_eval and a.continue_ and a.on_val and a.f == "x.npz" and a.out_dir == "predictions"


def test_hparams_defaults_and_yaml_anchors(tmp_path):
    (tmp_path / "train_hparams.yaml").write_text(
        "__CB_x: &X\n  nickname: x\nbuild: &BUILD\n  n_classes: 3\n  dim: 64\n  complexity_factor: 1\n"
        "fit:\n  batch_size: 8\n  callbacks: [*X]\n__VERSION__: Null\n")
    hp = C.load_hparams(str(tmp_path))
    assert hp["build"]["n_classes"] == 3 and hp["build"]["depth"] == 4 and hp["build"]["model_class_name"] == "UNet"
    assert hp["fit"]["batch_size"] == 8 and hp["fit"]["optimizer_kwargs"]["lr"] == 5e-5
    assert hp["fit"]["views"] == 6 and hp["fit"]["scaler"] == "RobustScaler" and hp["fit"]["bg_value"] == "1pct"


def test_views_file_and_best_model(tmp_path):
    v = C.load_or_create_views(str(tmp_path), 6, seed=1)
    assert v.shape == (6, 3) and np.allclose(np.linalg.norm(v, axis=1), 1) and (v[:, 2] >= 0).all()
    ang = np.rad2deg(np.arccos(np.clip(v @ v.T, -1, 1)))[np.triu_indices(6, 1)]
    assert ang.min() > 15
    np.testing.assert_array_equal(np.load(tmp_path / "views.npz")["arr_0"], v)
    m = tmp_path / "model"; m.mkdir()
    for n in ("@epoch_03_val_dice_0.71234.npz", "@epoch_07_val_dice_0.80011.npz", "model_weights.npz"):
        (m / n).write_bytes(b"")
    assert P.best_model_path(str(m)).endswith("@epoch_07_val_dice_0.80011.npz")


def test_mp_dispatch_rejects_out_of_scope_scripts():
    with pytest.raises(SystemExit):
        MP.entry_func(["cv_split"])
    assert MP.entry_func(["--help"]) == 0


def test_plane_basis_fast_agrees_with_reference_construction():
    """The sampler's pure-Python basis equals interpolation.plane_basis up to the reference's float32 round trips."""
    from multiplanarunet_amd.interpolation import plane_basis
    from multiplanarunet_amd.data import plane_basis_fast
    rng = np.random.RandomState(0)
    for i in range(500):
        v = rng.randn(3)
        if i % 7 == 0:
            v[:2] *= 1e-3
        if i % 50 == 0:
            v = np.array([0.0, 0.0, 1.0])
        nz = rng.randn(3) * 0.1 if i % 2 else None
        a = plane_basis(v, nz)
        b = np.array(plane_basis_fast(v, nz)).reshape(3, 3)
        assert np.abs(a - b).max() < 1e-6
        assert np.abs(b.T @ b - np.eye(3)).max() < 1e-6          # orthonormal


def test_audited_hparams_are_persisted_and_required(tmp_path):
    """Auditor write-back: dim / real_space_span / n_channels / n_classes land in train_hparams.yaml at train time;
    predict / train_fusion refuse to re-audit (mpunet/bin/train.py:210-228)."""
    (tmp_path / "train_hparams.yaml").write_text(
        "build:\n  model_class_name: UNet\n  complexity_factor: 1\nfit:\n  batch_size: 8\n__VERSION__: Null\n")
    hp = C.load_hparams(str(tmp_path))
    with pytest.raises(RuntimeError) as e:
        C.require_audited_hparams(hp, "mp predict")
    assert "build.dim" in str(e.value) and "fit.real_space_span" in str(e.value)
    hp["build"].update(dim=64, n_channels=1, n_classes=3)
    hp["fit"]["real_space_span"] = np.float64(63.5)
    assert C.save_audited_hparams(str(tmp_path), hp)
    assert not C.save_audited_hparams(str(tmp_path), hp)               # idempotent
    hp2 = C.load_hparams(str(tmp_path))
    C.require_audited_hparams(hp2, "mp predict")
    assert hp2["build"]["dim"] == 64 and hp2["build"]["n_classes"] == 3 and hp2["build"]["n_channels"] == 1
    assert hp2["fit"]["real_space_span"] == 63.5 and hp2["fit"]["batch_size"] == 8
    assert hp2["build"]["complexity_factor"] == 1                      # untouched keys survive the rewrite


def test_fusion_weights_belong_to_one_checkpoint(tmp_path):
    m = str(tmp_path / "model")
    p = C.fusion_weights_path(m, os.path.join(m, "@epoch_07_val_dice_0.80011.npz"))
    assert p == os.path.join(m, "fusion_weights", "@epoch_07_val_dice_0.80011_fusion_weights.npz")
    assert C.fusion_weights_path(m, os.path.join(m, "model_weights.npz")).endswith("model_weights_fusion_weights.npz")


def test_audited_hparams_patch_keeps_comments_and_layout(tmp_path):
    """the write-back patches only the audited lines (the reference's YAMLHParams edits its string
    representation in place, mpunet/hyperparameters/hparams.py:161-221): comments, `Null` placeholders with
    trailing comments, blank lines and unrelated sections survive byte for byte."""
    text = ("# project file\nbuild:\n  #\n  # Hyperparameters passed to the Model.build\n  #\n  model_class_name: \"UNet\"\n"
            "  dim: Null  # audited\n  n_classes: Null\n  n_channels: Null\n  complexity_factor: 2\n\n"
            "fit:\n  # training\n  batch_size: 16\n  real_space_span: Null\n\n__VERSION__: Null\n")
    (tmp_path / "train_hparams.yaml").write_text(text)
    hp = C.load_hparams(str(tmp_path))
    hp["build"].update(dim=128, n_channels=2, n_classes=5)
    hp["fit"]["real_space_span"] = 200.25
    assert C.save_audited_hparams(str(tmp_path), hp)
    out = (tmp_path / "train_hparams.yaml").read_text()
    expect = (text.replace("dim: Null  # audited", "dim: 128  # audited").replace("n_classes: Null", "n_classes: 5")
              .replace("n_channels: Null", "n_channels: 2").replace("real_space_span: Null", "real_space_span: 200.25"))
    assert out == expect
    # a key that is missing from its section is added at the end of that section only
    (tmp_path / "train_hparams.yaml").write_text("build:\n  complexity_factor: 1  # keep\n\nfit:\n  batch_size: 8\n")
    hp = C.load_hparams(str(tmp_path))
    hp["build"].update(dim=64, n_channels=1, n_classes=3)
    hp["fit"]["real_space_span"] = 63.5
    assert C.save_audited_hparams(str(tmp_path), hp)
    out = (tmp_path / "train_hparams.yaml").read_text()
    assert "complexity_factor: 1  # keep" in out and out.index("dim: 64") < out.index("fit:")
    hp2 = C.load_hparams(str(tmp_path))
    assert hp2["build"]["dim"] == 64 and hp2["fit"]["real_space_span"] == 63.5 and hp2["fit"]["batch_size"] == 8
