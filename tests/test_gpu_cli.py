"""End-to-end `mp train` -> `mp predict` on synthetic toy volumes (train-time sampler, checkpoints, fused predict)."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_mp_train_then_predict_synthetic(tmp_path):
    from multiplanarunet_amd.cli import mp
    proj = tmp_path / "proj"
    proj.mkdir()
    (proj / "train_hparams.yaml").write_text(
        "build:\n  model_class_name: UNet\n  n_classes: 3\n  n_channels: 1\n  dim: 64\n  depth: 3\n"
        "  complexity_factor: 0.0625\n  out_activation: softmax\n  seed: 0\n"
        "fit:\n  views: 3\n  noise_sd: 0.1\n  real_space_span: 64.0\n  batch_size: 8\n  n_epochs: 2\n"
        "  optimizer: Adam\n  optimizer_kwargs: {lr: 1.0e-3, decay: 0.0, beta_1: 0.9, beta_2: 0.999, epsilon: 1.0e-8}\n"
        "  loss: SparseCategoricalCrossentropy\n  fg_batch_fraction: 0.5\n  bg_value: 1pct\n  scaler: RobustScaler\n"
        "  augmenters: [{cls_name: Elastic2D, kwargs: {alpha: [0, 100], sigma: [6, 9], apply_prob: 0.333}}]\n")
    mp.entry_func(["train", "--project_dir", str(proj), "--synthetic", "4", "--epochs", "6",
                   "--train_images_per_epoch", "160", "--val_images_per_epoch", "32"])
    assert (proj / "model" / "model_weights.npz").exists() and (proj / "views.npz").exists()
    assert any(f.startswith("@epoch") for f in os.listdir(proj / "model"))
    from multiplanarunet_amd import hdf5, formats
    if hdf5.available():            # checkpoints in the reference's own format (Keras .h5 through libhdf5), as the YAML default names them
        assert all(f.endswith(".h5") for f in os.listdir(proj / "model") if f.startswith("@epoch"))
        w5 = formats.load_keras_h5(str(proj / "model" / "model_weights.h5"))
        with np.load(proj / "model" / "model_weights.npz") as z:
            assert sorted(w5) == sorted(k.replace("__", "/") for k in z.files)
            for k in z.files:
                np.testing.assert_array_equal(w5[k.replace("__", "/")], z[k])
    log = (proj / "logs" / "training.csv").read_text().strip().splitlines()
    losses = [float(l.split(",")[1]) for l in log[1:]]
    assert len(losses) == 6 and losses[-1] < losses[0]
    mp.entry_func(["predict", "--project_dir", str(proj), "--synthetic", "1", "--sum_fusion", "--overwrite"])
    out = proj / "predictions" / "nii_files" / "toy_5000_PRED.npz"
    assert out.exists()
    lab = np.load(out)["labels"]
    assert lab.shape == (64, 64, 64) and lab.dtype == np.uint8
    res = (proj / "predictions" / "csv" / "results.csv").read_text().splitlines()
    mean_dice = float(res[1].split(",")[1])
    assert mean_dice > 0.5, res
    # per-view evaluation inside the loop (bin/predict.py:334-346; round 6): one row per (image, view) with the mapped Dice
    pv = (proj / "predictions" / "csv" / "per_view.csv").read_text().splitlines()
    assert pv[0].startswith("image,view_index,view,mean_dice,class_0") and len(pv) == 1 + 3
    assert all(0.2 < float(r.split(",")[3]) <= 1.0 for r in pv[1:]), pv
    # fusion-model training on the project, then predict with the learned FusionLayer
    hist = mp.entry_func(["train_fusion", "--project_dir", str(proj), "--synthetic", "2", "--epochs", "4",
                          "--images_per_round", "2", "--batch_size", "65536", "--seed", "0"])
    fdir = proj / "model" / "fusion_weights"
    files = os.listdir(fdir)
    assert len(files) == 1 and files[0].endswith("_fusion_weights.npz")
    z = np.load(fdir / files[0])
    assert z["W"].shape == (3, 3) and z["b"].shape == (1, 3) and not np.allclose(z["W"], 1.0)
    mp.entry_func(["predict", "--project_dir", str(proj), "--synthetic", "1", "--overwrite"])
    res2 = (proj / "predictions" / "csv" / "results.csv").read_text().splitlines()
    assert float(res2[1].split(",")[1]) > 0.5, res2
    with pytest.raises(OSError):
        mp.entry_func(["train_fusion", "--project_dir", str(proj), "--synthetic", "1"])       # exists, no --overwrite
    # the same volume as a NIfTI file (native reader / writer, multiplanarunet_amd/nifti.py): identical label map, written
    # as <id>_PRED.nii.gz with the input's affine (mpunet/bin/predict.py:90-117)
    from multiplanarunet_amd.data import make_toy_volume
    from multiplanarunet_amd.formats import save_nifti
    from multiplanarunet_amd.nifti import read_nifti
    img, lab_true, aff = make_toy_volume(64, 5000)
    save_nifti(str(tmp_path / "toy_5000.v1.nii.gz"), img[..., 0], aff)
    save_nifti(str(tmp_path / "toy_5000_labels.nii.gz"), lab_true, aff)
    lab_npz = np.load(out)["labels"]
    mp.entry_func(["predict", "--project_dir", str(proj), "-f", str(tmp_path / "toy_5000.v1.nii.gz"),
                   "-l", str(tmp_path / "toy_5000_labels.nii.gz"), "--out_dir", "predictions_nii"])
    pred, aff2, hdr = read_nifti(str(proj / "predictions_nii" / "nii_files" / "toy_5000_PRED.nii.gz"), scaled=False)
    assert pred.dtype == np.uint8 and pred.shape == (64, 64, 64)
    np.testing.assert_array_equal(aff2, aff)
    np.testing.assert_array_equal(pred, lab_npz)
    res3 = (proj / "predictions_nii" / "csv" / "results.csv").read_text().splitlines()
    assert res3[1].startswith("toy_5000,") and res3[1].split(",")[1] == res2[1].split(",")[1], (res3, res2)
    # `--num_GPUs 2` outside a torchrun job: the script re-launches itself as two ranks (here both on GPU 0 over gloo, the
    # testing aid of tests/test_gpu_bench_multi.py) and the plane-sharded predict writes the same kind of result
    import subprocess, sys
    env = dict(os.environ, MPU_SHARE_GPU="1", MPU_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, "-m", "multiplanarunet_amd.cli.mp", "predict", "--project_dir", str(proj), "--synthetic", "1",
                        "--num_GPUs", "2", "--out_dir", "predictions_2gpu"], env=env, capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lab2 = np.load(proj / "predictions_2gpu" / "nii_files" / "toy_5000_PRED.npz")["labels"]
    assert lab2.shape == (64, 64, 64) and (lab2 != lab_npz).mean() <= 1e-3
    res4 = (proj / "predictions_2gpu" / "csv" / "results.csv").read_text().splitlines()
    assert abs(float(res4[1].split(",")[1]) - float(res2[1].split(",")[1])) <= 2e-3, (res4, res2)
    # `mp train_fusion --num_GPUs 2` (round 5): two ranks (sharing GPU 0 over gloo here) deal the round's images,
    # fit data parallel and write ONE weights file; the learned layer predicts as well as the one-rank fit's
    before = set(os.listdir(fdir))
    r = subprocess.run([sys.executable, "-m", "multiplanarunet_amd.cli.mp", "train_fusion", "--project_dir", str(proj), "--synthetic", "2",
                        "--epochs", "4", "--images_per_round", "2", "--batch_size", "65536", "--seed", "0", "--overwrite",
                        "--num_GPUs", "2"], env=env, capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert set(os.listdir(fdir)) == before and len(before) == 1            # one writer, the same single file
    assert r.stdout.count("Saved fusion weights:") == 1                    # rank 0 alone reports
    z2 = np.load(fdir / files[0])
    assert z2["W"].shape == (3, 3) and not np.allclose(z2["W"], 1.0)
    mp.entry_func(["predict", "--project_dir", str(proj), "--synthetic", "1", "--overwrite"])
    res5 = (proj / "predictions" / "csv" / "results.csv").read_text().splitlines()
    assert float(res5[1].split(",")[1]) > 0.5 and abs(float(res5[1].split(",")[1]) - float(res2[1].split(",")[1])) <= 2e-2, (res5, res2)


def test_mp_train_continue_training_resumes_where_the_reference_would(tmp_path):
    """`mp train --continue_training` end to end (mpunet/models/model_init.py:23-47): the newest @epoch_ checkpoint is loaded, the
    epoch loop restarts at its number + 1, logs/training.csv is cut back and appended to, the logged learning rate comes back."""
    import re
    from multiplanarunet_amd.cli import mp
    proj = tmp_path / "proj"
    proj.mkdir()
    (proj / "train_hparams.yaml").write_text(
        "build:\n  model_class_name: UNet\n  n_classes: 3\n  n_channels: 1\n  dim: 64\n  depth: 3\n"
        "  complexity_factor: 0.0625\n  out_activation: softmax\n  seed: 0\n"
        "fit:\n  views: 3\n  noise_sd: 0.1\n  real_space_span: 64.0\n  batch_size: 8\n  n_epochs: 2\n"
        "  optimizer: Adam\n  optimizer_kwargs: {lr: 2.0e-3, decay: 0.0, beta_1: 0.9, beta_2: 0.999, epsilon: 1.0e-8}\n"
        "  loss: SparseCategoricalCrossentropy\n  fg_batch_fraction: 0.5\n  bg_value: 1pct\n  scaler: RobustScaler\n")
    common = ["train", "--project_dir", str(proj), "--synthetic", "4", "--train_images_per_epoch", "96", "--val_images_per_epoch", "32"]
    mp.entry_func(common + ["--epochs", "4"])
    csv = proj / "logs" / "training.csv"
    rows0 = csv.read_text().strip().splitlines()
    head = rows0[0].split(",")
    ep0 = [int(r.split(",")[0]) for r in rows0[1:]]
    assert ep0 == [0, 1, 2, 3]
    loss0 = [float(r.split(",")[head.index("loss")]) for r in rows0[1:]]
    ck = [int(re.findall(r"@epoch_(\d+)_", f)[0]) for f in os.listdir(proj / "model") if f.startswith("@epoch")]
    assert ck, os.listdir(proj / "model")
    N = max(ck)                                             # Keras' 1-based epoch number of the newest (= best kept) checkpoint
    lr_logged = [float(r.split(",")[head.index("lr")]) for r in rows0[1:]]
    with pytest.raises(OSError):                            # neither --overwrite nor --continue_training: refuses, as the reference
        mp.entry_func(common + ["--epochs", "4"])
    mp.entry_func(common + ["--epochs", "7", "--continue_training"])
    rows1 = csv.read_text().strip().splitlines()
    ep1 = [int(r.split(",")[0]) for r in rows1[1:]]
    kept = [e for e in ep0 if e <= N]                       # rows [0, N] stay (utils.py:145-163) ...
    assert ep1 == kept + list(range(N + 1, 7)), (N, ep1)    # ... and the loop continues at N + 1 (model_init.py:43-47)
    assert rows1[1:1 + len(kept)] == rows0[1:1 + len(kept)]                  # the kept rows are the old ones, untouched
    loss1 = [float(r.split(",")[head.index("loss")]) for r in rows1[1 + len(kept):]]
    assert loss1 and loss1[0] < loss0[0] * 0.9, (loss0, loss1)               # trained weights came back, not a fresh network
    lr1 = float(rows1[1 + len(kept)].split(",")[head.index("lr")])
    want_lr = lr_logged[min(N, len(lr_logged) - 1)]         # the rate logged in row N of the CSV (the last row beyond it)
    assert lr1 <= want_lr * (1 + 1e-12) and lr1 >= want_lr * 0.9 ** 3, (lr1, want_lr)     # (ReduceLROnPlateau may have stepped since)
    # `mp predict --save_input_files` (bin/predict.py:90-117): prediction, input image and labels in a sub-folder per volume
    from multiplanarunet_amd.nifti import read_nifti
    from multiplanarunet_amd.data import make_toy_volume
    mp.entry_func(["predict", "--project_dir", str(proj), "--synthetic", "1", "--sum_fusion", "--save_input_files", "--out_format", "nii"])
    sub = proj / "predictions" / "nii_files" / "toy_5000"
    assert sorted(os.listdir(sub)) == ["toy_5000_IMAGE.nii.gz", "toy_5000_LABELS.nii.gz", "toy_5000_PRED.nii.gz"]
    img, lab_true, aff = make_toy_volume(64, 5000)
    im2, aff2, _ = read_nifti(str(sub / "toy_5000_IMAGE.nii.gz"))
    np.testing.assert_array_equal(np.asarray(im2, np.float32), img[..., 0])
    lb2, _, _ = read_nifti(str(sub / "toy_5000_LABELS.nii.gz"), scaled=False)
    np.testing.assert_array_equal(lb2, lab_true)
    np.testing.assert_array_equal(aff2, aff)
    with pytest.raises(OSError):                            # the sub-folder's prediction exists: --overwrite or --continue
        mp.entry_func(["predict", "--project_dir", str(proj), "--synthetic", "1", "--sum_fusion", "--save_input_files", "--out_format", "nii"])


def test_mp_train_two_ranks_data_parallel_then_resume(tmp_path):
    """`mp train --num_GPUs 2` end to end: the entry point re-launches itself as two ranks (sharing GPU 0 over gloo here, the testing
    aid of tests/test_gpu_bench_multi.py), each rank cuts its own half of the batch on its producer stream, gradients are SUM
    all-reduced, rank 0 alone writes checkpoints / CSV; `--continue_training` then resumes on both ranks from rank 0's decision."""
    import subprocess, sys, re
    proj = tmp_path / "proj"
    proj.mkdir()
    (proj / "train_hparams.yaml").write_text(
        "build:\n  model_class_name: UNet\n  n_classes: 3\n  n_channels: 1\n  dim: 64\n  depth: 3\n"
        "  complexity_factor: 0.0625\n  out_activation: softmax\n  seed: 0\n"
        "fit:\n  views: 3\n  noise_sd: 0.1\n  real_space_span: 64.0\n  batch_size: 8\n  n_epochs: 2\n"
        "  optimizer: Adam\n  optimizer_kwargs: {lr: 2.0e-3, decay: 0.0, beta_1: 0.9, beta_2: 0.999, epsilon: 1.0e-8}\n"
        "  loss: SparseCategoricalCrossentropy\n  fg_batch_fraction: 0.5\n  bg_value: 1pct\n  scaler: RobustScaler\n")
    env = dict(os.environ, MPU_SHARE_GPU="1", MPU_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, "-m", "multiplanarunet_amd.cli.mp", "train", "--project_dir", str(proj), "--synthetic", "4",
            "--train_images_per_epoch", "96", "--val_images_per_epoch", "32", "--num_GPUs", "2"]
    r = subprocess.run(base + ["--epochs", "3"], env=env, capture_output=True, text=True, cwd=root, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert r.stdout.count("Saved %s" % (proj / "model" / "model_weights.npz")) == 1      # rank 0 alone writes and reports
    rows = (proj / "logs" / "training.csv").read_text().strip().splitlines()
    head = rows[0].split(",")
    assert [int(x.split(",")[0]) for x in rows[1:]] == [0, 1, 2]
    loss = [float(x.split(",")[head.index("loss")]) for x in rows[1:]]
    assert loss[-1] < loss[0], loss
    ck = [int(re.findall(r"@epoch_(\d+)_", f)[0]) for f in os.listdir(proj / "model") if f.startswith("@epoch")]
    assert ck
    N = max(ck)
    r = subprocess.run(base + ["--epochs", "6", "--continue_training"], env=env, capture_output=True, text=True, cwd=root, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert r.stdout.count("Training continues from") == 1
    rows1 = (proj / "logs" / "training.csv").read_text().strip().splitlines()
    ep1 = [int(x.split(",")[0]) for x in rows1[1:]]
    assert ep1 == [e for e in (0, 1, 2) if e <= N] + list(range(N + 1, 6)), (N, ep1)
    loss1 = float(rows1[-1].split(",")[head.index("loss")])
    assert loss1 < loss[0], (loss, loss1)


def test_mp_train_then_predict_split_bf16(tmp_path):
    """`--dtype bf16x3` through the CLIs (round 6): the graphed pipeline with the split-bf16 step, checkpoints, then a predict in the
    same dtype on the saved weights."""
    from multiplanarunet_amd.cli import mp
    proj = tmp_path / "proj"
    proj.mkdir()
    (proj / "train_hparams.yaml").write_text(
        "build:\n  model_class_name: UNet\n  n_classes: 3\n  n_channels: 1\n  dim: 64\n  depth: 3\n"
        "  complexity_factor: 0.0625\n  out_activation: softmax\n  seed: 0\n"
        "fit:\n  views: 3\n  noise_sd: 0.1\n  real_space_span: 64.0\n  batch_size: 8\n  n_epochs: 2\n"
        "  optimizer: Adam\n  optimizer_kwargs: {lr: 1.0e-3, decay: 0.0, beta_1: 0.9, beta_2: 0.999, epsilon: 1.0e-8}\n"
        "  loss: SparseCategoricalCrossentropy\n  fg_batch_fraction: 0.5\n  bg_value: 1pct\n  scaler: RobustScaler\n")
    mp.entry_func(["train", "--project_dir", str(proj), "--synthetic", "4", "--epochs", "6", "--dtype", "bf16x3",
                   "--train_images_per_epoch", "160", "--no_val"])
    log = (proj / "logs" / "training.csv").read_text().strip().splitlines()
    losses = [float(l.split(",")[1]) for l in log[1:]]
    assert len(losses) == 6 and all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    mp.entry_func(["predict", "--project_dir", str(proj), "--synthetic", "1", "--sum_fusion", "--overwrite", "--dtype", "bf16x3"])
    res = (proj / "predictions" / "csv" / "results.csv").read_text().splitlines()
    assert float(res[1].split(",")[1]) > 0.5, res
