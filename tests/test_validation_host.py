"""Host logic of the validation callback and the Keras-semantics callbacks (CPU: torch ops on CPU tensors)."""
import os
import numpy as np
import torch

from multiplanarunet_amd import validation as V


def test_counts_match_numpy_bincount_and_dice_formula():
    rng = np.random.RandomState(0)
    K = 4
    y = rng.randint(0, K, 5000); p = rng.randint(0, K, 5000)
    p[:2000] = y[:2000]
    tps, rel, sel = V.count_cm_elements(torch.tensor(p), torch.tensor(y), K)
    np.testing.assert_array_equal(tps.numpy(), np.bincount(np.where(y == p, y, K), minlength=K + 1)[:-1])
    np.testing.assert_array_equal(rel.numpy(), np.bincount(y, minlength=K))
    np.testing.assert_array_equal(sel.numpy(), np.bincount(p, minlength=K))
    pr, rc, dc = V.compute_dice(tps.numpy(), rel.numpy(), sel.numpy())
    np.testing.assert_allclose(pr, tps.numpy() / sel.numpy(), rtol=1e-6)
    np.testing.assert_allclose(rc, tps.numpy() / rel.numpy(), rtol=1e-6)
    np.testing.assert_allclose(dc, 2 * tps.numpy() / (rel.numpy() + sel.numpy()), rtol=1e-5)   # 2PR/(P+R) == 2TP/(rel+sel)


def test_compute_dice_zero_denominators_give_zero():
    pr, rc, dc = V.compute_dice(np.array([0, 3, 0]), np.array([0, 4, 5]), np.array([0, 3, 0]))
    assert pr.tolist() == [0.0, 1.0, 0.0] and rc[0] == 0.0 and rc[2] == 0.0 and dc[0] == 0.0 and dc[2] == 0.0


class _M:
    def __init__(self):
        self.optimizer_kwargs = {"lr": 1.0}
        self.stop_training = False
        self.saved = []
        self.device = torch.device("cpu")

    def save_weights(self, path):
        open(path, "w").write("w"); self.saved.append(path)


def test_reduce_lr_on_plateau_keras_semantics():
    m = _M()
    cb = V.ReduceLROnPlateau(patience=2, factor=0.9, verbose=0)
    seq = [0.5, 0.6, 0.6, 0.60005, 0.7, 0.69, 0.68, 0.67]
    lrs = []
    for ep, v in enumerate(seq):
        cb.on_epoch_end(m, ep, {"val_dice": v}); lrs.append(m.optimizer_kwargs["lr"])
    # epochs 2,3 do not beat 0.6 by min_delta 1e-4 -> reduce at epoch 3; 0.7 resets; 0.69, 0.68 -> reduce at epoch 6
    np.testing.assert_allclose(lrs, [1, 1, 1, 0.9, 0.9, 0.9, 0.81, 0.81])


def test_early_stopping_and_checkpoint_clean(tmp_path):
    m = _M()
    es = V.EarlyStopping(patience=3, verbose=0)
    ck = V.ModelCheckPointClean(str(tmp_path / "model" / "@epoch_{epoch:02d}_val_dice_{val_dice:.5f}.npz"), verbose=0)
    vals = [0.3, 0.5, 0.4, 0.45, 0.5, 0.2]
    for ep, v in enumerate(vals):
        logs = {"val_dice": v}
        ck.on_epoch_end(m, ep, logs); es.on_epoch_end(m, ep, logs)
        if m.stop_training:
            break
    assert ep == 4 and es.stopped_epoch == 4                      # 3 epochs without beating 0.5 (equal is not better)
    assert os.listdir(tmp_path / "model") == ["@epoch_02_val_dice_0.50000.npz"]


def test_validation_logs_swapped_names_and_background_nan():
    K = 3
    class S:
        def __init__(self): self.i = 0
        def __call__(self):
            rng = np.random.RandomState(self.i); self.i += 1
            y = torch.tensor(rng.randint(0, K, (2, 16, 1)).astype(np.uint8))
            return y.float(), y, None
    class Mdl(_M):
        def predict_on_batch(self, x):                             # predicts the label, except class 2 -> 1 half the time
            lab = x.long().reshape(-1)
            flip = (torch.arange(lab.numel()) % 2 == 0) & (lab == 2)
            lab = torch.where(flip, torch.ones_like(lab), lab)
            return torch.nn.functional.one_hot(lab, K).float().reshape(2, 16, K)
    logs = {}
    cw = V.Validation(S(), steps=3, n_classes=K, verbose=False).on_epoch_end(Mdl(), 0, logs)
    assert np.isnan(cw["dice"][0]) and set(logs) == {"val_dice", "val_precision", "val_recall"}
    # class 2 is never over-predicted: TP/selected = 1, which the reference logs under "recall" (swapped names)
    assert cw["recall"][2] == 1.0 and cw["precision"][2] < 1.0
