"""Host logic of the validation callback and the Keras-semantics callbacks (CPU: torch ops on CPU tensors)."""
import os
import numpy as np
import torch

from multiplanarunet_amd import validation as V


def _golden():
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "validation_golden.npz")) as z:
        return {k: z[k] for k in z.files}


def test_oracle_counts_and_metrics_match_reference_goldens():
    """G8: the NumPy restatement (oracle/validation_ref.py) against outputs of the reference's own
    _count_cm_elements_from_queue / _compute_dice (mpunet/callbacks/validation.py:59-131)."""
    from oracle import validation_ref as R
    g = _golden()
    for ci, (K, steps, B, npx) in enumerate(g["g8_cases"]):
        tp = np.zeros(K, np.uint64); rel = np.zeros(K, np.uint64); sel = np.zeros(K, np.uint64)
        for s in range(steps):
            a, b, c = R.count_cm_elements(g["g8_pred_%d" % ci][s], g["g8_true_%d" % ci][s], int(K))
            tp += a; rel += b; sel += c
        np.testing.assert_array_equal(tp, g["g8_tp_%d" % ci])
        np.testing.assert_array_equal(rel, g["g8_rel_%d" % ci])
        np.testing.assert_array_equal(sel, g["g8_sel_%d" % ci])
        cw = R.class_wise_metrics(tp, rel, sel, ignore_bg=False)
        for name in ("precision", "recall", "dice"):
            np.testing.assert_array_equal(cw[name], g["g8_%s_%d" % (name, ci)])


def test_product_compute_dice_matches_reference_goldens_incl_swapped_naming():
    """The product's _compute_dice and its (swapped, as the reference) precision / recall naming."""
    g = _golden()
    for ci in range(len(g["g8_cases"])):
        tp, rel, sel = (g["g8_%s_%d" % (n, ci)] for n in ("tp", "rel", "sel"))
        # evalaute(): _compute_dice(tp=TPs, sel=relevant, rel=selected)  (validation.py:211-213)
        pr, rc, dc = V.compute_dice(tp, rel=sel, sel=rel)
        np.testing.assert_array_equal(pr, g["g8_precision_%d" % ci])
        np.testing.assert_array_equal(rc, g["g8_recall_%d" % ci])
        np.testing.assert_array_equal(dc, g["g8_dice_%d" % ci])
        assert pr.dtype == np.float32 and dc.dtype == np.float32
    # "precision" of the reference is TP / relevant
    tp, rel, sel = g["g8_tp_0"].astype(float), g["g8_rel_0"].astype(float), g["g8_sel_0"].astype(float)
    np.testing.assert_allclose(g["g8_precision_0"], tp / rel, rtol=1e-6)
    np.testing.assert_allclose(g["g8_recall_0"], tp / sel, rtol=1e-6)


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


def test_dice_from_counts_is_the_references_dice_all():
    """Per-view evaluation (mpunet/bin/predict.py:236-275) computes dice_all(ignore_zero=False) of two label arrays; the GPU path
    counts TP / relevant / selected per class and forms the Dice from the integers: identical float32 values, NaN for a class in
    neither array (oracle/geometry.dice_all is pinned on the reference's own outputs, golden G6)."""
    from multiplanarunet_amd.validation import dice_from_counts
    from oracle import geometry as G
    rng = np.random.RandomState(4)
    for K, holes in ((5, (3,)), (3, ()), (4, (0, 2))):
        y = rng.randint(0, K, 5000).astype(np.uint8)
        p = rng.randint(0, K, 5000).astype(np.uint8)
        for h in holes:                                   # a class absent from both arrays -> NaN
            y[y == h] = (h + 1) % K
            p[p == h] = (h + 1) % K
        counts = np.stack([np.bincount(y[y == p], minlength=K), np.bincount(y, minlength=K), np.bincount(p, minlength=K)]).astype(np.int64)
        for ign in (False, True):
            want = G.dice_all(y, p, n_classes=K, ignore_zero=ign)
            got = dice_from_counts(counts, ignore_zero=ign)
            assert got.dtype == np.float32
            np.testing.assert_array_equal(got, want)
