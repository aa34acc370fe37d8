"""
GPU parity of the validation counting kernel (mpu_validation_count through the C ABI and through
Validation.evaluate) against the reference goldens (G8) and the NumPy oracle. Integer work: exact.
"""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _golden():
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "validation_golden.npz")) as z:
        return {k: z[k] for k in z.files}


def test_counts_vs_reference_goldens():
    from multiplanarunet_amd import validation as V
    g = _golden()
    for ci, (K, steps, B, npx) in enumerate(g["g8_cases"]):
        cnt = None
        for s in range(steps):
            cnt = V.count_cm_elements(torch.tensor(g["g8_pred_%d" % ci][s], device="cuda"),
                                      torch.tensor(g["g8_true_%d" % ci][s], device="cuda"), int(K), counts=cnt)
        c = cnt.cpu().numpy()
        np.testing.assert_array_equal(c[0], g["g8_tp_%d" % ci].astype(np.int64))
        np.testing.assert_array_equal(c[1], g["g8_rel_%d" % ci].astype(np.int64))
        np.testing.assert_array_equal(c[2], g["g8_sel_%d" % ci].astype(np.int64))


@pytest.mark.parametrize("K", (1, 2, 3, 5, 8, 16))
def test_counts_vs_oracle_large_random_with_ties_and_nans(K):
    from multiplanarunet_amd import validation as V
    from oracle import validation_ref as R
    rng = np.random.RandomState(K)
    n = 16 * 128 * 128 + 37                                   # a batch of configs[1] slices + a ragged tail
    p = rng.rand(n, K).astype(np.float32)
    p[::7] = np.round(p[::7], 1)                              # exact ties: first maximum wins
    if K > 1:
        p[5::1001, rng.randint(0, K)] = np.nan                # np.argmax treats NaN as the maximum
    y = rng.randint(0, K, n).astype(np.uint8)
    cnt = V.count_cm_elements(torch.tensor(p, device="cuda"), torch.tensor(y, device="cuda"), K)
    cnt = V.count_cm_elements(torch.tensor(p, device="cuda"), torch.tensor(y, device="cuda"), K, counts=cnt)   # accumulates
    tp, rel, sel = R.count_cm_elements(p, y, K)
    np.testing.assert_array_equal(cnt.cpu().numpy(), 2 * np.stack([tp, rel, sel]).astype(np.int64))


def test_validation_evaluate_matches_oracle_metrics():
    """Validation.evaluate (sampler -> model.predict_on_batch -> counting kernel -> class-wise metrics) with a
    stand-in model whose scores are known, against the oracle incl. the reference's swapped precision/recall."""
    from multiplanarunet_amd import validation as V
    from oracle import validation_ref as R
    K, B, npx, steps = 4, 3, 500, 3
    rng = np.random.RandomState(0)
    batches = [(rng.rand(B, npx, K).astype(np.float32), rng.randint(0, K, (B, npx, 1)).astype(np.uint8)) for _ in range(steps)]

    class Model:
        device = torch.device("cuda")

        def __init__(self):
            self.i = 0

        def predict_on_batch(self, x):
            p = torch.tensor(batches[self.i][0], device="cuda"); self.i += 1
            return p

    it = iter(batches)
    sampler = lambda: (None, torch.tensor(next(it)[1], device="cuda"), None)
    cw = V.Validation(sampler, steps, K, logger=lambda *a: None, verbose=False).evaluate(Model())
    tp = np.zeros(K, np.uint64); rel = np.zeros(K, np.uint64); sel = np.zeros(K, np.uint64)
    for p, y in batches:
        a, b, c = R.count_cm_elements(p, y, K)
        tp += a; rel += b; sel += c
    ref = R.class_wise_metrics(tp, rel, sel, ignore_bg=True)
    for name in ("dice", "precision", "recall"):
        np.testing.assert_array_equal(cw[name], ref[name])
    assert np.isnan(cw["dice"][0])


def test_validation_logs_swapped_names_and_background_nan():
    from multiplanarunet_amd import validation as V
    K = 3

    class S:
        def __init__(self): self.i = 0
        def __call__(self):
            rng = np.random.RandomState(self.i); self.i += 1
            y = torch.tensor(rng.randint(0, K, (2, 16, 1)).astype(np.uint8), device="cuda")
            return y.float(), y, None

    class Mdl:
        device = torch.device("cuda")
        def predict_on_batch(self, x):                             # predicts the label, except class 2 -> 1 half the time
            lab = x.long().reshape(-1)
            flip = (torch.arange(lab.numel(), device=lab.device) % 2 == 0) & (lab == 2)
            lab = torch.where(flip, torch.ones_like(lab), lab)
            return torch.nn.functional.one_hot(lab, K).float().reshape(2, 16, K)
    logs = {}
    cw = V.Validation(S(), steps=3, n_classes=K, verbose=False).on_epoch_end(Mdl(), 0, logs)
    assert np.isnan(cw["dice"][0]) and set(logs) == {"val_dice", "val_precision", "val_recall"}
    # class 2 is never over-predicted: TP/selected = 1, which the reference logs under "recall" (swapped names)
    assert cw["recall"][2] == 1.0 and cw["precision"][2] < 1.0
