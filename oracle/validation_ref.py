"""
ORACLE -- TEST INFRASTRUCTURE ONLY. Never imported by the product path.

NumPy restatement of the epoch-end validation counting of the reference
(mpunet/callbacks/validation.py): PINNED by tests/golden/validation_golden.npz, which holds outputs of the
reference's own unmodified `_count_cm_elements_from_queue` / `_compute_dice` (oracle/gen_golden_validation.py).
"""
import numpy as np


def count_cm_elements(pred, true, n_classes):
    """validation.py:115-125: argmax over the class axis, then TP / relevant / selected per class via bincounts
    (a mismatch is binned into the dummy class n_classes, which is dropped)."""
    p = np.asarray(pred).argmax(-1).ravel()
    y = np.asarray(true).ravel()
    tps = np.bincount(np.where(y == p, y, n_classes), minlength=n_classes + 1)[:-1]
    rel = np.bincount(y, minlength=n_classes)
    sel = np.bincount(p, minlength=n_classes)
    return tps.astype(np.uint64), rel.astype(np.uint64), sel.astype(np.uint64)


def compute_dice(tp, rel, sel):
    """validation.py:59-89 (_compute_dice): precision = tp/sel, recall = tp/rel, dice = 2PR/(P+R); zeros where a
    denominator is zero; float32 results."""
    sel_mask = sel > 0
    rel_mask = rel > 0
    precisions = np.zeros(shape=tp.shape, dtype=np.float32)
    recalls = np.zeros_like(precisions)
    dices = np.zeros_like(precisions)
    precisions[sel_mask] = tp[sel_mask] / sel[sel_mask]
    recalls[rel_mask] = tp[rel_mask] / rel[rel_mask]
    intrs = (2 * precisions * recalls)
    union = (precisions + recalls)
    dice_mask = union > 0
    dices[dice_mask] = intrs[dice_mask] / union[dice_mask]
    return precisions, recalls, dices


def class_wise_metrics(tp, relevant, selected, ignore_bg=True):
    """validation.py:208-221 (evalaute): NOTE the swapped keywords -- sel=relevant, rel=selected -- so the
    reference's "precision" is TP/relevant and its "recall" TP/selected; background set to NaN."""
    precisions, recalls, dices = compute_dice(tp=tp, sel=relevant, rel=selected)
    if ignore_bg:
        precisions[0] = recalls[0] = dices[0] = np.nan
    return {"dice": dices, "recall": recalls, "precision": precisions}
