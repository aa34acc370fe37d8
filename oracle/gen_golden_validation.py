"""
TEST INFRASTRUCTURE ONLY -- generates tests/golden/validation_golden.npz by running the reference's own,
unmodified NumPy code (mpunet/callbacks/validation.py:59-89 `_compute_dice`, :91-131
`_count_cm_elements_from_queue`) through oracle/ref_shim.py. Run by hand in the build container:

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_validation.py

The file holds only data: seeded (pred, true) batches and the reference's counts / metrics (G8).
"""
import os
import sys
from queue import Queue
from threading import Lock
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
from mpunet.callbacks.validation import Validation  # noqa: E402


def main():
    rng = np.random.RandomState(8)
    out = {}
    cases = []
    for ci, (K, steps, B, npx) in enumerate(((3, 3, 2, 257), (5, 2, 3, 100), (2, 1, 1, 64), (8, 2, 2, 333))):
        preds, trues = [], []
        for s in range(steps):
            p = rng.rand(B, npx, K).astype(np.float32)
            if s == 0:
                p[0, :16] = p[0, :16].round(1)           # exact ties: argmax takes the first maximum
                p[0, 16:24] = 0.25
            y = rng.randint(0, K, size=(B, npx, 1)).astype(np.uint8)
            if K == 5:
                y[y == 4] = 0                            # a class absent from the targets
                p[..., 3] = -1.0                         # ... and one never selected
            preds.append(p); trues.append(y)
        q = Queue(maxsize=steps)
        TPs = {"task": np.zeros(K, np.uint64)}
        rel = {"task": np.zeros(K, np.uint64)}
        sel = {"task": np.zeros(K, np.uint64)}
        for p, y in zip(preds, trues):
            q.put([[p], [y]])
        Validation._count_cm_elements_from_queue(q, steps, TPs, rel, sel, ["task"], [K], Lock())
        out["g8_pred_%d" % ci] = np.stack(preds)
        out["g8_true_%d" % ci] = np.stack(trues)
        out["g8_tp_%d" % ci], out["g8_rel_%d" % ci], out["g8_sel_%d" % ci] = TPs["task"], rel["task"], sel["task"]
        # evalaute() passes sel=relevant, rel=selected (validation.py:211-213): keep that call
        pr, rc, dc = Validation._compute_dice(tp=TPs["task"], sel=rel["task"], rel=sel["task"])
        out["g8_precision_%d" % ci], out["g8_recall_%d" % ci], out["g8_dice_%d" % ci] = pr, rc, dc
        cases.append((K, steps, B, npx))
    out["g8_cases"] = np.array(cases)
    dst = os.path.join(HERE, "..", "tests", "golden", "validation_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", os.path.abspath(dst), "%.1f KB" % (os.path.getsize(dst) / 1e3), len(out), "arrays")


if __name__ == "__main__":
    main()
