"""
Epoch-end validation and the Keras callbacks that consume it.

  Validation ............ mpunet/callbacks/validation.py:59-306: predict `steps` validation batches, count
                          TP / relevant / selected per class (bincounts), per-class precision, recall, Dice,
                          logs val_dice / val_precision / val_recall = nanmean over classes (background NaN).
                          argmax + counting run in one HIP kernel (mpu_validation_count, csrc/validation.hip);
                          under data parallelism the int64 counts are SUM all-reduced.
                          NOTE the reference passes sel=relevant, rel=selected into _compute_dice
                          (validation.py:211-213), so its "precision" is TP/relevant and its "recall"
                          TP/selected; Dice is symmetric. The logs here carry the same (swapped) meaning.
  ReduceLROnPlateau ..... tf.keras semantics with the YAML's kwargs {patience 2, factor .9, monitor val_dice, max}
                          (min_delta 1e-4, cooldown 0, min_lr 0: Keras defaults)
  EarlyStopping ......... {monitor val_dice, min_delta 0, patience 15, mode max}
  ModelCheckPointClean .. mpunet/callbacks/mcp_clean.py:25-59: save_best_only checkpoints named
                          @epoch_{epoch:02d}_val_dice_{val_dice:.5f}, older ones removed.
"""
import os
import numpy as np
import torch


def count_cm_elements(pred, true, n_classes, counts=None):
    """
    validation.py:115-125 on the GPU: argmax of the class scores `pred` [..., K] (f32, device) fused with the
    per-class TP / relevant / selected counting (mpu_validation_count, csrc/validation.hip). ADDS into
    `counts` (int64 [3, K], device; created when None) and returns it.
    """
    from . import _lib
    K = int(n_classes)
    pred = pred.reshape(-1, K)
    if pred.dtype != torch.float32 or not pred.is_contiguous():
        pred = pred.to(torch.float32).contiguous()
    y = true.reshape(-1)
    if y.dtype != torch.uint8 or y.device != pred.device or not y.is_contiguous():
        y = y.to(device=pred.device, dtype=torch.uint8).contiguous()
    if y.shape[0] != pred.shape[0]:
        raise ValueError("count_cm_elements: %d predictions for %d targets" % (pred.shape[0], y.shape[0]))
    if pred.device.type != "cuda":
        raise _lib.MpuError("count_cm_elements needs a HIP device (no CPU path)")
    if counts is None:
        counts = torch.zeros((3, K), dtype=torch.int64, device=pred.device)
    _lib.call("mpu_validation_count", _lib.ptr(pred), _lib.ptr(y), pred.shape[0], K, _lib.ptr(counts), _lib.stream_ptr())
    return counts


def dice_from_counts(counts, smooth=1.0, ignore_zero=False):
    """mpunet/evaluate/metrics.py:26-52 (dice_all, n_classes given, skip_if_no_y=False) from the integer counts of
    count_cm_elements ([3, K]: TP, relevant = #(y == c), selected = #(p == c)): (smooth + 2 TP) / (smooth + relevant +
    selected) in float64 stored as float32, NaN where the class is in neither array. Integer work: the counts are exact, so
    the result IS the reference's."""
    c = counts.cpu().numpy() if torch.is_tensor(counts) else np.asarray(counts)
    tp, rel, sel = (c[i].astype(np.float64) for i in range(3))
    out = np.full(tp.shape, np.nan, dtype=np.float32)
    m = (rel > 0) | (sel > 0)
    out[m] = ((smooth + 2 * tp[m]) / (smooth + rel[m] + sel[m])).astype(np.float32)
    return out[1:] if ignore_zero else out


def compute_dice(tp, rel, sel):
    """validation.py:59-89 (_compute_dice): zeros where a denominator is zero. numpy float32 out."""
    tp, rel, sel = (np.asarray(a, dtype=np.float64) for a in (tp, rel, sel))
    precisions = np.zeros(tp.shape, np.float32)
    recalls = np.zeros_like(precisions)
    dices = np.zeros_like(precisions)
    sm, rm = sel > 0, rel > 0
    precisions[sm] = tp[sm] / sel[sm]
    recalls[rm] = tp[rm] / rel[rm]
    intrs, union = 2 * precisions * recalls, precisions + recalls
    dm = union > 0
    dices[dm] = intrs[dm] / union[dm]
    return precisions, recalls, dices


class Validation:
    def __init__(self, sampler, steps, n_classes, ignore_class_zero=True, logger=None, verbose=True):
        self.sampler, self.steps, self.n_classes = sampler, int(steps), int(n_classes)
        self.ignore_bg = ignore_class_zero
        self.logger = logger or print
        self.verbose = verbose

    def evaluate(self, model):
        K = self.n_classes
        cnt = torch.zeros((3, K), dtype=torch.int64, device=model.device)      # TP, relevant, selected
        for _ in range(self.steps):
            x, y, _w = self.sampler()
            pred = model.predict_on_batch(x)
            count_cm_elements(pred, y, K, counts=cnt)
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            torch.distributed.all_reduce(cnt)
        TP, REL, SEL = cnt[0], cnt[1], cnt[2]
        # as the reference: sel=relevant, rel=selected
        precisions, recalls, dices = compute_dice(TP.cpu().numpy(), rel=SEL.cpu().numpy(), sel=REL.cpu().numpy())
        if self.ignore_bg:
            precisions[0] = recalls[0] = dices[0] = np.nan
        return {"dice": dices, "recall": recalls, "precision": precisions}

    def on_epoch_end(self, model, epoch, logs):
        cw = self.evaluate(model)
        with np.errstate(all="ignore"):
            for name, values in cw.items():
                logs["val_" + name] = float(np.nanmean(values)) if np.any(~np.isnan(values)) else float("nan")
        if self.verbose:
            rows = ["Validation Results for epoch %d" % epoch,
                    "        " + "  ".join("%9s" % c for c in ["mean"] + ["cls %d" % i for i in range(self.n_classes)])]
            for name, values in cw.items():
                vals = [logs["val_" + name]] + list(values)
                rows.append("%-9s" % name + "  ".join("%9s" % ("-" if np.isnan(v) else "%.4f" % v) for v in vals))
            self.logger("\n".join(rows))
        return cw


class ReduceLROnPlateau:
    def __init__(self, monitor="val_dice", factor=0.9, patience=2, mode="max", min_delta=1e-4, cooldown=0, min_lr=0.0,
                 verbose=1, logger=None):
        if factor >= 1.0:
            raise ValueError("ReduceLROnPlateau does not support a factor >= 1.0.")
        self.monitor, self.factor, self.patience, self.mode = monitor, factor, patience, mode
        self.min_delta, self.cooldown, self.min_lr = min_delta, cooldown, min_lr
        self.logger = logger or print
        self.verbose = verbose
        self.best = -np.inf if mode == "max" else np.inf
        self.wait = 0
        self.cooldown_counter = 0

    def _better(self, a, b):
        return a > b + self.min_delta if self.mode == "max" else a < b - self.min_delta

    def on_epoch_end(self, model, epoch, logs):
        cur = logs.get(self.monitor)
        logs["lr"] = model.optimizer_kwargs["lr"]
        if cur is None or np.isnan(cur):
            return
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.wait = 0
        if self._better(cur, self.best):
            self.best, self.wait = cur, 0
        elif self.cooldown_counter <= 0:
            self.wait += 1
            if self.wait >= self.patience:
                old = float(model.optimizer_kwargs["lr"])
                if old > self.min_lr:
                    new = max(old * self.factor, self.min_lr)
                    model.optimizer_kwargs["lr"] = new
                    if self.verbose:
                        self.logger("Epoch %05d: ReduceLROnPlateau reducing learning rate to %s." % (epoch + 1, new))
                    self.cooldown_counter = self.cooldown
                    self.wait = 0


class EarlyStopping:
    def __init__(self, monitor="val_dice", min_delta=0, patience=15, mode="max", verbose=1, logger=None):
        self.monitor, self.min_delta, self.patience, self.mode = monitor, abs(min_delta), patience, mode
        self.logger = logger or print
        self.verbose = verbose
        self.best = -np.inf if mode == "max" else np.inf
        self.wait = 0
        self.stopped_epoch = 0

    def on_epoch_end(self, model, epoch, logs):
        cur = logs.get(self.monitor)
        if cur is None:
            return
        better = cur - self.min_delta > self.best if self.mode == "max" else cur + self.min_delta < self.best
        if better:
            self.best, self.wait = cur, 0
        else:
            self.wait += 1
            if self.wait >= self.patience:
                self.stopped_epoch = epoch
                model.stop_training = True
                if self.verbose:
                    self.logger("Epoch %05d: early stopping" % (epoch + 1))


class ModelCheckPointClean:
    """save_best_only + save_weights_only; the previously saved file is removed (mcp_clean.py:25-59)."""

    def __init__(self, filepath, monitor="val_dice", mode="max", verbose=1, logger=None):
        self.filepath, self.monitor, self.mode = filepath, monitor, mode
        self.logger = logger or print
        self.verbose = verbose
        self.best = -np.inf if mode == "max" else np.inf
        self.last_file = None

    def on_epoch_end(self, model, epoch, logs):
        cur = logs.get(self.monitor)
        if cur is None or np.isnan(cur):
            return
        if (cur > self.best) if self.mode == "max" else (cur < self.best):
            path = self.filepath.format(epoch=epoch + 1, **logs)
            if self.verbose:
                self.logger("Epoch %05d: %s improved from %.5f to %.5f, saving model to %s"
                            % (epoch + 1, self.monitor, self.best, cur, path))
            self.best = cur
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            model.save_weights(path)
            if self.last_file and self.last_file != path and os.path.exists(self.last_file):
                os.remove(self.last_file)
            self.last_file = path


class CSVLogger:
    """tf.keras.callbacks.CSVLogger with the YAML's kwargs {filename, separator, append}: a header `epoch,<sorted log
    keys>` on the first write (skipped when appending to a non-empty file), one row per epoch, keys missing from a later
    epoch's logs written as NA."""

    def __init__(self, filename, separator=",", append=False):
        self.filename, self.sep, self.append = filename, separator, append
        self.keys = None
        self._started = False

    def on_epoch_end(self, model, epoch, logs):
        os.makedirs(os.path.dirname(os.path.abspath(self.filename)), exist_ok=True)
        if not self._started:
            self._started = True
            self.keys = sorted(logs.keys())
            has_rows = self.append and os.path.exists(self.filename) and os.path.getsize(self.filename) > 0
            if not has_rows:
                with open(self.filename, "w") as f:
                    f.write(self.sep.join(["epoch"] + self.keys) + "\n")
        with open(self.filename, "a") as f:
            vals = ["NA" if logs.get(k) is None else ("%s" % logs[k]) for k in self.keys]
            f.write(self.sep.join([str(epoch)] + vals) + "\n")


class DelayedCallback:
    """mpunet/callbacks/callbacks.py:88-115: the wrapped callback stays inactive until epoch `start_from`."""

    def __init__(self, callback, start_from=0, logger=None):
        self.callback, self.start_from, self.logger = callback, start_from, logger or print

    def __getattr__(self, item):
        return getattr(self.callback, item)

    def on_epoch_end(self, model, epoch, logs):
        if epoch >= self.start_from - 1:
            self.callback.on_epoch_end(model, epoch, logs)
        else:
            self.logger("[%s] Not active at epoch %i - will be at %i" % (self.callback.__class__.__name__, epoch + 1,
                                                                        self.start_from))
