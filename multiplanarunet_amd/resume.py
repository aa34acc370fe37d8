"""
`mp train --continue_training`: where a training session resumes (mpunet/models/model_init.py:23-47 with the helpers of
mpunet/utils/utils.py:113-172). Same decisions as the reference, restated without pandas:

  * the newest `model/@epoch_<N>_*` checkpoint is loaded by layer name, else `model/model_weights.h5` (here also the `.npz`
    mirror) with epoch 0, else nothing;
  * with a checkpoint of epoch N > 0 `logs/training.csv` is cut back: trailing runs before the last `epoch == 0` row are
    dropped and rows [0, N] kept; with the generic weights file (epoch 0) the epoch is the last one the CSV holds;
  * training continues at `init_epoch = N + 1` (the reference passes that as Keras' 0-based `initial_epoch`; the checkpoint
    number N is Keras' 1-based epoch, so row N of the CSV -- the epoch after the checkpoint's -- is kept and not repeated:
    reproduced as it is);
  * the learning rate is the one logged in row N of the CSV's `lr` / `LR` / `learning_rate` / `LearningRate` column.

Only the weights come back (a Keras weights file carries no optimizer state): Adam restarts from zero moments, as there.
"""
import csv
import glob
import os
import re

LR_NAMES = ("lr", "LR", "learning_rate", "LearningRate")
WEIGHT_EXTS = (".h5", ".hdf5", ".npz")


def get_last_model(model_dir):
    """utils.py:113-131 -> (path, epoch) | (generic weights path, 0) | (None, None)."""
    models = [m for m in glob.glob(os.path.join(model_dir, "@epoch*")) if m.endswith(WEIGHT_EXTS)]
    found = []
    for m in models:
        hit = re.findall(r"@epoch_(\d+)_", m)
        if hit:
            found.append((int(hit[0]), m))
    if found:
        top = max(e for e, _ in found)
        # a checkpoint may exist as .h5 and as its .npz mirror: either holds the same tensors; prefer the Keras file
        best = sorted((m for e, m in found if e == top), key=lambda p: (not p.endswith((".h5", ".hdf5")), p))[0]
        return os.path.abspath(best), top
    for name in ("model_weights.h5", "model_weights.npz"):
        generic = os.path.join(model_dir, name)
        if os.path.exists(generic):
            return os.path.abspath(generic), 0          # "epoch 0 as we dont know where else to start" (utils.py:126-128)
    return None, None


def _read_csv(path):
    """(header, rows) of a CSVLogger file; rows are lists of strings. A missing / empty file gives (None, [])."""
    if not os.path.exists(path):
        return None, []
    with open(path, newline="") as f:
        rows = [r for r in csv.reader(f) if r]
    if not rows:
        return None, []
    return rows[0], rows[1:]


def get_last_epoch(csv_file):
    """utils.py:166-172: the `epoch` value of the CSV's last row (0 without a file)."""
    header, rows = _read_csv(csv_file)
    if header is None or not rows or "epoch" not in header:
        return 0
    return int(float(rows[-1][header.index("epoch")]))


def clear_csv_after_epoch(epoch, csv_file):
    """utils.py:145-163: drop runs before the last `epoch == 0` row, keep rows [0, epoch]; an empty file is removed."""
    if not os.path.exists(csv_file):
        return
    header, rows = _read_csv(csv_file)
    if header is None:
        os.remove(csv_file)                              # pandas' EmptyDataError branch
        return
    if "epoch" in header:
        k = header.index("epoch")
        zeros = [i for i, r in enumerate(rows) if int(float(r[k])) == 0]
        if zeros:
            rows = rows[zeros[-1]:]
    rows = rows[:int(epoch) + 1]
    with open(csv_file, "w", newline="") as f:
        w = csv.writer(f, lineterminator="\n")
        w.writerow(header)
        w.writerows(rows)


def get_lr_at_epoch(epoch, log_dir, logger=print):
    """utils.py:134-148 -> (lr, column name) | (None, None). Row `epoch` of the CSV (0-based row number)."""
    path = os.path.join(log_dir, "training.csv")
    header, rows = _read_csv(path)
    if header is None:
        logger("No training.csv file found at %s. Continuing with default learning rate found in parameter file." % log_dir)
        return None, None
    name = next((n for n in LR_NAMES if n in header), None)
    if name is None:
        return None, None
    if not 0 <= int(epoch) < len(rows):
        # the reference indexes df[name][epoch] and dies with a KeyError here; the last logged rate is the useful answer
        if not rows:
            return None, None
        logger("[OBS] training.csv holds %d rows, epoch %d asked for: using the last logged learning rate" % (len(rows), epoch))
        epoch = len(rows) - 1
    return float(rows[int(epoch)][header.index(name)]), name


def resume_state(project_dir, logger=print):
    """The decisions of model_initializer(continue_training=True) without the model: dict(model_path, epoch, init_epoch,
    lr, lr_name). Rank 0 calls it (it rewrites the CSV) and broadcasts the result."""
    model_path, epoch = get_last_model(os.path.join(project_dir, "model"))
    csv_path = os.path.join(project_dir, "logs", "training.csv")
    if epoch == 0:
        epoch = get_last_epoch(csv_path)
    else:
        if epoch is None:
            epoch = 0
        clear_csv_after_epoch(epoch, csv_path)
    lr, name = get_lr_at_epoch(epoch, os.path.join(project_dir, "logs"), logger)
    return dict(model_path=model_path, epoch=int(epoch), init_epoch=int(epoch) + 1, lr=lr, lr_name=name)


def apply_resume(model, hparams, state, logger=print):
    """Load the weights and set fit.init_epoch / the optimizer's learning rate (model_init.py:29-51)."""
    if state["model_path"]:
        model.load_weights(state["model_path"], by_name=True)
    hparams["fit"]["init_epoch"] = state["init_epoch"]
    if state["lr"]:
        kw = hparams["fit"].setdefault("optimizer_kwargs", {})
        # one learning-rate key only: the CSV column is named "lr" in this build, a project YAML may spell it
        # `learning_rate` (UNet.compile lets that alias win) -- the resumed rate replaces whichever spelling is there
        for alias in ("lr", "learning_rate", "LR", "LearningRate"):
            kw.pop(alias, None)
        kw[state["lr_name"]] = state["lr"]
    logger("[NOTICE] Training continues from:\nModel: %s\nEpoch: %i\nLR:    %s"
           % (os.path.split(state["model_path"])[-1] if state["model_path"] else "<No model found>", state["epoch"], state["lr"]))
    return model
