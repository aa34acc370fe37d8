"""
TEST INFRASTRUCTURE ONLY -- generates tests/golden/resume_golden.json by running the reference's own, unmodified
helpers behind `mp train --continue_training` (mpunet/utils/utils.py:113-172 `get_last_model`, `get_lr_at_epoch`,
`clear_csv_after_epoch`, `get_last_epoch`, and the decision sequence of mpunet/models/model_init.py:23-47) on small
project folders built in a temporary directory. Run by hand in the build container:

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_resume.py

The file holds only data: per scenario the model-folder file names and the training.csv text going in, and what the
reference returned / left on disk (G9).
"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
from mpunet.utils.utils import get_last_model, get_lr_at_epoch, clear_csv_after_epoch, get_last_epoch  # noqa: E402


def csv_text(rows, header=("epoch", "loss", "lr", "val_dice")):
    return "\n".join([",".join(header)] + [",".join(str(v) for v in r) for r in rows]) + "\n"


def run_rows(n, lr0=5e-5, drop_at=None, start=0):
    rows = []
    for e in range(start, n):
        lr = lr0 * (0.9 if drop_at is not None and e >= drop_at else 1.0)
        rows.append((e, round(1.0 / (e + 1), 6), lr, round(0.5 + 0.01 * e, 5)))
    return rows


SCENARIOS = [
    # name, model files, csv text (None = no file)
    ("checkpoint_mid_run", ["@epoch_05_val_dice_0.61234.h5"], csv_text(run_rows(9, drop_at=4))),
    ("two_checkpoints_takes_newest", ["@epoch_03_val_dice_0.5.h5", "@epoch_12_val_dice_0.7.h5", "model_weights.h5"],
     csv_text(run_rows(20, drop_at=10))),
    ("trailing_runs_are_dropped", ["@epoch_02_val_dice_0.55.h5"], csv_text(run_rows(6) + run_rows(4, lr0=1e-4))),
    ("generic_weights_only", ["model_weights.h5"], csv_text(run_rows(7, drop_at=5))),
    ("generic_weights_no_csv", ["model_weights.h5"], None),
    ("nothing_found", [], csv_text(run_rows(3))),
    ("nothing_found_no_csv", [], None),
    ("lr_column_named_learning_rate", ["@epoch_04_val_dice_0.6.h5"],
     csv_text(run_rows(8, drop_at=3), header=("epoch", "loss", "learning_rate", "val_dice"))),
    ("no_lr_column", ["@epoch_02_val_dice_0.6.h5"],
     csv_text([(e, 1.0 / (e + 1)) for e in range(5)], header=("epoch", "loss"))),
    ("empty_csv_file", ["@epoch_02_val_dice_0.6.h5"], ""),
    ("epoch_ten_vs_nine_numeric_order", ["@epoch_9_val_dice_0.6.h5", "@epoch_10_val_dice_0.5.h5"], csv_text(run_rows(14))),
]


def main():
    out = []
    for name, files, text in SCENARIOS:
        with tempfile.TemporaryDirectory() as d:
            os.makedirs(os.path.join(d, "model")); os.makedirs(os.path.join(d, "logs"))
            for f in files:
                open(os.path.join(d, "model", f), "w").close()
            csv_path = os.path.join(d, "logs", "training.csv")
            if text is not None:
                with open(csv_path, "w") as f:
                    f.write(text)
            rec = {"name": name, "model_files": files, "csv_in": text}
            # the decision sequence of model_init.py:28-47
            try:
                model_path, epoch = get_last_model(os.path.join(d, "model"))
                rec["model_name"] = os.path.split(model_path)[-1] if model_path else None
                rec["last_model_epoch"] = epoch
                if epoch == 0:
                    epoch = get_last_epoch(csv_path)
                else:
                    if epoch is None:
                        epoch = 0
                    clear_csv_after_epoch(epoch, csv_path)
                rec["epoch"] = int(epoch)
                rec["init_epoch"] = int(epoch) + 1
                rec["csv_exists_after"] = os.path.exists(csv_path)
                if rec["csv_exists_after"]:
                    import pandas as pd
                    try:
                        df = pd.read_csv(csv_path)
                        rec["csv_after_columns"] = list(df.columns)
                        rec["csv_after_rows"] = [[float(v) for v in r] for r in df.to_numpy().tolist()]
                    except pd.errors.EmptyDataError:
                        rec["csv_after_columns"], rec["csv_after_rows"] = [], []
                try:
                    lr, col = get_lr_at_epoch(epoch, os.path.join(d, "logs"))
                    rec["lr"], rec["lr_name"] = (None if lr is None else float(lr)), col
                except Exception as e:                   # the reference's own failure modes are part of the record
                    rec["lr_error"] = type(e).__name__
            except Exception as e:
                rec["error"] = type(e).__name__
            out.append(rec)
    dst = os.path.join(HERE, "..", "tests", "golden", "resume_golden.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    for r in out:
        print(r["name"], {k: v for k, v in r.items() if k not in ("csv_in", "csv_after_rows", "model_files", "name")})


if __name__ == "__main__":
    main()
