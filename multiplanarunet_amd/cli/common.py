"""Shared pieces of the `mp train` / `mp predict` shims: YAML hyper-parameters, views, datasets."""
import os
import numpy as np
import yaml

from ..data import list_volume_files, load_volume_file, as_volume, make_toy_volume, random_views, audit_dim_and_span

DEFAULT_HPARAMS = {
    "train_data": {"base_dir": None, "img_subdir": "images", "label_subdir": "labels", "bg_class": 0},
    "val_data": {"base_dir": None, "img_subdir": "images", "label_subdir": "labels", "bg_class": 0},
    "test_data": {"base_dir": None, "img_subdir": "images", "label_subdir": "labels", "bg_class": 0},
    "build": {"model_class_name": "UNet", "n_classes": None, "n_channels": None, "dim": None,
              "complexity_factor": 2, "out_activation": "softmax", "l1_reg": False, "l2_reg": False,
              "biased_output_layer": True, "depth": 4},
    "fit": {"views": 6, "noise_sd": 0.1, "real_space_span": None, "intrp_style": "iso_live",
            "loss": "SparseCategoricalCrossentropy", "metrics": ["sparse_categorical_accuracy"],
            "batch_size": 16, "n_epochs": 500, "optimizer": "Adam",
            "optimizer_kwargs": {"lr": 5.0e-05, "decay": 0.0, "beta_1": 0.9, "beta_2": 0.999, "epsilon": 1.0e-8},
            "fg_batch_fraction": 0.50, "bg_value": "1pct", "scaler": "RobustScaler"},
}


def validate_project_dir(project_dir):
    if not os.path.exists(project_dir) or not os.path.exists(os.path.join(project_dir, "train_hparams.yaml")):
        raise RuntimeError("The script was launched from directory:\n'%s'\n... but this is not a valid project "
                           "folder (no 'train_hparams.yaml')." % project_dir)


def load_hparams(project_dir):
    """train_hparams.yaml with the reference's sections (bin/defaults/MultiPlanar/train_hparams.yaml)."""
    with open(os.path.join(project_dir, "train_hparams.yaml")) as f:
        raw = yaml.safe_load(f) or {}
    hp = {k: dict(v) for k, v in DEFAULT_HPARAMS.items()}
    for sec, vals in raw.items():
        if sec.startswith("__") or not isinstance(vals, dict):
            continue
        hp.setdefault(sec, {}).update(vals)
    return hp


def load_dataset(cfg, project_dir, hp, device, synthetic=0, seed=0, need_labels=True):
    """List of Volume objects for one of train_data / val_data / test_data (or synthetic toy volumes)."""
    fit = hp["fit"]
    vols = []
    if synthetic:
        for i in range(synthetic):
            img, lab, aff = make_toy_volume(64, seed + i)
            vols.append(as_volume(img, lab, aff, fit.get("bg_value"), fit.get("scaler"), device, "toy_%d" % (seed + i)))
        return vols
    base = cfg.get("base_dir")
    if not base:
        return vols
    if not os.path.isabs(base):
        base = os.path.join(project_dir, base)
    for path in list_volume_files(base, cfg.get("img_subdir", "images")):
        img, lab, aff = load_volume_file(path)
        if lab is None:
            lp = os.path.join(base, cfg.get("label_subdir", "labels"), os.path.basename(path))
            if os.path.exists(lp):
                with np.load(lp) as z:
                    lab = z["labels"] if "labels" in z.files else z[z.files[0]]
        if lab is None and need_labels:
            raise ValueError("no labels for %s" % path)
        ident = os.path.splitext(os.path.basename(path))[0]
        vols.append(as_volume(img, lab, aff, fit.get("bg_value"), fit.get("scaler"), device, ident))
    return vols


def load_or_create_views(project_dir, n_views, seed=None):
    """views.npz (key arr_0, shape [V,3]) as data_preparation_funcs.py:116-154 persists it."""
    path = os.path.join(project_dir, "views.npz")
    if os.path.exists(path):
        return np.load(path)["arr_0"]
    views = random_views(int(n_views), 60.0, seed)
    np.savez(path, views)
    return views


def fill_build_from_data(hp, volumes, n_classes=None):
    """What Auditor.fill writes back into the YAML: dim, real_space_span, n_channels, n_classes
    (mpunet/image/auditor.py:100-120,199-209). Call on the FULL training set (before --just_one truncation)."""
    b, f = hp["build"], hp["fit"]
    dim, span = audit_dim_and_span(volumes, min_dim=32)
    if not b.get("dim"):
        b["dim"] = dim
    if not f.get("real_space_span"):
        f["real_space_span"] = span
    if not b.get("n_channels"):
        b["n_channels"] = volumes[0].n_channels
    if not b.get("n_classes"):
        labelled = [int(v.labels.max().item()) for v in volumes if v.labels is not None]
        if n_classes is None and not labelled:
            raise ValueError("build.n_classes is not set and no labelled volume is available to audit it from")
        b["n_classes"] = n_classes or max(labelled) + 1
    return hp


AUDITED_KEYS = (("build", "dim"), ("build", "n_channels"), ("build", "n_classes"), ("fit", "real_space_span"))


def _patch_yaml_value(text, sec, key, val):
    """Set `sec.key` in the YAML TEXT, touching only that line (comments, anchors and layout of the rest survive, as
    with the reference's YAMLHParams.set_value, mpunet/hyperparameters/hparams.py:161-221). Works on the block style
    the project YAMLs use (top-level `sec:` line, indented `key: value` lines)."""
    import re
    lines = text.split("\n")
    start = next((i for i, l in enumerate(lines) if re.match(r"^%s\s*:" % re.escape(sec), l)), None)
    if start is None:                                   # section missing: append it
        return text.rstrip("\n") + "\n\n%s:\n  %s: %s\n" % (sec, key, val)
    end = start + 1
    while end < len(lines) and (not lines[end].strip() or lines[end][:1] in " \t#"):
        end += 1
    pat = re.compile(r"^(\s+)%s\s*:\s*([^#]*?)(\s*#.*)?$" % re.escape(key))
    for i in range(start + 1, end):
        mt = pat.match(lines[i])
        if mt:
            lines[i] = "%s%s: %s%s" % (mt.group(1), key, val, mt.group(3) or "")
            return "\n".join(lines)
    last = end - 1                                      # key missing: add it behind the section's last entry
    while last > start and not lines[last].strip():
        last -= 1
    lines.insert(last + 1, "  %s: %s" % (key, val))
    return "\n".join(lines)


def save_audited_hparams(project_dir, hp):
    """Write the audited values back into train_hparams.yaml (the reference's Auditor.fill +
    YAMLHParams.save_current, mpunet/bin/train.py:210-228), so that `mp predict` / `mp train_fusion` use the
    geometry the model was trained with instead of re-auditing whatever volumes they are given. Only the audited
    lines are patched in the text: the user's comments and formatting are kept."""
    path = os.path.join(project_dir, "train_hparams.yaml")
    with open(path) as f:
        text = f.read()
    raw = yaml.safe_load(text) or {}
    changed = False
    for sec, key in AUDITED_KEYS:
        val = hp[sec].get(key)
        if val is None:
            continue
        val = float(val) if key == "real_space_span" else int(val)
        cur = raw.get(sec) if isinstance(raw.get(sec), dict) else {}
        if cur.get(key) != val:
            text = _patch_yaml_value(text, sec, key, repr(val))
            changed = True
    if changed:
        check = yaml.safe_load(text) or {}
        for sec, key in AUDITED_KEYS:                   # the patched text must parse back to the audited values
            val = hp[sec].get(key)
            if val is not None and float((check.get(sec) or {}).get(key, float("nan"))) != float(val):
                raise RuntimeError("could not patch %s.%s in %s" % (sec, key, path))
        tmp = path + ".tmp"
        with open(tmp, "w") as f:
            f.write(text)
        os.replace(tmp, path)
    return changed


def require_audited_hparams(hp, what):
    """`mp predict` / `mp train_fusion`: the model geometry must come from the training session's YAML."""
    missing = ["%s.%s" % (sec, key) for sec, key in AUDITED_KEYS if not hp[sec].get(key)]
    if missing:
        raise RuntimeError("%s: train_hparams.yaml lacks %s -- these are written by `mp train` (Auditor); "
                           "run it first or set them by hand. Re-auditing the volumes given here could silently "
                           "change the sampling geometry the weights were trained with." % (what, ", ".join(missing)))


def fusion_weights_path(model_dir, weights_path):
    """<model>/fusion_weights/<checkpoint name>_fusion_weights.npz (mpunet/bin/predict.py:222-229,
    mpunet/bin/train_fusion.py:318-325): fusion weights belong to ONE U-Net checkpoint."""
    base = os.path.splitext(os.path.basename(weights_path))[0]
    return os.path.join(model_dir, "fusion_weights", "%s_fusion_weights.npz" % base)
