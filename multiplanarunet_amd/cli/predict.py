"""
`mp predict` on MI355X: flags / project layout of mpunet/bin/predict.py:19-78,433-470; the 6-view
loop, back-mapping and fusion run in multiplanarunet_amd.predict (distributed: plane-sharded).
Outputs <out_dir>/nii_files/<id>_PRED.nii.gz for NIfTI inputs (mpunet/bin/predict.py:90-117; native writer, nifti.py) and
<id>_PRED.npz (labels + affine) for .npz / synthetic volumes; --out_format forces one of them.
"""
import os
from argparse import ArgumentParser
import numpy as np
import torch

from .common import (validate_project_dir, load_hparams, load_dataset, require_audited_hparams,
                     fusion_weights_path, note_inert_flags)


def get_argparser():
    p = ArgumentParser(description="Predict using a mpunet model (MI355X hot path).")
    p.add_argument("--project_dir", type=str, default="./")
    p.add_argument("-f", help="Predict on a single file (.nii / .nii.gz / .npz)")
    p.add_argument("-l", help="Optional single label file (.nii / .nii.gz / .npz) to use with -f")
    p.add_argument("--out_format", default="auto", choices=("auto", "nii", "npz"),
                   help="auto: <id>_PRED.nii.gz for NIfTI inputs, <id>_PRED.npz otherwise")
    p.add_argument("--dataset", type=str, default="test")
    p.add_argument("--out_dir", type=str, default="predictions")
    p.add_argument("--num_GPUs", type=int, default=1)
    p.add_argument("--sum_fusion", action="store_true")
    p.add_argument("--overwrite", action="store_true")
    p.add_argument("--no_eval", action="store_true")
    p.add_argument("--eval_prob", type=float, default=1.0)
    p.add_argument("--force_GPU", type=str, default="")
    p.add_argument("--save_input_files", action="store_true")
    p.add_argument("--no_argmax", action="store_true")
    p.add_argument("--on_val", action="store_true")
    p.add_argument("--wait_for", type=str, default="")
    p.add_argument("--continue", action="store_true", dest="continue_")
    p.add_argument("--synthetic", type=int, default=0)
    p.add_argument("--dtype", default="bf16", choices=("bf16", "f32", "bf16x3"),
                   help="bf16 (default, the benchmarked mode), f32 (exact-f32 MFMAs: the parity mode), bf16x3 (f32 storage, three bf16 "
                        "MFMAs per product: f32-grade results at 2.6x the f32 speed)")
    return p


def best_model_path(model_dir):
    """get_best_model (mpunet/utils/utils.py:88-110) over this build's .npz and the reference's .h5 checkpoints."""
    from ..formats import get_best_model
    return get_best_model(model_dir)


def run(args):
    from .. import distributed as D
    from ..unet import UNet
    from ..fusion_model import FusionModel
    from ..predict import multi_view_predict
    from ..interpolation import dice_all
    from ..data import load_volume_file, load_label_file, as_volume
    from ..nifti import volume_identifier
    from ..formats import save_nifti
    project_dir = os.path.abspath(args.project_dir)
    validate_project_dir(project_dir)
    for req in ("views.npz", "model"):
        if not os.path.exists(os.path.join(project_dir, req)):
            raise RuntimeError("Invalid project folder: needs train_hparams.yaml, views.npz and model/ (missing %s)" % req)
    if args.force_GPU:
        os.environ["HIP_VISIBLE_DEVICES"] = args.force_GPU
    rank, world, device = D.init_from_env()
    log = (lambda *a, **k: print(*a, flush=True)) if rank == 0 else (lambda *a, **k: None)
    hp = load_hparams(project_dir)
    fit, build = hp["fit"], hp["build"]
    if args.f:
        img, lab, aff = load_volume_file(args.f)
        if args.l:
            lab = load_label_file(args.l)
        vols = [as_volume(img, lab, aff, fit.get("bg_value"), fit.get("scaler"), device, volume_identifier(args.f))]
        vols[0].source_path = args.f
    else:
        key = "val_data" if args.on_val else args.dataset.replace("_data", "") + "_data"
        vols = load_dataset(hp[key], project_dir, hp, device, args.synthetic, seed=5000, need_labels=False)
    if not vols:
        raise OSError("no volumes to predict on")
    require_audited_hparams(hp, "mp predict")             # geometry of the training session, never re-audited here
    views = np.load(os.path.join(project_dir, "views.npz"))["arr_0"]
    bkw = {k: v for k, v in build.items() if k != "model_class_name"}
    model = UNet(logger=log, dtype=args.dtype, device=device, **bkw)
    wpath = best_model_path(os.path.join(project_dir, "model"))
    model.load_weights(wpath, by_name=True)
    log("Loaded weights:", wpath)
    fm = None
    if not args.sum_fusion:
        fm = FusionModel(len(views), build["n_classes"], verbose=False, device=device)
        fpath = fusion_weights_path(os.path.join(project_dir, "model"), wpath)
        if os.path.exists(fpath):
            fm.load_weights(fpath)
            log("Loaded fusion weights:", fpath)
        else:
            log("[OBS] no fusion weights for this checkpoint (%s): using the FusionLayer initialisation "
                "(W=1, b=0)" % os.path.basename(fpath))
    out_dir = os.path.join(project_dir, args.out_dir) if not os.path.isabs(args.out_dir) else args.out_dir
    nii = os.path.join(out_dir, "nii_files")
    if rank == 0:
        os.makedirs(nii, exist_ok=True)
    results, per_view = {}, {}
    for v in vols:
        src = str(getattr(v, "source_path", "") or "")
        as_nii = args.out_format == "nii" or (args.out_format == "auto" and src.endswith((".nii", ".nii.gz")))
        # bin/predict.py:90-117: with --save_input_files the prediction goes into a sub-folder <identifier>/ of nii_files,
        # next to <identifier>_IMAGE and <identifier>_LABELS
        out_base = os.path.join(nii, v.identifier) if args.save_input_files else nii
        dst = os.path.join(out_base, "%s_PRED.%s" % (v.identifier, "nii.gz" if as_nii else "npz"))
        if os.path.exists(dst) and args.continue_:
            continue
        if os.path.exists(dst) and not args.overwrite:
            raise OSError("%s exists (use --overwrite or --continue)" % dst)
        if world > 1:
            res = D.multi_view_predict_sharded(model, v, views, build["dim"], fit["real_space_span"], fm,
                                               sum_fusion=args.sum_fusion, batch_size=None, want_probs=args.no_argmax)
            probs, labels = res if args.no_argmax else (None, res)
        else:
            pve = None
            if v.labels is not None and not args.no_eval:     # bin/predict.py:334-346: per-view Dice inside the loop
                rows = per_view.setdefault(v.identifier, {})
                pve = dict(eval_prob=args.eval_prob, n_classes=build["n_classes"], log=log,
                           report=lambda i, view, vd, md, mean, rows=rows: rows.__setitem__(i, (view, vd, md, float(mean))))
            probs, labels = multi_view_predict(model, v, views, build["dim"], fit["real_space_span"], fm,
                                               sum_fusion=args.sum_fusion, batch_size=None,
                                               want_probs=args.no_argmax, per_view_eval=pve)
        if rank == 0 and args.save_input_files:
            sub = out_base                                # the volume as it was read: unscaled image, label map
            os.makedirs(sub, exist_ok=True)
            raw = v.image.cpu().numpy()                   # (the volume as read: the scaler is applied by the sampling kernel)
            raw = raw[..., 0] if raw.shape[-1] == 1 else raw
            lab_h = None if v.labels is None else v.labels.cpu().numpy().astype(np.uint8)
            if as_nii:
                save_nifti(os.path.join(sub, "%s_IMAGE.nii.gz" % v.identifier), raw, v.affine)
                if lab_h is not None:
                    save_nifti(os.path.join(sub, "%s_LABELS.nii.gz" % v.identifier), lab_h, v.affine)
            else:
                np.savez_compressed(os.path.join(sub, "%s_IMAGE.npz" % v.identifier), image=raw, affine=v.affine)
                if lab_h is not None:
                    np.savez_compressed(os.path.join(sub, "%s_LABELS.npz" % v.identifier), labels=lab_h, affine=v.affine)
        if rank == 0:
            if as_nii:                                # the label map, or with --no_argmax the fused [X,Y,Z,K] probabilities
                pred = probs.cpu().numpy() if (args.no_argmax and probs is not None) else labels.cpu().numpy().astype(np.uint8)
                save_nifti(dst, pred, v.affine)
            else:
                out = {"labels": labels.cpu().numpy(), "affine": v.affine}
                if args.no_argmax and probs is not None:
                    out["probs"] = probs.cpu().numpy()
                np.savez_compressed(dst, **out)
            if v.labels is not None and not args.no_eval:
                d = dice_all(v.labels, labels, n_classes=build["n_classes"], ignore_zero=True)
                results[v.identifier] = d
                log("%s: dices %s mean %.4f" % (v.identifier, np.round(d, 4), float(np.nanmean(d))))
    if rank == 0 and results:
        os.makedirs(os.path.join(out_dir, "csv"), exist_ok=True)
        with open(os.path.join(out_dir, "csv", "results.csv"), "w") as f:
            f.write("image,mean_dice," + ",".join("class_%d" % (c + 1) for c in range(build["n_classes"] - 1)) + "\n")
            for k, d in results.items():
                f.write("%s,%.5f,%s\n" % (k, float(np.nanmean(d)), ",".join("%.5f" % x for x in d)))
        if any(per_view.values()):                      # one row per (image, evaluated view): mean + per-class mapped Dice
            with open(os.path.join(out_dir, "csv", "per_view.csv"), "w") as f:
                f.write("image,view_index,view,mean_dice," + ",".join("class_%d" % c for c in range(build["n_classes"])) + "\n")
                for k, rows in per_view.items():
                    for i in sorted(rows):
                        view, _, md, mean = rows[i]
                        f.write("%s,%d,%s,%.5f,%s\n" % (k, i, " ".join("%.6g" % x for x in np.asarray(view).ravel()), mean,
                                                        ",".join("%.5f" % x for x in md)))
    return results


def entry_func(args=None):
    import sys
    argv = list(sys.argv[1:] if args is None else args)
    args = get_argparser().parse_args(argv)
    from .common import relaunch_per_gpu, await_pids
    if args.wait_for and "RANK" not in os.environ:        # (once, in the launching process: utils.py:337-375)
        await_pids(args.wait_for)
    relaunch_per_gpu("predict", argv, args.num_GPUs)     # --num_GPUs N > 1: one process per GPU (returns inside a torchrun job)
    run(args)


if __name__ == "__main__":
    entry_func()
