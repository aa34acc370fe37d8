"""
`mp train_fusion` on MI355X: flags and flow of mpunet/bin/train_fusion.py:44-362. Per round of
`images_per_round` images: every view is sampled, predicted and mapped back to the voxel grid on the GPU
(predict_and_map, utils/fusion/fusion_training.py:40-89), the [n_voxels, V, K] points stay in HBM, are
shuffled, split 80/20, and the FusionLayer is fitted with the per-point generalized Dice loss and Adam(1e-3)
(FusionModel.fit -> mpu_fusion_train_step), early-stopped on the validation Dice. Weights are written to
<project>/model/fusion_weights/<model>_fusion_weights.npz, where `mp predict` looks for them.

--num_GPUs N (one process per GPU under torch.distributed; the reference builds its models under MirroredStrategy,
train_fusion.py:336): the IMAGES of a round are dealt over the ranks -- each rank predicts and maps all views of its images
(the expensive part: V U-Net passes per image) and keeps their points -- and FusionModel.fit runs data parallel: per step one
SUM all-reduce of V*K + K + 2 doubles, per epoch one of the 3K validation counts. Every rank holds the same FusionLayer
weights at all times; rank 0 alone writes the weights file.
"""
import os
from argparse import ArgumentParser
import numpy as np
import torch

from .common import (validate_project_dir, load_hparams, load_dataset, require_audited_hparams,
                     fusion_weights_path)
from .predict import best_model_path


def get_argparser():
    p = ArgumentParser(description="Fit a fusion model for a mpunet project (MI355X hot path).")
    p.add_argument("--project_dir", type=str, default="./")
    p.add_argument("--overwrite", action="store_true")
    p.add_argument("--num_GPUs", type=int, default=1)
    p.add_argument("--images_per_round", type=int, default=5)
    p.add_argument("--batch_size", type=int, default=2 ** 17)
    p.add_argument("--epochs", type=int, default=30)
    p.add_argument("--early_stopping", type=int, default=3)
    p.add_argument("--continue_training", action="store_true")
    p.add_argument("--force_GPU", type=str, default="")
    p.add_argument("--eval_prob", type=float, default=1.0)
    p.add_argument("--wait_for", type=str, default="")
    p.add_argument("--dice_weight", type=str, default="uniform")
    p.add_argument("--synthetic", type=int, default=0, help="use N synthetic toy volumes instead of val_data")
    p.add_argument("--dtype", default="bf16", choices=("bf16", "f32", "bf16x3"),
                   help="bf16 (default, the benchmarked mode), f32 (exact-f32 MFMAs: the parity mode), bf16x3 (f32 storage, three bf16 "
                        "MFMAs per product: f32-grade results at 2.6x the f32 speed)")
    p.add_argument("--seed", type=int, default=None)
    return p


def predict_and_map(model, sampler, volume, view, batch_size, n_planes="same+20"):
    """One view: sample planes, predict, nearest-map to the voxel grid -> probabilities [X*Y*Z, K] (device)."""
    from ..interpolation import predict_volume, map_real_space_pred
    X, y, grid, inv_basis = sampler.get_view_from(volume, view, n_planes=n_planes)
    pred = predict_volume(model, X, axis=2, batch_size=batch_size)
    mapped = map_real_space_pred(pred, grid, inv_basis, volume)
    return mapped.reshape(-1, mapped.shape[-1]), (y, pred)


def collect_points(model, sampler, volumes, views, n_classes, batch_size, log, eval_prob=1.0, rng=None):
    """points [sum voxels, V, K] f32 and targets [sum voxels] u8, both resident on the GPU."""
    from ..interpolation import dice_all
    rng = rng or np.random
    xs, ys = [], []
    if not volumes:                                        # a rank without an image in this round (fewer images than ranks)
        dev = model.device
        return (torch.empty((0, len(views), n_classes), dtype=torch.float32, device=dev),
                torch.empty((0,), dtype=torch.uint8, device=dev))
    for vol in volumes:
        n = int(np.prod(vol.image.shape[:3]))
        pts = torch.empty((n, len(views), n_classes), dtype=torch.float32, device=vol.image.device)
        for k, view in enumerate(views):
            mapped, (yv, pred) = predict_and_map(model, sampler, vol, view, batch_size)
            pts[:, k, :] = mapped
            if rng.rand() <= eval_prob:
                d = dice_all(vol.labels.reshape(-1), mapped.argmax(-1), n_classes=n_classes, ignore_zero=False)
                log("  %s view %s: mapped dice %s" % (vol.identifier, np.round(view, 3), np.round(d, 4)))
        xs.append(pts)
        ys.append(vol.labels.reshape(-1).to(torch.uint8))
    return torch.cat(xs), torch.cat(ys)


def run(args):
    from ..unet import UNet
    from ..fusion_model import FusionModel
    from ..interpolation import ViewSampler
    project_dir = os.path.abspath(args.project_dir)
    validate_project_dir(project_dir)
    if args.force_GPU:
        os.environ["HIP_VISIBLE_DEVICES"] = args.force_GPU
    from .. import distributed as D
    rank, world, device = D.init_from_env()
    log = (lambda *a, **k: print(*a, flush=True)) if rank == 0 else (lambda *a, **k: None)
    seed = args.seed
    if world > 1:                                          # every rank draws the same image order, rounds and splits
        box = [int(np.random.randint(0, 2 ** 31 - 1)) if seed is None else int(seed)]
        torch.distributed.broadcast_object_list(box, src=0)
        seed = box[0]
    rng = np.random.RandomState(seed)
    hp = load_hparams(project_dir)
    fit, build = hp["fit"], hp["build"]
    views = np.load(os.path.join(project_dir, "views.npz"))["arr_0"]
    model_dir = os.path.join(project_dir, "model")
    wpath = best_model_path(model_dir)
    fpath = fusion_weights_path(model_dir, wpath)
    fdir = os.path.dirname(fpath)
    if os.path.exists(fpath) and not (args.overwrite or args.continue_training):
        raise OSError("Fusion weights already exist at '%s' (use --overwrite or --continue_training)" % fpath)
    # validation images first; training images are added when there are fewer than 15 (train_fusion.py:283-312)
    images = load_dataset(hp["val_data"], project_dir, hp, device, args.synthetic, seed=7000)
    if not args.synthetic and len(images) < 15:
        extra = load_dataset(hp["train_data"], project_dir, hp, device, 0, seed=0)
        if extra:
            need = 15 - len(images)
            idx = rng.choice(np.arange(len(extra)), need, replace=need > len(extra))
            images += [extra[i] for i in idx]
    if not images:
        raise OSError("no images to fit the fusion model on")
    require_audited_hparams(hp, "mp train_fusion")
    n_classes = int(build["n_classes"])
    bkw = {k: v for k, v in build.items() if k != "model_class_name"}
    unet = UNet(logger=log, dtype=args.dtype, device=device, **bkw)
    unet.load_weights(wpath, by_name=True)
    log("Loaded weights:", wpath)
    fm = FusionModel(len(views), n_classes, weight=args.dice_weight, logger=log, verbose=False, device=device)
    if args.continue_training and os.path.exists(fpath):
        fm.load_weights(fpath)
        log("[OBS] CONTINUED TRAINING FROM:", fpath)
    fm.compile("Adam", optimizer_kwargs={"lr": 1e-3})
    sampler = ViewSampler(views, build["dim"], fit["real_space_span"])
    # rounds of images_per_round images (appended to a multiple of the round size, shuffled)
    ids = list(range(len(images)))
    sub = max(1, min(args.images_per_round, len(ids)))
    rest = int(sub * np.ceil(len(ids) / sub)) - len(ids)
    if rest:
        ids += list(rng.choice(ids, rest, replace=False))
    rng.shuffle(ids)
    rounds = np.array_split(ids, len(ids) // sub)
    history = []
    if rank == 0:
        os.makedirs(fdir, exist_ok=True)
    for r, ids_r in enumerate(rounds):
        log("Set %d/%d: %s" % (r + 1, len(rounds), [images[i].identifier for i in ids_r]))
        mine = [images[i] for i in list(ids_r)[rank::world]]          # the round's images dealt over the ranks
        eval_rng = np.random.RandomState(int(rng.randint(0, 2 ** 31 - 1)) + rank)   # (keeps `rng` in step on every rank)
        X, y = collect_points(unet, sampler, mine, views, n_classes, int(fit["batch_size"]),
                              log, args.eval_prob, eval_rng)
        fit_seed = int(rng.randint(0, 2 ** 31 - 1))
        perm = torch.from_numpy(np.random.RandomState(fit_seed + 7919 * rank).permutation(X.shape[0])).to(device)
        X, y = X[perm], y[perm]
        nv = int(0.20 * X.shape[0])                                   # 80 / 20 split of the rank's own points
        h = fm.fit(X[nv:], y[nv:], batch_size=args.batch_size, epochs=args.epochs, validation_data=(X[:nv], y[:nv]),
                   early_stopping=args.early_stopping, verbose=1, seed=fit_seed + rank)
        history.append(h)
        if rank == 0:                                                 # one writer (the weights are identical on every rank)
            fm.save_weights(fpath)
        W, b = fm.get_weights()
        log("fusion weights W:\n%s\nb: %s" % (np.round(W, 4), np.round(b, 4)))
        del X, y
    if world > 1:
        torch.distributed.barrier()
    log("Saved fusion weights:", fpath)
    return history


def entry_func(args=None):
    import sys
    argv = list(sys.argv[1:] if args is None else args)
    args = get_argparser().parse_args(argv)
    from .common import relaunch_per_gpu, await_pids
    if args.wait_for and "RANK" not in os.environ:        # (once, in the launching process: utils.py:337-375)
        await_pids(args.wait_for)
    relaunch_per_gpu("train_fusion", argv, args.num_GPUs)   # --num_GPUs N > 1: one process per GPU (returns inside a torchrun job)
    run(args)


if __name__ == "__main__":
    entry_func()
