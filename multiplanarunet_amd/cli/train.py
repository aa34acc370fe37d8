"""
`mp train` on MI355X: the flags and project layout of mpunet/bin/train.py:18-107,320-376, with the
training step, the plane sampler and the validation forward on the GPU. One process per GPU under
torchrun replaces MirroredStrategy (gradient SUM all-reduce over RCCL).
"""
import os
import shutil
from argparse import ArgumentParser
import numpy as np
import torch

from .common import (validate_project_dir, load_hparams, load_dataset, load_or_create_views,
                     fill_build_from_data, save_audited_hparams, set_bias_weights_on_all_outputs,
                     init_callback_objects, remove_validation_callbacks, DEFAULT_CALLBACKS, note_inert_flags)


def get_argparser():
    p = ArgumentParser(description="Fit a mpunet model defined in a project folder (MI355X hot path).")
    p.add_argument("--project_dir", type=str, default="./")
    p.add_argument("--num_GPUs", type=int, default=1, help="(one process per GPU: launch with torchrun for >1)")
    p.add_argument("--force_GPU", type=str, default="", help="sets HIP_VISIBLE_DEVICES")
    p.add_argument("--continue_training", action="store_true")
    p.add_argument("--overwrite", action="store_true")
    p.add_argument("--just_one", action="store_true")
    p.add_argument("--no_val", action="store_true")
    p.add_argument("--no_images", action="store_true")
    p.add_argument("--debug", action="store_true")
    p.add_argument("--wait_for", type=str, default="")
    p.add_argument("--train_images_per_epoch", type=int, default=2500)
    p.add_argument("--val_images_per_epoch", type=int, default=3500)
    p.add_argument("--max_loaded_images", type=int, default=None)
    p.add_argument("--epochs", type=int, default=None)
    p.add_argument("--num_access", type=int, default=50)
    p.add_argument("--cpu", action="store_true", help="alias of --num_GPUs=0 (rejected: there is no CPU path)")
    p.add_argument("--synthetic", type=int, default=0, help="train on N generated toy volumes (no files needed)")
    p.add_argument("--dtype", default="bf16", choices=("bf16", "f32", "bf16x3"),
                   help="bf16 (default, the benchmarked mode), f32 (exact-f32 MFMAs: the parity mode), bf16x3 (f32 storage, three bf16 "
                        "MFMAs per product: f32-grade results at 2.6x the f32 speed)")
    p.add_argument("--no_overlap", action="store_true", help="cut every batch on the training stream (no side-stream "
                   "producer; A/B aid: the default overlaps the sampler with the train step)")
    p.add_argument("--no_graph", action="store_true", help="launch the train step's kernels eagerly instead of replaying "
                   "one HIP graph (A/B aid, independent of --no_overlap)")
    return p


RANK0_ONLY = ("ModelCheckPointClean", "CSVLogger")         # file writers: one process only


def runs_on_this_rank(cb, rank):
    """File-writing callbacks run on rank 0 only; a DelayedCallback is judged by the callback it wraps (the wrapper's
    own class name let every rank write and remove the same checkpoint / CSV file)."""
    return rank == 0 or type(getattr(cb, "callback", cb)).__name__ not in RANK0_ONLY


def validate_args(args):
    if args.continue_training and args.overwrite:
        raise ValueError("Cannot both continue training and overwrite the previous training session.")
    if args.train_images_per_epoch <= 0:
        raise ValueError("train_images_per_epoch must be a positive integer")
    if args.val_images_per_epoch <= 0:
        raise ValueError("val_images_per_epoch must be a positive integer. Use --no_val instead.")
    if args.force_GPU and args.num_GPUs != 1:
        raise ValueError("Should not specify both --force_GPU and --num_GPUs")
    if args.num_GPUs < 0:
        raise ValueError("num_GPUs must be a positive integer")
    if args.num_GPUs == 0 or args.cpu:
        raise NotImplementedError("this build has no CPU execution path (reference: 'Using CPU based "
                                  "computations only!', mpunet/utils/system.py:80-81)")


def run(args):
    from .. import distributed as D
    from ..unet import UNet
    from ..data import TrainSampler
    project_dir = os.path.abspath(args.project_dir)
    validate_project_dir(project_dir)
    if args.force_GPU:
        os.environ["HIP_VISIBLE_DEVICES"] = args.force_GPU
    rank, world, device = D.init_from_env()
    log = (lambda *a, **k: print(*a, flush=True)) if rank == 0 else (lambda *a, **k: None)
    model_dir = os.path.join(project_dir, "model")
    # checks that can fail run on EVERY rank before the first barrier (a rank-0-only raise would leave the
    # others waiting in it); only rank 0 then touches the project folder
    if os.path.exists(model_dir) and os.listdir(model_dir) and not (args.overwrite or args.continue_training):
        raise OSError("There seems to be existing files in the project 'model' folder. "
                      "Use --overwrite or --continue_training.")
    hp = load_hparams(project_dir)
    note_inert_flags(args, [("no_images", "no sample images are written during training"), ("debug", "the TF debugger has no counterpart"),
                            ("max_loaded_images", "volumes stay resident in HBM: there is no image queue"),
                            ("num_access", "volumes stay resident in HBM: there is no image queue")], log)
    train = load_dataset(hp["train_data"], project_dir, hp, device, args.synthetic, seed=0)
    val = [] if args.no_val else load_dataset(hp["val_data"], project_dir, hp, device,
                                              max(1, args.synthetic // 4) if args.synthetic else 0, seed=1000)
    if not train:
        raise OSError("no training volumes (set train_data.base_dir to a folder with images/*.npz, or --synthetic N)")
    fill_build_from_data(hp, train)                        # audit the FULL training set (before --just_one)
    train_all = list(train)
    if args.just_one:
        train, val = train[:1], val[:1]
    fit, build = hp["fit"], hp["build"]
    if world > 1:
        torch.distributed.barrier()                        # every rank has passed the checks and read the YAML
    views = None
    if rank == 0:
        if args.overwrite and os.path.exists(model_dir):
            shutil.rmtree(model_dir)
        os.makedirs(model_dir, exist_ok=True)
        os.makedirs(os.path.join(project_dir, "logs"), exist_ok=True)
        save_audited_hparams(project_dir, hp)              # Auditor.fill: predict / train_fusion read them back
        views = load_or_create_views(project_dir, fit["views"], seed=0)
    if world > 1:
        torch.distributed.barrier()
    if views is None:
        views = np.load(os.path.join(project_dir, "views.npz"))["arr_0"]
    bkw = {k: v for k, v in build.items() if k != "model_class_name"}
    model = UNet(logger=log, flatten_output=True, dtype=args.dtype, device=device, **bkw)
    last = os.path.join(model_dir, "model_weights.npz")
    init_epoch = 0
    if args.continue_training:
        # model_init.py:23-47: newest @epoch_ checkpoint, init_epoch, the learning rate logged for that epoch, CSV cut back.
        # Rank 0 decides (it rewrites logs/training.csv) and every rank applies the same decision.
        from ..resume import resume_state, apply_resume
        state = [resume_state(project_dir, log) if rank == 0 else None]
        if world > 1:
            torch.distributed.broadcast_object_list(state, src=0)
        apply_resume(model, hp, state[0], log)
        init_epoch = int(fit["init_epoch"])
    # Initialize the bias of the output layer from the class frequencies (bin/train.py:293-299; YAML default
    # build.biased_output_layer: True). Counted on the volumes of the FULL training set of this rank's process: every rank
    # loads the same volumes, so the replicas start from identical weights.
    if not args.continue_training and build.get("biased_output_layer"):
        set_bias_weights_on_all_outputs(model, train_all, hp, log)
    model.compile(fit["optimizer"], fit["loss"], fit.get("metrics"), optimizer_kwargs=fit.get("optimizer_kwargs"))
    if world > 1:
        D.DataParallelTrainer(model)
    B = int(fit["batch_size"])
    per_rank = max(1, B // world)
    from ..augmentation import build_augmenters
    mk = lambda vols, noise, seed, augs: TrainSampler(vols, views, build["dim"], fit["real_space_span"], per_rank,
                                                      build["n_classes"], noise_sd=noise,
                                                      fg_batch_fraction=fit["fg_batch_fraction"], seed=seed,
                                                      augmenters=augs)
    augs = build_augmenters(fit.get("augmenters"), seed=1000 + rank)     # YAML fit.augmenters (Elastic2D)
    if augs:
        log("Augmenters:", ", ".join(str(a) for a in augs))
    tr = mk(train, fit["noise_sd"], 17 + rank, augs)
    va = mk(val, 0.0, 99 + rank, None) if val else None      # distinct planes per rank; counts are SUM-reduced
    epochs = args.epochs or int(fit["n_epochs"])
    steps = max(1, int(np.ceil(args.train_images_per_epoch / B)))
    vsteps = max(1, int(np.ceil(args.val_images_per_epoch / B)))
    from ..validation import Validation, ReduceLROnPlateau, EarlyStopping, ModelCheckPointClean
    validation = Validation(va, vsteps, build["n_classes"], logger=log, verbose=rank == 0) if va is not None else None
    # the YAML's callback list fit.callbacks (bin/defaults/MultiPlanar/train_hparams.yaml:7-45,139; the descriptors
    # behind the __CB_* anchors): kwargs are honoured, a project that edits them gets what it wrote
    descr = fit.get("callbacks")
    descr = [dict(c) for c in descr] if descr is not None else [dict(c) for c in DEFAULT_CALLBACKS]
    if validation is None:
        descr = remove_validation_callbacks(descr, log)    # bin/train.py:259-262 (--no_val)
    callbacks, cb_by_name = init_callback_objects(descr, project_dir, log)
    # Producer / consumer overlap (trainer.py:246-257: fit(workers=5, max_queue_size=5)): batch i+1 is cut on a side
    # stream while step i -- one HIP-graph replay at N = 1 -- runs; the loss is summed on the device and read once per
    # epoch (pipeline.TrainPipeline). No host synchronisation per step.
    from ..pipeline import TrainPipeline
    pipe = TrainPipeline(model, tr, overlap=not args.no_overlap, graphed=False if args.no_graph else None)
    if pipe.side_latency_us is not None:
        log("Producer stream: probe latency %.0f us" % pipe.side_latency_us)
        if pipe.side_latency_us >= 1500.0 and pipe._cal is None:   # (no candidates to try under the real loop: a stream BEHIND the training stream's queue)
            log("[WARNING] no hardware queue beside the training stream was found: the batch producer will run behind every "
                "train step (about 0.65 of the step rate). --no_overlap gives the serial loop.")
    try:
        for ep in range(init_epoch, epochs):
            loss = pipe.run_epoch(steps)
            if ep == init_epoch and pipe.side_loop_ms:     # (the first CAL_STREAMS x CAL_WINDOW steps tried the candidates)
                log("Producer stream: candidates under the training loop %s ms per step%s" % (
                    pipe.side_loop_ms, "" if pipe._cal is None else " (choice continues next epoch)"))
            if world > 1:                                  # the logged loss is the mean over all replicas' slices
                t = torch.tensor([loss], dtype=torch.float64, device=device)
                torch.distributed.all_reduce(t)
                loss = float(t.item()) / world
            logs = {"loss": loss}
            if validation is not None:
                validation.on_epoch_end(model, ep, logs)
            log("Epoch %d/%d - " % (ep + 1, epochs) + " - ".join("%s: %.5f" % kv for kv in logs.items()))
            logs["lr"] = model.optimizer_kwargs["lr"]
            for cb in callbacks:                           # list order, as Keras runs them
                if runs_on_this_rank(cb, rank):
                    cb.on_epoch_end(model, ep, logs)
            if model.stop_training:
                break
    except KeyboardInterrupt:
        log("Interrupted: saving weights")
    finally:
        if rank == 0:
            model.save_weights(last)
            log("Saved", last)
            try:                                           # bin/train.py:303-317: [project]/model/model_weights.h5 (Keras layout),
                from ..formats import _h5_backend          # where this host can write HDF5 (h5py, or libhdf5 through hdf5.py)
                _h5_backend()
                model.save_weights(os.path.splitext(last)[0] + ".h5")
                log("Saved", os.path.splitext(last)[0] + ".h5")
            except ImportError:
                pass
    return model


def entry_func(args=None):
    import sys
    argv = list(sys.argv[1:] if args is None else args)
    args = get_argparser().parse_args(argv)
    validate_args(args)
    from .common import relaunch_per_gpu, await_pids
    if args.wait_for and "RANK" not in os.environ:        # (once, in the launching process: utils.py:337-375)
        await_pids(args.wait_for)
    relaunch_per_gpu("train", argv, args.num_GPUs)       # --num_GPUs N > 1: one process per GPU (returns inside a torchrun job)
    run(args)


if __name__ == "__main__":
    entry_func()
