"""
The 6-view predict+fuse loop of `mp predict` (mpunet/bin/predict.py:294-366) on
one GPU: per view sample planes (HIP) -> U-Net forward (HIP) -> after all views one
fused nearest-map + weighted-softmax + argmax kernel. `combined[V,X,Y,Z,K]` and
the fp64 voxel grid of the reference are never materialised.
"""
import numpy as np
import torch

from .interpolation import ViewGeometry, sample_view, map_and_fuse, map_real_space_pred


def per_view_evaluation(pred, true, mapped_pred, mapped_true, n_classes):
    """_per_view_evaluation's arithmetic (mpunet/bin/predict.py:248-275; `evaluate` :236-246: argmax, then dice_all with
    ignore_zero=False) on the GPU: view_dices of the view-space prediction [P,dim,dim,K] against the sampled labels, mapped_dices of
    the back-mapped prediction [X,Y,Z,K] against the volume's labels -- one pass each of mpu_validation_count (argmax + integer TP /
    relevant / selected counts), Dice from the counts -- and mean_dice = mean of the non-NaN mapped dices without class 0."""
    from .validation import count_cm_elements, dice_from_counts
    view_dices = dice_from_counts(count_cm_elements(pred, true, n_classes))
    mapped_dices = dice_from_counts(count_cm_elements(mapped_pred, mapped_true, n_classes))
    mean_dice = mapped_dices[~np.isnan(mapped_dices)][1:].mean()
    return view_dices, mapped_dices, mean_dice


def multi_view_predict(model, volume, views, dim, real_space_span, fusion_model=None,
                       sum_fusion=False, batch_size=None, n_planes="same+20",
                       want_probs=True, timings=None, per_view_eval=None):
    """
    Returns (merged f32 [X,Y,Z,K] or None, merged_map u8 [X,Y,Z]).
    per_view_eval: None, or dict(eval_prob=float, n_classes=int, report=callable(view_index, view, view_dices, mapped_dices,
    mean_dice) [, log=callable]): the reference's per-view evaluation inside the loop (predict.py:334-346) for volumes with
    labels -- skipped for a view when np.random.rand() > eval_prob, as there.
    batch_size=None: even chunks of the view's planes as large as the kernels' operand bound allows
    (UNet.auto_batch; 276 planes of 256x256 -> 3 x 92). The same batch size gives the same bits run after run; ANOTHER batch size
    selects other kernel schedules for some layers, i.e. another bf16 rounding pattern (measured on an untrained configs[1] network,
    128^3: fused probabilities within 1.7e-3, 0.013 % of the labels -- voxels whose two best classes are within 3e-4 -- differ).
    fusion_model: object with .W (V,K) and .b (1,K) device tensors (FusionModel) or None with sum_fusion.
    """
    if fusion_model is None and not sum_fusion:
        raise ValueError("need a fusion model unless sum_fusion")
    view_preds = []
    ev = []
    for view in views:
        geom = ViewGeometry(view, dim, real_space_span, n_planes)
        if timings is not None:
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
        pve = per_view_eval if (per_view_eval is not None and volume.labels is not None) else None
        X, y_view = sample_view(volume, geom, want_labels=pve is not None)
        if timings is not None:
            e1.record()
        pred = model.predict(X, batch_size=batch_size)
        if pred.ndim == 3:                                  # flatten_output models
            pred = pred.reshape(X.shape[0], dim, dim, -1)
        if timings is not None:
            e2.record()
            ev.append((e0, e1, e2))
        view_preds.append((pred, (geom.real_axis, geom.real_axis, geom.offsets), geom.inv_basis,
                           geom.device_axes(volume.device)))
        if pve is not None:
            say = pve.get("log") or (lambda *a: None)
            if np.random.rand() > pve["eval_prob"]:
                say("Skipping evaluation for view %s... (eval_prob=%.3f)" % (view, pve["eval_prob"]))
            else:
                mapped = map_real_space_pred(pred.permute(1, 2, 0, 3), (geom.real_axis, geom.real_axis, geom.offsets),
                                             geom.inv_basis, volume)
                vd, md, mean = per_view_evaluation(pred, y_view, mapped, volume.labels, pve["n_classes"])
                del mapped
                say("View dice scores:   ", vd)
                say("Mapped dice scores: ", md)
                say("Mean dice (n=%i): " % (len(md) - 1), mean)
                pve["report"](len(view_preds) - 1, view, vd, md, mean)
    if timings is not None:
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record()
    W = b = None
    if not sum_fusion:
        W, b = fusion_model.W, fusion_model.b
    probs, labels = map_and_fuse(volume, view_preds, W, b, sum_fusion=sum_fusion,
                                 want_probs=want_probs, want_labels=True)
    if timings is not None:
        f1.record()
        torch.cuda.synchronize()
        timings["sample_ms"] = sum(a.elapsed_time(b_) for a, b_, _ in ev)
        timings["unet_ms"] = sum(b_.elapsed_time(c) for _, b_, c in ev)
        timings["map_fuse_ms"] = f0.elapsed_time(f1)
    return probs, labels
