"""
ORACLE (test infrastructure only: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline) - restatement
of the reference's fusion-model training arithmetic with torch-CPU autograd.

  FusionLayer ........................ mpunet/models/fusion_model.py:9-39
      out = softmax(sum_v W[v,k] * x[n,v,k] + b[0,k]); regulariser 1e-6 * mean(W^2) on W and on b
  sparse generalized Dice loss ....... mpunet/evaluate/loss_functions.py:23-30,207-246, restated op by op
      (one-hot, EMPTY reduction dims for [N,K] predictions, inf class weights -> largest finite weight,
      eps 1e-6, 1 - mean over classes), reduction SUM_OVER_BATCH_SIZE (fusion_model.py:55-56)
  optimiser .......................... Adam(lr=1e-3), Keras defaults b1 .9, b2 .999, eps 1e-7 (train_fusion.py:343)

PARITY UNPINNED: the arithmetic runs inside tensorflow==2.3.2 (absent here); the reference's tests hold no
vectors for it. tests/test_oracle_fusion_kat.py checks this restatement against hand-derived closed forms.
"""
import numpy as np
import torch

from .unet_ref import adam_update


def sparse_generalized_dice_loss(y_true, y_pred, type_weight="Simple"):
    """y_true [N] or [N,1] integer, y_pred [N,K] -> per-sample loss [N,1] (loss_functions.py:207-246)."""
    n_classes = y_pred.shape[-1]
    yt = y_true.reshape(y_pred.shape[:-1]).long()
    one_hot = torch.zeros_like(y_pred)
    ok = (yt >= 0) & (yt < n_classes)                       # tf.one_hot: out-of-range index -> all zeros
    one_hot[ok.nonzero(as_tuple=True) + (yt[ok],)] = 1.0
    # reduction_dims = range(len(shape))[1:-1] is EMPTY for rank-2 predictions: no reduction
    ref_vol, intersect, seg_vol = one_hot, one_hot * y_pred, y_pred
    tw = type_weight.lower()
    if tw == "square":
        weights = 1.0 / (ref_vol * ref_vol)
    elif tw == "simple":
        weights = 1.0 / ref_vol
    elif tw == "uniform":
        weights = torch.ones_like(ref_vol)
    else:
        raise ValueError('The variable type_weight "%s" is not defined.' % type_weight)
    new_weights = torch.where(torch.isinf(weights), torch.zeros_like(weights), weights)
    weights = torch.where(torch.isinf(weights), torch.ones_like(weights) * new_weights.max(), weights)
    eps = 1e-6
    score = 2 * weights * intersect / (weights * (seg_vol + ref_vol) + eps)
    return 1 - score.mean(dim=-1, keepdim=True)


def fusion_forward(W, b, x):
    return torch.softmax((W[None] * x).sum(dim=1) + b.reshape(1, -1), dim=-1)


def reg(t):
    return 1e-6 * (t * t).sum() / t.numel()


def loss_and_grads(W, b, x, y, type_weight="Simple", dtype=torch.float64):
    """Batch loss (mean of per-sample GDL + regularisers) and its gradients. numpy in, numpy out."""
    Wt = torch.tensor(np.asarray(W), dtype=dtype, requires_grad=True)
    bt = torch.tensor(np.asarray(b).reshape(1, -1), dtype=dtype, requires_grad=True)
    xt = torch.tensor(np.asarray(x), dtype=dtype)
    yt = torch.tensor(np.asarray(y).reshape(-1).astype(np.int64))
    loss = sparse_generalized_dice_loss(yt, fusion_forward(Wt, bt, xt), type_weight).mean() + reg(Wt) + reg(bt)
    loss.backward()
    return float(loss.detach()), Wt.grad.numpy(), bt.grad.numpy()


def train_step(W, b, m, v, t, x, y, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7, type_weight="Simple"):
    """One fit() batch: returns loss (before the update), new (W, b, m, v); m, v = dict(W=..., b=...)."""
    loss, gW, gb = loss_and_grads(W, b, x, y, type_weight)
    W2, mW, vW = adam_update(np.asarray(W, np.float32), gW.astype(np.float32), m["W"], v["W"], t, lr, b1, b2, eps)
    b2_, mb, vb = adam_update(np.asarray(b, np.float32).reshape(1, -1), gb.astype(np.float32), m["b"], v["b"], t, lr, b1, b2, eps)
    return loss, W2.astype(np.float32), b2_.astype(np.float32), dict(W=mW, b=mb), dict(W=vW, b=vb), (gW, gb)
