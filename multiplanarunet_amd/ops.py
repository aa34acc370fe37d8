"""
Thin torch-tensor wrappers over the single-layer C-ABI entry points
(mpu_conv2d_*). Used by the unit tests and for layer-wise integration; the
U-Net itself runs through mpu_unet_forward / mpu_unet_backward.
"""
import torch
from . import _lib

CONV3, UPCONV2, CONV3S2, CONV1 = 0, 1, 2, 3
_NTAPS = {CONV3: 9, UPCONV2: 4, CONV3S2: 9, CONV1: 1}


def _dt(dtype):
    return {torch.float32: _lib.MPU_F32, torch.bfloat16: _lib.MPU_BF16}[dtype]


def pack_weights(w_hwio, mode, dtype, x3=False):
    """fp32 Keras HWIO kernel (device) -> (forward operand, data-gradient operand). x3: the split-bf16 form of the f32 operands
    (mpu_dtype MPU_F32X3: each word holds bf16 hi | bf16 lo), for conv2d(x3=True)."""
    kh, kw, ci, co = w_hwio.shape
    w = w_hwio.to(torch.float32).contiguous()
    wf = torch.empty(kh * kw * co * ci, dtype=dtype, device=w.device)
    wd = torch.empty(9 * ci * co, dtype=dtype, device=w.device)
    _lib.call("mpu_conv2d_pack_weights", _lib.MPU_F32X3 if x3 else _dt(dtype), mode, _lib.ptr(w), ci, co,
              _lib.ptr(wf), _lib.ptr(wd), _lib.stream_ptr())
    return wf, wd


def _dtx(dtype, x3):
    if x3 and dtype != torch.float32:
        raise ValueError("x3 (split-bf16 products, MPU_F32X3) applies to f32 tensors")
    return _lib.MPU_F32X3 if x3 else _dt(dtype)


def conv2d(mode, x0, w_packed, cout, out_hw, bias=None, x1=None, mask=None, relu=False,
           w_tap_stride=None, w_row_stride=None, workspace=None, x3=False):
    """x0 [B,Hi,Wi,C0] (+ x1 concatenated on channels) -> [B,Ho,Wo,cout]. workspace: optional f32 scratch tensor
    (split-K partial sums of the deep-level schedules; 8*B*Ho*Wo*cout floats always suffice).
    x3: f32 tensors, every product as three bf16 MFMAs on hi + lo split operands (mpu_dtype MPU_F32X3)."""
    B = x0.shape[0]
    C0 = x0.shape[-1]
    C1 = 0 if x1 is None else x1.shape[-1]
    Ho, Wo = out_hw
    out = torch.empty((B, Ho, Wo, cout), dtype=x0.dtype, device=x0.device)
    if w_row_stride is None:
        w_row_stride = C0 + C1
    if w_tap_stride is None:
        w_tap_stride = cout * (C0 + C1)
    args = [_dtx(x0.dtype, x3), mode, _lib.ptr(x0), C0, _lib.ptr(x1), C1,
            _lib.ptr(w_packed), w_tap_stride, w_row_stride, _lib.ptr(bias), _lib.ptr(mask),
            _lib.ptr(out), B, Ho, Wo, cout, int(relu)]
    if workspace is None:
        _lib.call("mpu_conv2d_igemm", *args, _lib.stream_ptr())
    else:
        _lib.call("mpu_conv2d_igemm_ws", *args, _lib.ptr(workspace), workspace.numel(), _lib.stream_ptr())
    return out


def conv2d_wgrad_first_layer(x, n_image_channels, dz):
    """(dW [9, 8, Cout], db [Cout]) f32 of the first 3x3 conv: x [B,H,W,8] holds n_image_channels real channels."""
    B, H, W, cout = dz.shape
    assert x.shape[-1] == 8
    n = _lib.load().mpu_conv2d_wgrad_first_layer_workspace_floats(cout, B * H * W)
    ws = torch.empty(n, dtype=torch.float32, device=dz.device)
    dW = torch.empty((9, 8, cout), dtype=torch.float32, device=dz.device)
    db = torch.empty(cout, dtype=torch.float32, device=dz.device)
    _lib.call("mpu_conv2d_wgrad_first_layer", _dt(dz.dtype), _lib.ptr(x), n_image_channels, _lib.ptr(dz), cout,
              B, H, W, _lib.ptr(ws), _lib.ptr(dW), _lib.ptr(db), _lib.stream_ptr())
    return dW, db


def conv2d_wgrad(mode, x0, dz, x1=None, x3=False):
    """dW [taps, Cin, Cout] f32 of the conv whose input was concat(x0,x1) and output gradient dz (x3: as conv2d)."""
    B, Ho, Wo, cout = dz.shape
    C0 = x0.shape[-1]
    C1 = 0 if x1 is None else x1.shape[-1]
    nt = _NTAPS[mode]
    n = _lib.load().mpu_conv2d_wgrad_workspace_floats(mode, C0 + C1, cout, B * Ho * Wo)
    ws = torch.empty(n, dtype=torch.float32, device=dz.device)
    dW = torch.empty((nt, C0 + C1, cout), dtype=torch.float32, device=dz.device)
    _lib.call("mpu_conv2d_wgrad", _dtx(dz.dtype, x3), mode, _lib.ptr(x0), C0, _lib.ptr(x1), C1,
              _lib.ptr(dz), cout, B, Ho, Wo, _lib.ptr(ws), _lib.ptr(dW), _lib.stream_ptr())
    return dW
