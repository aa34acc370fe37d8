"""
multiplanarunet_amd -- MI355X (gfx950) native hot path of perslev/MultiPlanarUNet:
per-plane 2-D U-Net forward/backward, predict-time plane resampling and
multi-view softmax fusion, behind the reference's UNet / FusionModel surface.
All arithmetic runs in hand-written HIP kernels (libmpunet_hip.so, C ABI in
include/mpunet_hip.h); torch is used for device memory, streams and
torch.distributed only.
"""
__version__ = "0.1.0"
