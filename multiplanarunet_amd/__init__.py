"""
multiplanarunet_amd -- MI355X (gfx950) native hot path of perslev/MultiPlanarUNet:
per-plane 2-D U-Net forward/backward, predict-time plane resampling and
multi-view softmax fusion, behind the reference's UNet / FusionModel surface.
All arithmetic runs in hand-written HIP kernels (libmpunet_hip.so, C ABI in
include/mpunet_hip.h); torch is used for device memory, streams and
torch.distributed only.
"""
import os as _os

# multi-process GPU work on this platform needs dmabuf IPC (RCCL / device-tensor sharing fail with "hipIpcGetMemHandle:
# invalid argument" under the legacy mode); must be set before the HIP runtime initialises
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

__version__ = "0.1.0"
