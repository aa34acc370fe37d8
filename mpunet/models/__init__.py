from multiplanarunet_amd.unet import UNet
from multiplanarunet_amd.fusion_model import FusionModel
from .model_init import init_model, model_initializer

__all__ = ["UNet", "FusionModel", "init_model", "model_initializer"]
