"""
Drop-in import paths of the reference (`mpunet.models.UNet`, `mpunet.models.FusionModel`,
SURVEY.md section 8b) re-exporting the MI355X implementation in multiplanarunet_amd.
Newly written skeleton: nothing here is copied from perslev/MultiPlanarUNet.
"""
__version__ = "0.2.12+mi355x"
