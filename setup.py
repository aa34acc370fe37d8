"""Optional packaging of the MI355X hot path (`pip install -e . --no-deps --no-build-isolation` keeps the library in-tree,
where __graft_entry__.build() puts it). Installs the reference's console entry: `mp train | train_fusion | predict`."""
from setuptools import setup, find_packages

setup(
    name="multiplanarunet-amd",
    version="0.4.0",
    description="MI355X-native hot path of MultiPlanarUNet: per-plane U-Net step, plane resampling, multi-view fusion (HIP, gfx950)",
    packages=find_packages(include=["multiplanarunet_amd", "multiplanarunet_amd.*", "mpunet", "mpunet.*"]),
    package_data={"multiplanarunet_amd": ["lib/*.so", "csrc/*.hip", "csrc/*.h"]},
    python_requires=">=3.9",
    install_requires=["numpy", "torch", "pyyaml"],
    entry_points={"console_scripts": ["mp = mpunet.bin.mp:entry_func"]},
)
