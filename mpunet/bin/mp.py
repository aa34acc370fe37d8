"""Console entry `mp` of the reference (setup.py:30-33 -> mpunet.bin.mp:entry_func)."""
from multiplanarunet_amd.cli.mp import entry_func  # noqa: F401

if __name__ == "__main__":
    entry_func()
