"""
TEST INFRASTRUCTURE ONLY -- import shim that lets the *unmodified* NumPy-only
functions of the reference (mpunet 0.2.12, /root/reference) be imported in a
container that lacks tensorflow / nibabel / h5py / ruamel.yaml.

Only `oracle/gen_golden.py` (run by hand in the build container, where
/root/reference exists) uses this. Nothing shipped, nothing on the GPU box and
no test imports it at run time: the goldens it produces are committed under
tests/golden/ as plain .npz data.

What it does:
  1. a sys.meta_path finder that fabricates permissive dummy modules for
     tensorflow*, nibabel, ruamel, h5py, keras, matplotlib-free imports;
  2. aliases scipy.interpolate.interpnd._ndim_coords_from_arrays
     (used at mpunet/interpolation/regular_grid_interpolator.py:3);
  3. restores the removed NumPy aliases np.int / np.bool / np.float
     (mpunet/evaluate/metrics.py:18, mpunet/models/unet.py:226).
Anything that *executes* a mocked object (UNet, FusionModel) is NOT an oracle.
"""
import sys
import types
import importlib.abc
import importlib.machinery

REFERENCE_ROOT = "/root/reference"
_MOCKED = {"tensorflow", "tensorflow_addons", "nibabel", "ruamel", "h5py", "keras"}


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Dummy()


class _MockModule(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        cls = type(name, (_Dummy,), {})
        setattr(self, name, cls)
        return cls


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _MOCKED:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _MockModule(spec.name)

    def exec_module(self, module):
        pass


def install():
    import numpy as np
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder())
    import scipy.interpolate.interpnd as ip
    from scipy.interpolate._interpnd import _ndim_coords_from_arrays
    ip._ndim_coords_from_arrays = _ndim_coords_from_arrays
    for alias, typ in (("int", int), ("bool", bool), ("float", float)):
        if not hasattr(np, alias):
            setattr(np, alias, typ)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    sys.dont_write_bytecode = True
