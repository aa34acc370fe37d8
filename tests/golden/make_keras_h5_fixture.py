"""Writes tests/golden/keras_unet_d1.{h5,full.h5,npz} with the REAL h5py (run with an interpreter that has it; in the build
image: /opt/conda/bin/python3.9 tests/golden/make_keras_h5_fixture.py). numpy + h5py only.

The .h5 files follow tf.keras 2.3's save_weights_to_hdf5_group step by step (keras/saving/hdf5_format.py), for the layer
list of the reference's UNet at depth 1 (mpunet/models/unet.py:114-216; names of the weightless layers included, as Keras
lists every layer): root attrs layer_names (model order) / backend / keras_version, groups created in SORTED layer-name
order, per group the attr weight_names ([] for weightless layers) and one dataset per weight named "<layer>/<var>:0",
created empty and then assigned.
  keras_unet_d1.h5 ....... what `model.save_weights(path)` writes (the reference's checkpoints, mcp_clean.py:57); the head
                           is auto-named conv2d_7 as in a process that had built other models before.
  keras_unet_d1.full.h5 .. what `model.save(path)` writes around it: the same content under /model_weights, with str
                           (variable-length) attributes as older h5py versions stored them.
  keras_unet_d1.npz ...... the arrays, keyed "<layer>/<var>".
"""
import os
import numpy as np
import h5py

F0, F1, K = 4, 8, 3                      # complexity_factor = 1/256: 64 -> 4 filters
rng = np.random.RandomState(20260927)


def conv(name, kh, cin, cout):
    return name, [("kernel:0", rng.randn(kh, kh, cin, cout).astype(np.float32) * 0.1), ("bias:0", rng.randn(cout).astype(np.float32) * 0.1)]


def bn(name, c):
    return name, [("gamma:0", 1 + 0.1 * rng.randn(c).astype(np.float32)), ("beta:0", 0.1 * rng.randn(c).astype(np.float32)),
                  ("moving_mean:0", 0.1 * rng.randn(c).astype(np.float32)), ("moving_variance:0", (1 + 0.1 * rng.rand(c)).astype(np.float32))]


HEAD = "conv2d_7"
LAYERS = [("input_1", []),
          conv("encoder_L0_conv1", 3, 1, F0), conv("encoder_L0_conv2", 3, F0, F0), bn("encoder_L0_BN", F0), ("encoder_L0_pool", []),
          conv("bottom_conv1", 3, F0, F1), conv("bottom_conv2", 3, F1, F1), bn("bottom_BN", F1),
          ("upsample_L0_up", []), conv("upsample_L0_conv1", 2, F1, F0), bn("upsample_L0_BN1", F0), ("upsample_L0_concat", []),
          conv("upsample_L0_conv2", 3, 2 * F0, F0), conv("upsample_L0_conv3", 3, F0, F0), bn("upsample_L0_BN2", F0),
          conv(HEAD, 1, F0, K), ("flatten_output", [])]


def save_weights_to_hdf5_group(f, vlen_str):
    enc = (lambda s: s) if vlen_str else (lambda s: s.encode("utf8"))
    f.attrs["layer_names"] = [enc(n) for n, _ in LAYERS]
    f.attrs["backend"] = enc("tensorflow")
    f.attrs["keras_version"] = enc("2.4.0")
    for name, ws in sorted(LAYERS, key=lambda l: l[0]):
        g = f.create_group(name)
        g.attrs["weight_names"] = [enc("%s/%s" % (name, v)) for v, _ in ws]
        for v, val in ws:
            d = g.create_dataset("%s/%s" % (name, v), val.shape, dtype=val.dtype)
            d[:] = val


here = os.path.dirname(os.path.abspath(__file__))
with h5py.File(os.path.join(here, "keras_unet_d1.h5"), "w") as f:
    save_weights_to_hdf5_group(f, vlen_str=False)
with h5py.File(os.path.join(here, "keras_unet_d1.full.h5"), "w") as f:
    f.attrs["keras_version"] = "2.4.0"
    f.attrs["backend"] = "tensorflow"
    f.attrs["model_config"] = '{"class_name": "Functional"}'
    save_weights_to_hdf5_group(f.create_group("model_weights"), vlen_str=True)
np.savez(os.path.join(here, "keras_unet_d1.npz"),
         **{"%s/%s" % (n, v.split(":")[0]): a for n, ws in LAYERS for v, a in ws})
print("h5py", h5py.__version__, "HDF5", h5py.version.hdf5_version)
