"""
On-disk formats of a reference mpunet project, as OPTIONAL adapters: the accelerated path
itself only needs name-keyed arrays. NIfTI volumes are read and written natively (nifti.py: numpy + gzip, no nibabel);
Keras .h5 weight files degrade to a clear ImportError when h5py is not installed (it is absent from the build image;
the .npz mirror is what this build writes). Nothing on the hot path imports this module.

  Keras weights  .h5 ... model.save_weights / load_weights(by_name=True) of tf.keras 2.3
                         (mpunet/models/model_init.py:31,56, mpunet/callbacks/mcp_clean.py:57):
                         root attr `layer_names`; per layer a group with attr `weight_names`
                         ("<layer>/<var>:0") and one dataset per weight at <layer>/<layer>/<var>:0.
                         Kernels HWIO, BatchNormalization [gamma, beta, moving_mean, moving_variance].
  Volumes  .nii/.nii.gz  mpunet/image/image_pair.py:164-198 (get_fdata, affine), predictions
                         <id>_PRED.nii.gz (mpunet/bin/predict.py:90-117).
  Checkpoint names ..... "@epoch_{epoch:02d}_val_dice_{val_dice:.5f}.h5", best one chosen by get_best_model
                         (mpunet/utils/utils.py:88-110).
"""
import glob
import os
import re
import numpy as np

KERAS_VAR_ORDER = {"conv": ("kernel", "bias"), "bn": ("gamma", "beta", "moving_mean", "moving_variance")}


def keras_layer_names(depth=4):
    """Weight-carrying layers in the creation order of mpunet/models/unet.py:114-216; the
    unnamed 1x1 head gets Keras' auto-name `conv2d`."""
    names = []
    for i in range(depth):
        names += ["encoder_L%d_conv1" % i, "encoder_L%d_conv2" % i, "encoder_L%d_BN" % i]
    names += ["bottom_conv1", "bottom_conv2", "bottom_BN"]
    for i in range(depth):
        p = "upsample_L%d" % i
        names += [p + "_conv1", p + "_BN1", p + "_conv2", p + "_conv3", p + "_BN2"]
    return names + ["conv2d"]


def layer_weight_names(layer):
    kind = "bn" if "_BN" in layer else "conv"
    return ["%s/%s" % (layer, v) for v in KERAS_VAR_ORDER[kind]]


def h5_entries(weights, depth=4):
    """[(layer, [(keras weight name "<layer>/<var>:0", array), ...]), ...] for a {'<layer>/<var>': array} dict."""
    out = []
    for layer in keras_layer_names(depth):
        ws = [(n + ":0", np.asarray(weights[n])) for n in layer_weight_names(layer) if n in weights]
        if ws:
            out.append((layer, ws))
    return out


def _h5py():
    try:
        import h5py
        return h5py
    except ImportError as e:
        raise ImportError("Keras .h5 weight files need h5py (pip install h5py); without it use the name-keyed .npz "
                          "files this build writes, or convert on a machine that has h5py with "
                          "tools/convert_weights.py") from e


def save_keras_h5(path, weights, depth=4):
    """Write {'<layer>/<var>': array} as a tf.keras `save_weights` HDF5 file."""
    h5py = _h5py()
    entries = h5_entries(weights, depth)
    with h5py.File(path, "w") as f:
        f.attrs["layer_names"] = np.array([l.encode("utf8") for l, _ in entries])
        f.attrs["backend"] = b"tensorflow"
        f.attrs["keras_version"] = b"2.4.0"
        for layer, ws in entries:
            g = f.create_group(layer)
            g.attrs["weight_names"] = np.array([n.encode("utf8") for n, _ in ws])
            for n, arr in ws:
                g.create_dataset(n, data=np.asarray(arr, np.float32))


def load_keras_h5(path):
    """{'<layer>/<var>': array} from a tf.keras weight file (also the `model_weights` group of a full-model file)."""
    h5py = _h5py()
    out = {}
    with h5py.File(path, "r") as f:
        root = f["model_weights"] if "layer_names" not in f.attrs and "model_weights" in f else f
        for layer in root.attrs["layer_names"]:
            layer = layer.decode("utf8") if isinstance(layer, bytes) else str(layer)
            g = root[layer]
            for wn in g.attrs["weight_names"]:
                wn = wn.decode("utf8") if isinstance(wn, bytes) else str(wn)
                out[wn.rsplit(":", 1)[0]] = np.asarray(g[wn])
    return out


def load_nifti(path, dtype=np.float32):
    """(image [X,Y,Z,C] as `dtype`, affine) as ImagePair does (mpunet/image/image_pair.py:164-198): get_fdata with the
    header's intensity scaling, a channel axis added to 3-D data. Native reader (nifti.py), no nibabel."""
    from .nifti import read_nifti
    img, aff, _ = read_nifti(path, dtype=dtype)
    if img.ndim == 3:
        img = img[..., None]
    return img, aff


def load_nifti_labels(path, dtype=np.uint8):
    """labels_obj.get_fdata().astype(uint8) (image_pair.py:189-197); a trailing singleton axis is dropped."""
    from .nifti import read_nifti
    lab, _, _ = read_nifti(path, dtype=np.float64)
    lab = lab.astype(dtype)
    if lab.ndim == 4 and lab.shape[-1] == 1:
        lab = lab[..., 0]
    return lab


def save_nifti(path, volume, affine):
    """<id>_PRED.nii.gz (mpunet/bin/predict.py:90-117): nib.save(nib.Nifti1Image(volume, affine), path)."""
    from .nifti import write_nifti
    write_nifti(path, np.asarray(volume), np.asarray(affine, np.float64))


def get_best_model(model_dir, extensions=(".h5", ".npz")):
    """mpunet/utils/utils.py:88-110: patterns tried in order val_dice (max), val_loss (min), dice (max), loss (min),
    then model_weights.<ext>; the score is the first decimal number in the file name."""
    if len(os.listdir(model_dir)) == 0:
        raise OSError("Model dir {} is empty.".format(model_dir))
    patterns = [("@epoch*val_dice*", np.argmax), ("@epoch*val_loss*", np.argmin),
                ("@epoch*dice*", np.argmax), ("@epoch*loss*", np.argmin)]
    for pattern, select in patterns:
        models = [m for m in glob.glob(os.path.join(model_dir, pattern)) if m.endswith(tuple(extensions))]
        if models:
            scores = [float(re.findall(r"(\d+[.]\d+)", os.path.basename(m))[0]) for m in models]
            return os.path.abspath(models[int(select(np.array(scores)))])
    for ext in extensions:
        m = os.path.abspath(os.path.join(model_dir, "model_weights" + ext))
        if os.path.exists(m):
            return m
    raise OSError("Did not find any model files matching the patterns {} and did not find a model_weights file."
                  .format([p for p, _ in patterns]))
