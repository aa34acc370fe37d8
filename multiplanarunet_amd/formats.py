"""
On-disk formats of a reference mpunet project, as OPTIONAL adapters: the accelerated path
itself only needs name-keyed arrays. NIfTI volumes are read and written natively (nifti.py: numpy + gzip, no nibabel);
Keras .h5 weight files go through h5py when it imports, else through the HDF5 C library itself (hdf5.py: libhdf5 via
ctypes -- this image has /opt/conda/lib/libhdf5.so but no h5py for its interpreter), else a clear ImportError; the .npz
mirror is what this build writes by default. Nothing on the hot path imports this module.

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
    out, seen = [], set()
    for layer in keras_layer_names(depth):
        ws = [(n + ":0", np.asarray(weights[n])) for n in layer_weight_names(layer) if n in weights]
        seen.update(n for n in layer_weight_names(layer))
        if ws:
            out.append((layer, ws))
    # layers outside the canonical list (an auto-named head `conv2d_<N>` of a converted reference checkpoint): kept, in name
    # order, variables in Keras' order
    extra = {}
    for n in weights:
        if n not in seen and "/" in n:
            extra.setdefault(n.split("/", 1)[0], []).append(n)
    order = {v: i for i, v in enumerate(KERAS_VAR_ORDER["conv"] + KERAS_VAR_ORDER["bn"])}
    for layer in sorted(extra):
        names = sorted(extra[layer], key=lambda n: order.get(n.split("/", 1)[1], 99))
        out.append((layer, [(n + ":0", np.asarray(weights[n])) for n in names]))
    return out


def _h5_backend():
    """"h5py" when it imports, else "libhdf5" (hdf5.py: the HDF5 C library through ctypes), else ImportError."""
    force = os.environ.get("MPU_H5_BACKEND", "")                      # "h5py" | "libhdf5": no fall-through (tests)
    if force not in ("", "h5py", "libhdf5"):
        raise ValueError("MPU_H5_BACKEND must be h5py or libhdf5")
    if force != "libhdf5":
        try:
            import h5py  # noqa: F401
            return "h5py"
        except ImportError:
            if force == "h5py":
                raise
    from . import hdf5
    if hdf5.available():
        return "libhdf5"
    raise ImportError("Keras .h5 weight files need h5py (pip install h5py) or the HDF5 C library (libhdf5.so on the "
                      "loader path, or MPU_LIBHDF5=/path/to/libhdf5.so); without either use the name-keyed .npz "
                      "files this build writes, or convert elsewhere with tools/convert_weights.py")


KERAS_ATTR_LIMIT = 64512       # HDF5_OBJECT_HEADER_LIMIT of keras/saving/hdf5_format.py: longer name lists are split into <name>0, <name>1, ...


def _name_chunks(name, values):
    """save_attributes_to_hdf5_group: one attribute, or `name0`, `name1`, ... when the list exceeds the header limit."""
    vals = [v.encode("utf8") for v in values]
    width = max([len(v) for v in vals] + [1])
    if width * len(vals) <= KERAS_ATTR_LIMIT:
        return [(name, vals)]
    per = max(1, KERAS_ATTR_LIMIT // width)
    return [("%s%d" % (name, i), vals[k:k + per]) for i, k in enumerate(range(0, len(vals), per))]


def save_keras_h5(path, weights, depth=4):
    """Write {'<layer>/<var>': array} as a tf.keras `save_weights` HDF5 file (keras/saving/hdf5_format.py
    save_weights_to_hdf5_group: root attrs layer_names / backend / keras_version, one group per layer with attr
    weight_names and one dataset per weight at <layer>/<layer>/<var>:0)."""
    backend = _h5_backend()
    entries = h5_entries(weights, depth)
    if backend == "h5py":
        import h5py
        with h5py.File(path, "w") as f:
            for n, vals in _name_chunks("layer_names", [l for l, _ in entries]):
                f.attrs[n] = np.array(vals)
            f.attrs["backend"] = b"tensorflow"
            f.attrs["keras_version"] = b"2.4.0"
            for layer, ws in entries:
                g = f.create_group(layer)
                for n, vals in _name_chunks("weight_names", [n for n, _ in ws]):
                    g.attrs[n] = np.array(vals)
                for n, arr in ws:
                    g.create_dataset(n, data=np.asarray(arr, np.float32))
        return
    from . import hdf5 as H
    with H.open_file(path, "w") as f:
        for n, vals in _name_chunks("layer_names", [l for l, _ in entries]):
            H.write_strings_attr(f, n, vals)
        H.write_strings_attr(f, "backend", b"tensorflow", scalar=True)
        H.write_strings_attr(f, "keras_version", b"2.4.0", scalar=True)
        for layer, ws in entries:
            with H.create_group(f, layer) as g:
                for n, vals in _name_chunks("weight_names", [n for n, _ in ws]):
                    H.write_strings_attr(g, n, vals)
                made = set()
                for n, arr in ws:                                     # "<layer>/<var>:0": the dataset sits in a nested group
                    parts = n.split("/")
                    for k in range(1, len(parts)):
                        sub = "/".join(parts[:k])
                        if sub not in made:
                            with H.create_group(g, sub):
                                pass
                            made.add(sub)
                    H.write_dataset(g, n, np.asarray(arr, np.float32))


def _chunked_names(read, has, name):
    """load_attributes_from_hdf5_group: `name`, or the concatenation of `name0`, `name1`, ..."""
    if has(name):
        return list(read(name))
    out, i = [], 0
    while has("%s%d" % (name, i)):
        out += list(read("%s%d" % (name, i)))
        i += 1
    return out


def load_keras_h5(path):
    """{'<layer>/<var>': array} from a tf.keras weight file (also the `model_weights` group of a full-model file)."""
    backend = _h5_backend()
    out = {}
    dec = lambda v: v.decode("utf8") if isinstance(v, bytes) else str(v)          # noqa: E731
    if backend == "h5py":
        import h5py
        with h5py.File(path, "r") as f:
            root = f["model_weights"] if "layer_names" not in f.attrs and "layer_names0" not in f.attrs and "model_weights" in f else f
            for layer in _chunked_names(lambda n: root.attrs[n], lambda n: n in root.attrs, "layer_names"):
                g = root[dec(layer)]
                for wn in _chunked_names(lambda n: g.attrs[n], lambda n: n in g.attrs, "weight_names"):
                    out[dec(wn).rsplit(":", 1)[0]] = np.asarray(g[dec(wn)])
        return out
    from . import hdf5 as H
    with H.open_file(path, "r") as f:
        def read_layers(root):
            for layer in _chunked_names(lambda n: H.read_strings_attr(root, n), lambda n: H.has_attr(root, n), "layer_names"):
                with H.open_group(root, layer) as g:
                    for wn in _chunked_names(lambda n: H.read_strings_attr(g, n), lambda n: H.has_attr(g, n), "weight_names"):
                        out[wn.rsplit(":", 1)[0]] = H.read_dataset(g, wn)
        if not H.has_attr(f, "layer_names") and not H.has_attr(f, "layer_names0") and H.has_link(f, "model_weights"):
            with H.open_group(f, "model_weights") as root:
                read_layers(root)
        else:
            read_layers(f)
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
