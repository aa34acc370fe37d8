"""Convert U-Net weights between the reference's Keras .h5 files and this build's name-keyed .npz files.
usage: python tools/convert_weights.py IN.{h5,npz} OUT.{npz,h5} [--depth 4]      (the .h5 side goes through h5py, or without it the HDF5 C library -- multiplanarunet_amd/hdf5.py)"""
import argparse
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiplanarunet_amd import formats as F


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src"); ap.add_argument("dst"); ap.add_argument("--depth", type=int, default=4)
    a = ap.parse_args()
    if a.src.endswith((".h5", ".hdf5")):
        w = F.load_keras_h5(a.src)
    else:
        with np.load(a.src) as z:
            w = {k.replace("__", "/"): z[k] for k in z.files}
    missing = [n for l in F.keras_layer_names(a.depth) for n in F.layer_weight_names(l) if n not in w]
    if missing:
        print("warning: %d expected tensors missing, e.g. %s" % (len(missing), missing[:3]))
    if a.dst.endswith((".h5", ".hdf5")):
        F.save_keras_h5(a.dst, w, a.depth)
    else:
        np.savez(a.dst, **{k.replace("/", "__"): v for k, v in w.items()})
    print("wrote %s (%d tensors, %d parameters)" % (a.dst, len(w), sum(int(np.prod(v.shape)) for v in w.values())))


if __name__ == "__main__":
    main()
