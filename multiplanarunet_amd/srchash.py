"""sha256 (16 hex digits) over a set of kernel sources: PMC summaries under profiles/ carry the hash of the sources they were
measured on, and bench.py reports their HBM bytes only while the hash still matches (VERDICT r3 hygiene: a kernel change
without a new counter pass used to report stale `roofline.traffic`)."""
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
CONV_SOURCES = ("common.h", "kernels.h", "env.h", "reduce.h", "conv_c8.hip", "conv_glds.hip", "conv_halo.hip", "conv_halo16.hip", "conv_igemm.hip",
                "conv_ws.hip", "wgrad_c8.hip", "wgrad_taps.hip", "unet_model.hip", "unet_ops.hip")
GEOMETRY_SOURCES = ("common.h", "geometry.hip")


def source_sha16(names):
    h = hashlib.sha256()
    for n in sorted(names):
        with open(os.path.join(CSRC, n), "rb") as f:
            h.update(n.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]
