"""Content hashes of the kernel sources (sha256, 16 hex digits).

* `source_sha16(names)`: PMC summaries under profiles/ carry the hash of the sources they were measured on, and bench.py reports
  their HBM bytes only while the hash still matches (VERDICT r3 hygiene: a kernel change without a new counter pass used to
  report stale `roofline.traffic`).
* `build_sha16()`: the hash of EVERYTHING libmpunet_hip.so is built from (every csrc/*.hip and *.h, include/mpunet_hip.h and the
  compiler flags). build.py compiles it into the library (`mpu_build_hash()`), `_lib.load()` refuses a library whose hash is not
  that of the sources next to it (VERDICT r4 hygiene: `build()` compared mtimes only, so a stale binary could ship unnoticed).
"""
import hashlib
import os

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HEADER = os.path.join(HERE, "..", "include", "mpunet_hip.h")
CONV_SOURCES = ("common.h", "kernels.h", "env.h", "reduce.h", "conv_c8.hip", "conv_deepk.hip", "conv_glds.hip", "conv_halo.hip", "conv_halo16.hip", "conv_igemm.hip",
                "conv_ws.hip", "wgrad_c8.hip", "wgrad_taps.hip", "unet_model.hip", "unet_ops.hip")
GEOMETRY_SOURCES = ("common.h", "geometry.hip")


def source_sha16(names):
    h = hashlib.sha256()
    for n in sorted(names):
        with open(os.path.join(CSRC, n), "rb") as f:
            h.update(n.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


def sources_present():
    return os.path.isdir(CSRC) and os.path.exists(HEADER)


def _file_digest(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).digest()


def header_digest():
    """One digest over every header a translation unit may include."""
    h = hashlib.sha256()
    for n in sorted(f for f in os.listdir(CSRC) if f.endswith(".h")):
        h.update(n.encode() + b"\0" + _file_digest(os.path.join(CSRC, n)))
    h.update(b"mpunet_hip.h\0" + _file_digest(HEADER))
    return h.digest()


def unit_sha16(src, flags, hdr=None):
    """Hash of what ONE object file is built from: its source, every header, its flags."""
    h = hashlib.sha256()
    h.update(src.encode() + b"\0" + _file_digest(os.path.join(CSRC, src)))
    h.update(hdr if hdr is not None else header_digest())
    h.update("\0".join(flags).encode())
    return h.hexdigest()[:16]


def build_sha16(flags_of):
    """flags_of(src) -> list of compiler flags. Hash over all translation units."""
    hdr = header_digest()
    h = hashlib.sha256()
    for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
        h.update(unit_sha16(src, flags_of(src), hdr).encode())
    return h.hexdigest()[:16]
