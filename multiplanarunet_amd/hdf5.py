"""
The few HDF5 calls a tf.keras weight file needs, made on the HDF5 C library itself (libhdf5, through ctypes) for hosts that
have the library but not h5py -- this image is one: /opt/conda/lib/libhdf5.so 1.10.6, no h5py for the interpreter in use.
Files written here are written BY libhdf5, files read here are parsed BY libhdf5: nothing of the format is re-implemented.

Used by formats.save_keras_h5 / load_keras_h5 when `import h5py` fails. The library is looked for in
$MPU_LIBHDF5, the loader's search path ("hdf5", "hdf5_serial"), and the usual prefixes; `available()` tells.

Only what the Keras layout uses is bound: files, groups, float / integer datasets, string attributes (arrays and scalars,
fixed- or variable-length on read; fixed-length null-padded on write, which is what h5py stores for a list of bytes).
"""
import ctypes as C
import ctypes.util
import os
import sys
import numpy as np

_lib = None
_err = None

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5S_SCALAR = 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
H5T_STR_NULLPAD = 1
H5T_CSET_UTF8 = 1


class Hdf5Error(OSError):
    pass


def _candidates():
    env = os.environ.get("MPU_LIBHDF5")
    if env:
        yield env
    for name in ("hdf5", "hdf5_serial"):
        p = ctypes.util.find_library(name)
        if p:
            yield p
    for prefix in (sys.prefix, sys.base_prefix, "/opt/conda", "/usr", "/usr/local"):
        for sub in ("lib", "lib64", "lib/x86_64-linux-gnu", "lib/x86_64-linux-gnu/hdf5/serial"):
            for so in ("libhdf5.so", "libhdf5_serial.so"):
                yield os.path.join(prefix, sub, so)


def _load():
    global _lib, _err
    if _lib is not None or _err is not None:
        return _lib
    tried = []
    for p in _candidates():
        if "/" in p and not os.path.exists(p):
            continue
        try:
            L = C.CDLL(p)
            if L.H5open() < 0:
                raise OSError("H5open failed")
            _declare(L)                                               # AttributeError: a build without one of the calls below
        except (OSError, AttributeError) as e:                        #   -> this library is unusable, try the next
            tried.append("%s (%s)" % (p, e))
            continue
        L.path = p
        _lib = L
        return _lib
    _err = "no usable libhdf5 found" + (": " + "; ".join(tried) if tried else "")
    return None


def _declare(L):
    maj, mnr, rel = C.c_uint(), C.c_uint(), C.c_uint()
    L.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel))
    L.version = (maj.value, mnr.value, rel.value)
    hid = C.c_int64 if L.version >= (1, 10, 0) else C.c_int           # hid_t grew to 64 bits in 1.10
    L.hid = hid
    hs, sz, p = C.c_uint64, C.c_size_t, C.c_void_p
    sig = {
        "H5Fcreate": (hid, [C.c_char_p, C.c_uint, hid, hid]), "H5Fopen": (hid, [C.c_char_p, C.c_uint, hid]),
        "H5Fclose": (C.c_int, [hid]),
        "H5Gcreate2": (hid, [hid, C.c_char_p, hid, hid, hid]), "H5Gopen2": (hid, [hid, C.c_char_p, hid]),
        "H5Gclose": (C.c_int, [hid]),
        "H5Lexists": (C.c_int, [hid, C.c_char_p, hid]),
        "H5Screate": (hid, [C.c_int]), "H5Screate_simple": (hid, [C.c_int, C.POINTER(hs), C.POINTER(hs)]),
        "H5Sget_simple_extent_ndims": (C.c_int, [hid]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid, C.POINTER(hs), C.POINTER(hs)]),
        "H5Sget_simple_extent_npoints": (C.c_int64, [hid]), "H5Sclose": (C.c_int, [hid]),
        "H5Tcopy": (hid, [hid]), "H5Tset_size": (C.c_int, [hid, sz]), "H5Tget_size": (sz, [hid]),
        "H5Tset_strpad": (C.c_int, [hid, C.c_int]), "H5Tset_cset": (C.c_int, [hid, C.c_int]),
        "H5Tget_class": (C.c_int, [hid]), "H5Tis_variable_str": (C.c_int, [hid]), "H5Tget_sign": (C.c_int, [hid]),
        "H5Tclose": (C.c_int, [hid]),
        "H5Dcreate2": (hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]), "H5Dopen2": (hid, [hid, C.c_char_p, hid]),
        "H5Dget_space": (hid, [hid]), "H5Dget_type": (hid, [hid]),
        "H5Dread": (C.c_int, [hid, hid, hid, hid, hid, p]), "H5Dwrite": (C.c_int, [hid, hid, hid, hid, hid, p]),
        "H5Dclose": (C.c_int, [hid]),
        "H5Acreate2": (hid, [hid, C.c_char_p, hid, hid, hid, hid]), "H5Aopen": (hid, [hid, C.c_char_p, hid]),
        "H5Aexists": (C.c_int, [hid, C.c_char_p]), "H5Aget_type": (hid, [hid]), "H5Aget_space": (hid, [hid]),
        "H5Aread": (C.c_int, [hid, hid, p]), "H5Awrite": (C.c_int, [hid, hid, p]), "H5Aclose": (C.c_int, [hid]),
        "H5Eset_auto2": (C.c_int, [hid, p, p]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    # variable-length reclaim: H5Dvlen_reclaim is deprecated (absent from builds without deprecated symbols); 1.12+ has H5Treclaim
    L.vlen_reclaim = None
    for name in ("H5Dvlen_reclaim", "H5Treclaim"):
        f = getattr(L, name, None)
        if f is not None:
            f.restype, f.argtypes = C.c_int, [hid, hid, hid, p]
            L.vlen_reclaim = f
            break
    if L.vlen_reclaim is None:
        raise AttributeError("neither H5Dvlen_reclaim nor H5Treclaim")
    L.H5Eset_auto2(0, None, None)                                     # errors come back as return codes, not on stderr
    L.T = lambda name: hid.in_dll(L, name + "_g").value               # predefined datatype ids (valid after H5open)


def available():
    return _load() is not None


def library():
    L = _load()
    if L is None:
        raise ImportError(_err)
    return L


def _ok(v, what):
    if v < 0:
        raise Hdf5Error("HDF5: %s failed" % what)
    return v


class _Closing:
    """`with _Closing(id, closer) as id:` -- every HDF5 id opened here is closed on the way out."""
    def __init__(self, ident, closer, what):
        self.id, self.closer = _ok(ident, what), closer

    def __enter__(self):
        return self.id

    def __exit__(self, *exc):
        self.closer(self.id)
        return False


def open_file(path, mode="r"):
    L = library()
    b = os.fsencode(path)
    if mode == "r":
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        return _Closing(L.H5Fopen(b, H5F_ACC_RDONLY, 0), L.H5Fclose, "open %s" % path)
    if mode == "w":
        return _Closing(L.H5Fcreate(b, H5F_ACC_TRUNC, 0, 0), L.H5Fclose, "create %s" % path)
    raise ValueError("mode must be 'r' or 'w'")


def open_group(loc, name):
    L = library()
    return _Closing(L.H5Gopen2(loc, name.encode("utf8"), 0), L.H5Gclose, "open group %s" % name)


def create_group(loc, name):
    L = library()
    return _Closing(L.H5Gcreate2(loc, name.encode("utf8"), 0, 0, 0), L.H5Gclose, "create group %s" % name)


def has_link(loc, name):
    return library().H5Lexists(loc, name.encode("utf8"), 0) > 0


def has_attr(loc, name):
    return library().H5Aexists(loc, name.encode("utf8")) > 0


def write_strings_attr(loc, name, values, scalar=False):
    """A list of str / bytes as a 1-D array of fixed-length, null-padded strings (h5py's form for a numpy `S` array);
    scalar=True: one string on a scalar dataspace."""
    L = library()
    vals = [v if isinstance(v, bytes) else str(v).encode("utf8") for v in ([values] if scalar else values)]
    width = max([len(v) for v in vals] + [1])
    buf = b"".join(v.ljust(width, b"\0") for v in vals)
    with _Closing(L.H5Tcopy(L.T("H5T_C_S1")), L.H5Tclose, "string type") as t:
        _ok(L.H5Tset_size(t, width), "H5Tset_size")
        _ok(L.H5Tset_strpad(t, H5T_STR_NULLPAD), "H5Tset_strpad")
        n = (C.c_uint64 * 1)(len(vals))
        space = L.H5Screate(H5S_SCALAR) if scalar else L.H5Screate_simple(1, n, None)
        with _Closing(space, L.H5Sclose, "dataspace") as s:
            with _Closing(L.H5Acreate2(loc, name.encode("utf8"), t, s, 0, 0), L.H5Aclose, "create attribute %s" % name) as a:
                cbuf = C.create_string_buffer(buf, len(buf)) if buf else None
                if vals:
                    _ok(L.H5Awrite(a, t, cbuf), "write attribute %s" % name)


def read_strings_attr(loc, name):
    """A string attribute (array or scalar, fixed- or variable-length) as a list of str; an empty / non-string attribute
    (Keras writes `weight_names = []` as an empty float array for layers without weights) gives []."""
    L = library()
    with _Closing(L.H5Aopen(loc, name.encode("utf8"), 0), L.H5Aclose, "open attribute %s" % name) as a:
        with _Closing(L.H5Aget_type(a), L.H5Tclose, "attribute type") as t, \
                _Closing(L.H5Aget_space(a), L.H5Sclose, "attribute space") as s:
            n = int(L.H5Sget_simple_extent_npoints(s))
            if n <= 0 or L.H5Tget_class(t) != H5T_STRING:
                return []
            if L.H5Tis_variable_str(t) > 0:
                ptrs = (C.c_char_p * n)()
                _ok(L.H5Aread(a, t, ptrs), "read attribute %s" % name)
                out = [(ptrs[i] or b"").decode("utf8") for i in range(n)]
                L.vlen_reclaim(t, s, 0, ptrs)
                return out
            width = int(L.H5Tget_size(t))
            buf = C.create_string_buffer(n * width)
            _ok(L.H5Aread(a, t, buf), "read attribute %s" % name)
            raw = buf.raw
            return [raw[i * width:(i + 1) * width].split(b"\0", 1)[0].rstrip(b" ").decode("utf8") for i in range(n)]


def write_dataset(loc, name, array):
    """A float32 / float64 / int32 / int64 array as a contiguous little-endian dataset (h5py's create_dataset(data=...))."""
    L = library()
    arr = np.ascontiguousarray(array)
    types = {"f4": ("H5T_IEEE_F32LE", "H5T_NATIVE_FLOAT"), "f8": ("H5T_IEEE_F64LE", "H5T_NATIVE_DOUBLE"),
             "i4": ("H5T_STD_I32LE", "H5T_NATIVE_INT32"), "i8": ("H5T_STD_I64LE", "H5T_NATIVE_INT64")}
    key = arr.dtype.str[1:]
    if key not in types:
        raise Hdf5Error("write_dataset: dtype %s not supported" % arr.dtype)
    ftype, mtype = (L.T(n) for n in types[key])
    dims = (C.c_uint64 * max(arr.ndim, 1))(*arr.shape)
    space = L.H5Screate_simple(arr.ndim, dims, None) if arr.ndim else L.H5Screate(H5S_SCALAR)
    with _Closing(space, L.H5Sclose, "dataspace") as s:
        with _Closing(L.H5Dcreate2(loc, name.encode("utf8"), ftype, s, 0, 0, 0), L.H5Dclose, "create dataset %s" % name) as d:
            if arr.size:
                _ok(L.H5Dwrite(d, mtype, 0, 0, 0, arr.ctypes.data_as(C.c_void_p)), "write dataset %s" % name)


def read_dataset(loc, name):
    """A float or integer dataset as a numpy array (float32 / float64 / int64 by stored width; libhdf5 converts byte
    order, layout and filters)."""
    L = library()
    with _Closing(L.H5Dopen2(loc, name.encode("utf8"), 0), L.H5Dclose, "open dataset %s" % name) as d:
        with _Closing(L.H5Dget_space(d), L.H5Sclose, "dataset space") as s, _Closing(L.H5Dget_type(d), L.H5Tclose, "dataset type") as t:
            nd = _ok(L.H5Sget_simple_extent_ndims(s), "ndims")
            dims = (C.c_uint64 * max(nd, 1))()
            if nd:
                _ok(L.H5Sget_simple_extent_dims(s, dims, None), "dims")
            shape = tuple(int(dims[i]) for i in range(nd))
            cls, size = L.H5Tget_class(t), int(L.H5Tget_size(t))
            if cls == H5T_FLOAT:
                dt, mt = (np.float32, "H5T_NATIVE_FLOAT") if size <= 4 else (np.float64, "H5T_NATIVE_DOUBLE")
            elif cls == H5T_INTEGER:
                dt, mt = np.int64, "H5T_NATIVE_INT64"
            else:
                raise Hdf5Error("read_dataset %s: datatype class %d not supported" % (name, cls))
            out = np.empty(shape, dt)
            if out.size:
                _ok(L.H5Dread(d, L.T(mt), 0, 0, 0, out.ctypes.data_as(C.c_void_p)), "read dataset %s" % name)
            return out
