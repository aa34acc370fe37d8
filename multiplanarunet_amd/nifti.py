"""
NIfTI-1 single-file volumes (.nii / .nii.gz) read and written with numpy + gzip only -- the image has no nibabel, and the
reference's project folders hold nothing else.

What the reference does with nibabel, and what is restated here:
  nib.load(path).get_fdata(caching="unchanged", dtype=float32), .affine ... mpunet/image/image_pair.py:81-84,164-187
  labels: get_fdata().astype(uint8) ................................... mpunet/image/image_pair.py:189-197
  nib.save(nib.Nifti1Image(pred, affine=image_pair.affine), "<id>_PRED.nii.gz") ... mpunet/bin/predict.py:90-117
  nib.Nifti1Header.quaternion_threshold = -1e-6 (lenient quaternions) ............. mpunet/image/image_pair.py:24

Header layout: the NIfTI-1.1 standard (nifti1.h), 348 bytes + a 4-byte extension flag, data at vox_offset in Fortran order.
Conventions that are nibabel's rather than the standard's are marked "(nibabel)": the affine preference sform > qform >
base affine, the x-flipped base affine, NaN scl_slope / scl_inter on write, sform code 2 + qform code 0 for an image made
from (data, affine). Not supported (clear errors): NIfTI-2, .hdr/.img pairs, RGB / complex data, header extensions are
skipped on read and never written.
"""
import gzip
import struct
import numpy as np

# datatype code -> numpy dtype (nifti1.h DT_*)
_DTYPES = {2: "u1", 4: "i2", 8: "i4", 16: "f4", 64: "f8", 256: "i1", 512: "u2", 768: "u4", 1024: "i8", 1280: "u8"}
_CODES = {np.dtype(v).str[1:]: k for k, v in _DTYPES.items()}
QUATERNION_THRESHOLD = -1e-6          # image_pair.py:24


class NiftiError(ValueError):
    pass


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


def quat2mat(q):
    """Rotation matrix of a quaternion (w, x, y, z); identity for a (near-)zero quaternion."""
    w, x, y, z = [float(v) for v in q]
    n = w * w + x * x + y * y + z * z
    if n < np.finfo(np.float64).eps:
        return np.eye(3)
    s = 2.0 / n
    X, Y, Z = x * s, y * s, z * s
    wX, wY, wZ = w * X, w * Y, w * Z
    xX, xY, xZ = x * X, x * Y, x * Z
    yY, yZ, zZ = y * Y, y * Z, z * Z
    return np.array([[1.0 - (yY + zZ), xY - wZ, xZ + wY],
                     [xY + wZ, 1.0 - (xX + zZ), yZ - wX],
                     [xZ - wY, yZ + wX, 1.0 - (xX + yY)]])


def mat2quat(M):
    """Quaternion (w >= 0) of a rotation matrix: the principal eigenvector of the symmetric 4x4 form (robust to a matrix
    that is only nearly orthogonal)."""
    Qxx, Qyx, Qzx, Qxy, Qyy, Qzy, Qxz, Qyz, Qzz = np.asarray(M, np.float64).flat
    K = np.array([[Qxx - Qyy - Qzz, 0, 0, 0],
                  [Qyx + Qxy, Qyy - Qxx - Qzz, 0, 0],
                  [Qzx + Qxz, Qzy + Qyz, Qzz - Qxx - Qyy, 0],
                  [Qyz - Qzy, Qzx - Qxz, Qxy - Qyx, Qxx + Qyy + Qzz]]) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], int(np.argmax(vals))]
    return -q if q[0] < 0 else q


class Header(dict):
    """The fields the path reads, by their nifti1.h names."""


def _parse_header(raw):
    if len(raw) < 348:
        raise NiftiError("not a NIfTI-1 file: %d header bytes" % len(raw))
    for e in ("<", ">"):
        if struct.unpack(e + "i", raw[:4])[0] == 348:
            break
    else:
        if struct.unpack("<i", raw[:4])[0] == 540 or struct.unpack(">i", raw[:4])[0] == 540:
            raise NiftiError("NIfTI-2 files are not supported")
        raise NiftiError("not a NIfTI-1 file (sizeof_hdr != 348)")
    magic = raw[344:348]
    if magic[:3] == b"ni1":
        raise NiftiError(".hdr/.img pairs are not supported (convert to a single .nii file)")
    if magic[:3] != b"n+1":
        raise NiftiError("not a NIfTI-1 file (magic %r)" % magic)
    h = Header(endian=e)
    h["dim"] = struct.unpack(e + "8h", raw[40:56])
    h["datatype"], h["bitpix"] = struct.unpack(e + "2h", raw[70:74])
    h["pixdim"] = struct.unpack(e + "8f", raw[76:108])
    h["vox_offset"], h["scl_slope"], h["scl_inter"] = struct.unpack(e + "3f", raw[108:120])
    h["xyzt_units"] = raw[123]
    h["qform_code"], h["sform_code"] = struct.unpack(e + "2h", raw[252:256])
    h["quatern"] = struct.unpack(e + "3f", raw[256:268])
    h["qoffset"] = struct.unpack(e + "3f", raw[268:280])
    h["srow"] = np.array(struct.unpack(e + "12f", raw[280:328]), np.float64).reshape(3, 4)
    nd = h["dim"][0]
    if not 1 <= nd <= 7:
        raise NiftiError("bad dim[0] = %d" % nd)
    h["shape"] = tuple(int(d) for d in h["dim"][1:1 + nd])
    if any(d < 1 for d in h["shape"]):
        raise NiftiError("bad dim %r" % (h["dim"],))
    if h["datatype"] not in _DTYPES:
        raise NiftiError("unsupported datatype code %d (RGB / complex / float128 data)" % h["datatype"])
    return h


def qform_affine(h):
    """NIfTI-1 method 2: rotation from the quaternion (b, c, d), voxel sizes pixdim[1..3], qfac = pixdim[0] on the third
    axis, translation qoffset."""
    b, c, d = [float(v) for v in h["quatern"]]
    a2 = 1.0 - (b * b + c * c + d * d)
    if a2 < QUATERNION_THRESHOLD:
        raise NiftiError("quaternion (b, c, d) = (%g, %g, %g) is longer than 1" % (b, c, d))
    a = np.sqrt(max(a2, 0.0))
    R = quat2mat((a, b, c, d))
    vox = np.array(h["pixdim"][1:4], np.float64)
    if np.any(vox < 0):
        raise NiftiError("pixdim[1..3] must be positive")
    qfac = -1.0 if h["pixdim"][0] == -1 else 1.0          # (anything else counts as +1, as the standard says)
    vox[2] *= qfac
    A = np.eye(4)
    A[:3, :3] = R @ np.diag(vox)
    A[:3, 3] = h["qoffset"]
    return A


def base_affine(h):
    """(nibabel) no sform, no qform: voxel sizes on the diagonal, x flipped, the centre voxel at the origin."""
    shape = np.array((h["shape"] + (1, 1, 1))[:3], np.float64)
    z = np.array(h["pixdim"][1:4], np.float64)
    z[0] *= -1
    A = np.eye(4)
    A[:3, :3] = np.diag(z)
    A[:3, 3] = -((shape - 1) / 2.0) * z
    return A


def best_affine(h):
    """(nibabel) sform when its code is set, else qform when its code is set, else the base affine."""
    if h["sform_code"] != 0:
        A = np.eye(4)
        A[:3] = h["srow"]
        return A
    if h["qform_code"] != 0:
        return qform_affine(h)
    return base_affine(h)


def read_nifti(path, dtype=np.float32, scaled=True):
    """(data, affine 4x4 f64, header). `data` has the stored shape; with `scaled` the stored values go through
    scl_slope / scl_inter (when the slope is finite and non-zero) and are returned as `dtype` -- get_fdata(dtype=...)."""
    with _open(path, "rb") as f:
        raw = f.read(348)
        h = _parse_header(raw)
        off = int(h["vox_offset"])
        if off < 352:
            off = 352
        f.read(off - 348)                                         # extension flag + extensions
        dt = np.dtype(h["endian"] + _DTYPES[h["datatype"]])
        n = int(np.prod(h["shape"], dtype=np.int64))
        buf = f.read(n * dt.itemsize)
    if len(buf) != n * dt.itemsize:
        raise NiftiError("%s: truncated data (%d of %d bytes)" % (path, len(buf), n * dt.itemsize))
    data = np.frombuffer(buf, dt).reshape(h["shape"], order="F")
    if scaled:
        slope, inter = float(h["scl_slope"]), float(h["scl_inter"])
        if np.isfinite(slope) and slope != 0.0 and not (slope == 1.0 and (inter == 0.0 or not np.isfinite(inter))):
            data = data.astype(np.float64) * slope + (inter if np.isfinite(inter) else 0.0)
        data = np.array(data, dtype=dtype, order="C", copy=True)      # (own, writable memory: the file buffer is read-only)
    else:
        data = np.array(data, dtype=dt.newbyteorder("="), order="C", copy=True)
    return data, best_affine(h), h


def _qform_params(affine):
    """(quaternion b, c, d, qfac, voxel sizes) of an affine, as set_qform derives them: zooms = column lengths, a
    left-handed rotation part flips the third axis (qfac = -1), the nearest orthogonal matrix (polar part) gives the
    quaternion."""
    RZS = np.asarray(affine, np.float64)[:3, :3]
    zooms = np.sqrt(np.sum(RZS * RZS, axis=0))
    zooms[zooms == 0] = 1.0
    R = RZS / zooms
    qfac = 1.0
    if np.linalg.det(R) < 0:
        R = R.copy()
        R[:, 2] *= -1
        qfac = -1.0
    P, _, Qs = np.linalg.svd(R)
    q = mat2quat(P @ Qs)
    return q[1:], qfac, zooms


def write_nifti(path, data, affine):
    """nib.save(nib.Nifti1Image(data, affine), path): sform code 2 (aligned) carrying the affine, qform parameters from
    the same affine with code 0 (nibabel), no intensity scaling, little endian."""
    data = np.asarray(data)
    if data.dtype == np.bool_:
        data = data.astype(np.uint8)
    key = data.dtype.newbyteorder("<").str[1:]
    if key not in _CODES:
        raise NiftiError("cannot store dtype %s in a NIfTI-1 file" % data.dtype)
    if not 1 <= data.ndim <= 7:
        raise NiftiError("NIfTI-1 holds 1 to 7 dimensions, got %d" % data.ndim)
    affine = np.asarray(affine, np.float64)
    if affine.shape != (4, 4):
        raise NiftiError("affine must be 4x4")
    (qb, qc, qd), qfac, zooms = _qform_params(affine)
    dim = [data.ndim] + list(data.shape) + [1] * (7 - data.ndim)
    pixdim = [qfac] + list(zooms) + [1.0] * 4
    if data.ndim < 3:
        pixdim[1 + data.ndim:4] = [1.0] * (3 - data.ndim)
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    hdr[38:39] = b"r"
    struct.pack_into("<8h", hdr, 40, *dim)
    struct.pack_into("<2h", hdr, 70, _CODES[key], data.dtype.itemsize * 8)
    struct.pack_into("<8f", hdr, 76, *pixdim)
    struct.pack_into("<3f", hdr, 108, 352.0, float("nan"), float("nan"))
    struct.pack_into("<2h", hdr, 252, 0, 2)
    struct.pack_into("<3f", hdr, 256, qb, qc, qd)
    struct.pack_into("<3f", hdr, 268, *affine[:3, 3])
    struct.pack_into("<12f", hdr, 280, *affine[:3].ravel())
    hdr[344:348] = b"n+1\0"
    payload = np.asfortranarray(data.astype(data.dtype.newbyteorder("<"), copy=False)).tobytes(order="F")
    with (gzip.open(path, "wb", compresslevel=1) if str(path).endswith(".gz") else open(path, "wb")) as f:
        f.write(bytes(hdr))
        f.write(b"\0\0\0\0")
        f.write(payload)


def volume_identifier(path):
    """ImagePair's identifier (image_pair.py:131-135): the file name up to its FIRST dot for NIfTI files
    ("sub-01.T1.nii.gz" -> "sub-01"); other volume files (.npz) keep everything before the extension."""
    import os
    name = os.path.basename(str(path))
    if name.endswith((".nii", ".nii.gz")):
        return name.split(".")[0]
    return os.path.splitext(name)[0]
