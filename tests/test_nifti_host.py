"""Native NIfTI-1 reader / writer (multiplanarunet_amd/nifti.py) against headers built here field by
field from the NIfTI-1.1 standard (nifti1.h offsets), so reader and test do not share a layout table: sform / qform / no
form affines, qfac = -1, intensity scaling, big-endian files, 4-D data, Fortran order, gzip; the writer through the
reader and through an independent parse of its bytes; the volume loaders on a project-style folder."""
import gzip
import os
import struct
import numpy as np
import pytest
from multiplanarunet_amd import nifti as N
from multiplanarunet_amd import formats as F


def _header(shape, code, bitpix, e="<", pixdim=(1, 1, 1, 1, 1, 1, 1, 1), slope=0.0, inter=0.0, qform=0, sform=0,
            quat=(0, 0, 0), qoff=(0, 0, 0), srow=None, vox_offset=352.0, magic=b"n+1\0"):
    h = bytearray(348)
    h[0:4] = struct.pack(e + "i", 348)                         # sizeof_hdr
    dim = [len(shape)] + list(shape) + [1] * (7 - len(shape))
    h[40:56] = struct.pack(e + "8h", *dim)                     # dim[8]
    h[70:72] = struct.pack(e + "h", code)                      # datatype
    h[72:74] = struct.pack(e + "h", bitpix)                    # bitpix
    h[76:108] = struct.pack(e + "8f", *pixdim)                 # pixdim[8]
    h[108:112] = struct.pack(e + "f", vox_offset)              # vox_offset
    h[112:116] = struct.pack(e + "f", slope)                   # scl_slope
    h[116:120] = struct.pack(e + "f", inter)                   # scl_inter
    h[252:254] = struct.pack(e + "h", qform)                   # qform_code
    h[254:256] = struct.pack(e + "h", sform)                   # sform_code
    h[256:268] = struct.pack(e + "3f", *quat)                  # quatern_b, _c, _d
    h[268:280] = struct.pack(e + "3f", *qoff)                  # qoffset_x, _y, _z
    if srow is not None:
        h[280:328] = struct.pack(e + "12f", *np.asarray(srow, np.float64).ravel())   # srow_x, _y, _z
    h[344:348] = magic
    return bytes(h)


def _write(path, header, data, e="<", pad=4):
    raw = header + b"\0" * pad + np.asarray(data).astype(np.asarray(data).dtype.newbyteorder(e)).tobytes(order="F")
    if str(path).endswith(".gz"):
        with gzip.open(path, "wb") as f:
            f.write(raw)
    else:
        with open(path, "wb") as f:
            f.write(raw)


def _ramp(shape, dtype):
    idx = np.indices(shape)
    return sum(idx[k] * 10 ** k for k in range(len(shape))).astype(dtype)      # data[i,j,k] = i + 10 j + 100 k


def test_sform_int16_fortran_order_and_scaling(tmp_path):
    data = _ramp((4, 3, 5), np.int16)
    srow = np.array([[0.9, 0.1, 0.0, -12.5], [-0.1, 1.1, 0.05, 7.25], [0.0, 0.2, 2.0, 30.0]])
    p = str(tmp_path / "a.nii")
    _write(p, _header((4, 3, 5), 4, 16, sform=1, qform=1, quat=(0.5, 0.5, 0.5), srow=srow, slope=0.5, inter=-3.0), data)
    out, aff, h = N.read_nifti(p)
    assert out.dtype == np.float32 and out.shape == (4, 3, 5)
    np.testing.assert_array_equal(out, data.astype(np.float64) * 0.5 - 3.0)
    assert out[2, 1, 3] == (2 + 10 + 300) * 0.5 - 3.0                          # first index fastest on disk
    np.testing.assert_array_equal(aff[:3], srow.astype(np.float32).astype(np.float64))   # sform wins over qform
    np.testing.assert_array_equal(aff[3], [0, 0, 0, 1])
    raw, _, _ = N.read_nifti(p, scaled=False)
    assert raw.dtype == np.int16
    np.testing.assert_array_equal(raw, data)


@pytest.mark.parametrize("slope,inter", [(0.0, 5.0), (float("nan"), float("nan")), (1.0, 0.0)])
def test_no_scaling_when_slope_is_zero_or_nan_or_identity(tmp_path, slope, inter):
    data = _ramp((3, 2, 2), np.uint8)
    p = str(tmp_path / "b.nii.gz")
    _write(p, _header((3, 2, 2), 2, 8, slope=slope, inter=inter, sform=2, srow=np.eye(4)[:3]), data)
    out, _, _ = N.read_nifti(p, dtype=np.float64)
    np.testing.assert_array_equal(out, data)


def test_qform_rotation_and_qfac(tmp_path):
    # quaternion (a,b,c,d) = (0,1,0,0): rotation by pi about x = diag(1,-1,-1); pixdim 2,3,4; qfac -1 flips the third column
    data = np.zeros((2, 2, 2), np.float32)
    for qfac, third in ((1.0, -4.0), (-1.0, 4.0), (0.0, -4.0)):
        p = str(tmp_path / ("q%g.nii" % qfac))
        _write(p, _header((2, 2, 2), 16, 32, pixdim=(qfac, 2, 3, 4, 1, 1, 1, 1), qform=1, quat=(1, 0, 0), qoff=(5, 6, 7)), data)
        _, aff, _ = N.read_nifti(p)
        np.testing.assert_allclose(aff, [[2, 0, 0, 5], [0, -3, 0, 6], [0, 0, third, 7], [0, 0, 0, 1]], atol=1e-12)
    # a 90-degree turn about z: (a,b,c,d) = (cos 45, 0, 0, sin 45) maps x -> y, y -> -x
    s = float(np.sqrt(0.5))
    p = str(tmp_path / "qz.nii")
    _write(p, _header((2, 2, 2), 16, 32, pixdim=(1, 1, 1, 1, 1, 1, 1, 1), qform=1, quat=(0, 0, s)), data)
    _, aff, _ = N.read_nifti(p)
    np.testing.assert_allclose(aff[:3, :3], [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-6)
    # slightly too long a quaternion is tolerated down to the reference's threshold (image_pair.py:24), not beyond
    p = str(tmp_path / "qlong.nii")
    _write(p, _header((2, 2, 2), 16, 32, qform=1, quat=(1.0000003, 0, 0)), data)
    N.read_nifti(p)
    _write(p, _header((2, 2, 2), 16, 32, qform=1, quat=(1.01, 0, 0)), data)
    with pytest.raises(N.NiftiError, match="quaternion"):
        N.read_nifti(p)


def test_base_affine_without_forms(tmp_path):
    data = np.zeros((5, 4, 3), np.uint8)
    p = str(tmp_path / "c.nii")
    _write(p, _header((5, 4, 3), 2, 8, pixdim=(1, 2.0, 1.5, 3.0, 1, 1, 1, 1)), data)
    _, aff, _ = N.read_nifti(p)
    np.testing.assert_allclose(aff, [[-2.0, 0, 0, 4.0], [0, 1.5, 0, -2.25], [0, 0, 3.0, -3.0], [0, 0, 0, 1]])


def test_big_endian_4d_and_extension_offset(tmp_path):
    data = _ramp((3, 2, 2, 2), np.float32) / 7
    p = str(tmp_path / "d.nii.gz")
    # vox_offset 368: a 16-byte header extension sits between the flag and the data
    _write(p, _header((3, 2, 2, 2), 16, 32, e=">", sform=1, srow=np.eye(4)[:3], vox_offset=368.0), data, e=">", pad=20)
    out, _, h = N.read_nifti(p)
    assert h["endian"] == ">" and out.dtype == np.float32 and out.dtype.isnative
    np.testing.assert_array_equal(out, data)
    img, _ = F.load_nifti(p)
    assert img.shape == (3, 2, 2, 2)                                          # 4-D stays [X,Y,Z,C]


def test_rejected_files(tmp_path):
    data = np.zeros((2, 2, 2), np.uint8)
    p = str(tmp_path / "e.nii")
    _write(p, _header((2, 2, 2), 2, 8, magic=b"ni1\0"), data)
    with pytest.raises(N.NiftiError, match="pairs"):
        N.read_nifti(p)
    _write(p, _header((2, 2, 2), 128, 24), data)
    with pytest.raises(N.NiftiError, match="datatype"):
        N.read_nifti(p)
    with open(p, "wb") as f:
        f.write(struct.pack("<i", 540) + b"\0" * 600)
    with pytest.raises(N.NiftiError, match="NIfTI-2"):
        N.read_nifti(p)
    _write(p, _header((4, 4, 4), 2, 8), data)                                 # 8 of 64 voxels
    with pytest.raises(N.NiftiError, match="truncated"):
        N.read_nifti(p)
    with pytest.raises(N.NiftiError, match="dtype"):
        N.write_nifti(p, np.zeros((2, 2, 2), np.complex64), np.eye(4))


def _random_affine(rng, flip):
    q = rng.randn(4); q /= np.linalg.norm(q)
    R = N.quat2mat(q)
    assert abs(np.linalg.det(R) - 1) < 1e-12
    A = np.eye(4)
    A[:3, :3] = R @ np.diag([0.7, 1.3, 2.1 * (-1 if flip else 1)])
    A[:3, 3] = rng.randn(3) * 50
    return A


def test_quaternion_helpers_are_inverse():
    rng = np.random.RandomState(0)
    for _ in range(20):
        q = rng.randn(4); q /= np.linalg.norm(q)
        if q[0] < 0:
            q = -q
        np.testing.assert_allclose(N.mat2quat(N.quat2mat(q)), q, atol=1e-12)
    np.testing.assert_allclose(N.quat2mat((1, 0, 0, 0)), np.eye(3))


@pytest.mark.parametrize("flip", [False, True])
@pytest.mark.parametrize("ext", [".nii", ".nii.gz"])
def test_writer_through_reader_and_independent_parse(tmp_path, flip, ext):
    rng = np.random.RandomState(3 + flip)
    lab = rng.randint(0, 5, (6, 5, 4)).astype(np.uint8)
    A = _random_affine(rng, flip)
    p = str(tmp_path / ("w_PRED" + ext))
    N.write_nifti(p, lab, A)
    out, aff, h = N.read_nifti(p, scaled=False)
    assert out.dtype == np.uint8
    np.testing.assert_array_equal(out, lab)
    np.testing.assert_array_equal(aff[:3], A[:3].astype(np.float32).astype(np.float64))   # sform carries the affine
    # the bytes, parsed here: what nibabel writes for Nifti1Image(data, affine)
    raw = (gzip.open(p, "rb") if ext.endswith(".gz") else open(p, "rb")).read()
    assert struct.unpack("<i", raw[0:4])[0] == 348 and raw[344:348] == b"n+1\0" and raw[348:352] == b"\0\0\0\0"
    assert struct.unpack("<8h", raw[40:56]) == (3, 6, 5, 4, 1, 1, 1, 1)
    assert struct.unpack("<2h", raw[70:74]) == (2, 8)
    assert struct.unpack("<f", raw[108:112])[0] == 352.0
    assert all(np.isnan(struct.unpack("<2f", raw[112:120])))
    assert struct.unpack("<2h", raw[252:256]) == (0, 2)                        # qform unknown, sform aligned
    pix = struct.unpack("<8f", raw[76:108])
    assert pix[0] == (-1.0 if flip else 1.0)
    np.testing.assert_allclose(pix[1:4], [0.7, 1.3, 2.1], rtol=1e-6)
    assert len(raw) == 352 + lab.size
    np.testing.assert_array_equal(np.frombuffer(raw[352:], np.uint8).reshape(lab.shape, order="F"), lab)
    # the qform parameters describe the same affine: read the file again with the sform code cleared
    raw2 = bytearray(raw); raw2[252:256] = struct.pack("<2h", 1, 0)
    p2 = str(tmp_path / "w_q.nii")
    open(p2, "wb").write(bytes(raw2))
    _, aq, _ = N.read_nifti(p2)
    np.testing.assert_allclose(aq, A, atol=2e-5)


def test_float_probabilities_and_identifier(tmp_path):
    probs = np.random.RandomState(1).rand(3, 4, 5, 2).astype(np.float32)
    p = str(tmp_path / "sub-01.T1.nii.gz")
    F.save_nifti(p, probs, np.diag([1.0, 0.8, 1.5, 1.0]))
    img, aff = F.load_nifti(p)
    np.testing.assert_array_equal(img, probs)
    np.testing.assert_allclose(aff, np.diag([1.0, 0.8, 1.5, 1.0]), atol=1e-7)
    assert N.volume_identifier(p) == "sub-01"
    assert N.volume_identifier("/x/y/vol_3.npz") == "vol_3"
    assert N.volume_identifier("a.nii") == "a"


def test_project_folder_with_nifti_volumes(tmp_path):
    from multiplanarunet_amd.data import list_volume_files, load_volume_file, load_label_file, make_toy_volume
    img, lab, _ = make_toy_volume(16, 0)
    A = np.diag([1.0, 1.0, 2.0, 1.0]); A[:3, 3] = [-8, -8, -16]
    os.makedirs(tmp_path / "images"); os.makedirs(tmp_path / "labels")
    F.save_nifti(str(tmp_path / "images" / "case_1.nii.gz"), img[..., 0], A)
    F.save_nifti(str(tmp_path / "labels" / "case_1.nii.gz"), lab, A)
    files = list_volume_files(str(tmp_path))
    assert [os.path.basename(f) for f in files] == ["case_1.nii.gz"]
    im2, l2, a2 = load_volume_file(files[0])
    assert im2.shape == (16, 16, 16, 1) and im2.dtype == np.float32 and l2 is None
    np.testing.assert_array_equal(im2, img)
    np.testing.assert_array_equal(a2, A)
    l3 = load_label_file(str(tmp_path / "labels" / "case_1.nii.gz"))
    assert l3.dtype == np.uint8
    np.testing.assert_array_equal(l3, lab)
