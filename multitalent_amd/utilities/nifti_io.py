"""Image file I/O for the drivers (validate / predict_cases): what the reference does through SimpleITK
(preprocessing/cropping.py:61-81 `load_case_from_list_of_files`, inference/segmentation_export.py:148-160).

SimpleITK is used when it is importable.  Otherwise `.nii` / `.nii.gz` files are read and written by the small NIfTI-1
codec below (single-file format, little endian, scalar volumes, sform or qform geometry): enough for CT volumes and label
maps, nothing else.  Geometry follows ITK's conventions so that the `itk_*` properties of a case mean the same thing either
way: arrays are indexed [z, y, x], spacing / origin are (x, y, z), the direction is the row-major 3x3 cosine matrix in LPS
(NIfTI stores RAS: x and y are negated on the way in and out).  Host code, not part of the device path."""
import gzip
import struct

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16, 768: np.uint32,
           1024: np.int64, 1280: np.uint64}
_CODES = {np.dtype(v).name: k for k, v in _DTYPES.items()}
_LPS = np.diag([-1.0, -1.0, 1.0])


def _have_sitk():
    try:
        import SimpleITK  # noqa: F401
        return True
    except ImportError:
        return False


class Image(object):
    """array [z, y, x] + ITK-style geometry."""

    def __init__(self, array, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), direction=(1, 0, 0, 0, 1, 0, 0, 0, 1)):
        self.array = array
        self.spacing = tuple(float(i) for i in spacing)
        self.origin = tuple(float(i) for i in origin)
        self.direction = tuple(float(i) for i in direction)

    def GetSize(self):
        return tuple(int(i) for i in self.array.shape[::-1])

    def GetSpacing(self):
        return self.spacing

    def GetOrigin(self):
        return self.origin

    def GetDirection(self):
        return self.direction


def _quaternion_to_matrix(b, c, d, qfac):
    a2 = 1.0 - (b * b + c * c + d * d)
    a = np.sqrt(a2) if a2 > 1e-7 else 0.0
    if a == 0.0:
        n = 1.0 / np.sqrt(b * b + c * c + d * d)
        b, c, d = b * n, c * n, d * n
    R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
    if qfac < 0:
        R[:, 2] = -R[:, 2]
    return R


def _matrix_to_quaternion(R):
    """-> (b, c, d, qfac) of a 3x3 orthonormal matrix (nifti1_io's mat44_to_quatern without the polar decomposition)."""
    R = np.array(R, dtype=np.float64)
    qfac = 1.0
    if np.linalg.det(R) < 0:
        R[:, 2] = -R[:, 2]
        qfac = -1.0
    a = R[0, 0] + R[1, 1] + R[2, 2] + 1.0
    if a > 0.5:
        a = 0.5 * np.sqrt(a)
        b, c, d = 0.25 * (R[2, 1] - R[1, 2]) / a, 0.25 * (R[0, 2] - R[2, 0]) / a, 0.25 * (R[1, 0] - R[0, 1]) / a
    else:
        xd, yd, zd = 1.0 + R[0, 0] - (R[1, 1] + R[2, 2]), 1.0 + R[1, 1] - (R[0, 0] + R[2, 2]), 1.0 + R[2, 2] - (R[0, 0] + R[1, 1])
        if xd > 1.0:
            b = 0.5 * np.sqrt(xd)
            c, d, a = 0.25 * (R[0, 1] + R[1, 0]) / b, 0.25 * (R[0, 2] + R[2, 0]) / b, 0.25 * (R[2, 1] - R[1, 2]) / b
        elif yd > 1.0:
            c = 0.5 * np.sqrt(yd)
            b, d, a = 0.25 * (R[0, 1] + R[1, 0]) / c, 0.25 * (R[1, 2] + R[2, 1]) / c, 0.25 * (R[0, 2] - R[2, 0]) / c
        else:
            d = 0.5 * np.sqrt(zd)
            b, c, a = 0.25 * (R[0, 2] + R[2, 0]) / d, 0.25 * (R[1, 2] + R[2, 1]) / d, 0.25 * (R[1, 0] - R[0, 1]) / d
        if a < 0:
            b, c, d = -b, -c, -d
    return float(b), float(c), float(d), qfac


def _read_nifti(fname):
    opener = gzip.open if fname.endswith('.gz') else open
    with opener(fname, 'rb') as f:
        raw = f.read()
    if len(raw) < 348:
        raise IOError("%s: not a NIfTI-1 file (shorter than its header)" % fname)
    end = '<'
    if struct.unpack('<i', raw[:4])[0] != 348:
        end = '>'
        if struct.unpack('>i', raw[:4])[0] != 348:
            raise IOError("%s: not a NIfTI-1 file (sizeof_hdr != 348)" % fname)
    if raw[344:347] != b'n+1':
        raise IOError("%s: only single-file NIfTI-1 ('n+1') is supported" % fname)
    dim = struct.unpack(end + '8h', raw[40:56])
    datatype, = struct.unpack(end + 'h', raw[70:72])
    pixdim = struct.unpack(end + '8f', raw[76:108])
    vox_offset, slope, inter = struct.unpack(end + '3f', raw[108:120])
    qform_code, sform_code = struct.unpack(end + '2h', raw[252:256])
    if datatype not in _DTYPES:
        raise IOError("%s: unsupported NIfTI datatype code %d" % (fname, datatype))
    nd = dim[0]
    if nd < 3 or any(d != 1 for d in dim[4:nd + 1]):
        raise IOError("%s: only 3D scalar volumes are supported (dim = %s)" % (fname, str(dim)))
    nx, ny, nz = dim[1:4]
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(end)
    off = int(vox_offset)
    arr = np.frombuffer(raw, dtype=dt, count=nx * ny * nz, offset=off).reshape(nz, ny, nx)
    arr = arr.astype(dt.newbyteorder('='))
    if slope != 0 and not (slope == 1 and inter == 0) and np.isfinite(slope):
        arr = arr.astype(np.float64) * slope + inter
    # geometry precedence as ITK's NiftiImageIO (the reference reads through SimpleITK): the qform (rotation from the quaternion, spacing
    # from pixdim) when qform_code > 0, else the sform (spacing = column norms); a file whose two transforms disagree then gives the
    # same spacing / direction whether or not SimpleITK is installed
    if qform_code > 0:
        b, c, d, qx, qy, qz = struct.unpack(end + '6f', raw[256:280])
        Rras = _quaternion_to_matrix(b, c, d, -1.0 if pixdim[0] < 0 else 1.0)
        spacing, t = np.abs(np.array(pixdim[1:4], dtype=np.float64)), np.array([qx, qy, qz], dtype=np.float64)
    elif sform_code > 0:
        A = np.array([struct.unpack(end + '4f', raw[280 + 16 * r:296 + 16 * r]) for r in range(3)], dtype=np.float64)
        M, t = A[:, :3], A[:, 3]
        spacing = np.sqrt((M * M).sum(0))
        Rras = M / spacing
    else:
        Rras, spacing, t = np.diag([-1.0, -1.0, 1.0]), np.abs(np.array(pixdim[1:4], dtype=np.float64)), np.zeros(3)
    return Image(arr, spacing, _LPS @ t, (_LPS @ Rras).ravel())


def _write_nifti(img, fname):
    arr = np.ascontiguousarray(img.array)
    if arr.dtype == np.bool_:
        arr = arr.astype(np.uint8)
    if arr.dtype.name not in _CODES:
        raise IOError("cannot write dtype %s as NIfTI" % arr.dtype)
    arr = arr.astype(arr.dtype.newbyteorder('<'))
    nz, ny, nx = arr.shape
    sp = np.array(img.spacing, dtype=np.float64)
    Rras = _LPS @ np.array(img.direction, dtype=np.float64).reshape(3, 3)
    t = _LPS @ np.array(img.origin, dtype=np.float64)
    b, c, d, qfac = _matrix_to_quaternion(Rras)
    hdr = bytearray(348)
    struct.pack_into('<i', hdr, 0, 348)
    struct.pack_into('<8h', hdr, 40, 3, nx, ny, nz, 1, 1, 1, 1)
    struct.pack_into('<hh', hdr, 70, _CODES[arr.dtype.name], arr.dtype.itemsize * 8)
    struct.pack_into('<8f', hdr, 76, qfac, sp[0], sp[1], sp[2], 0, 0, 0, 0)
    struct.pack_into('<3f', hdr, 108, 352.0, 1.0, 0.0)
    hdr[123] = 2                                          # millimetres
    struct.pack_into('<2h', hdr, 252, 1, 1)
    struct.pack_into('<6f', hdr, 256, b, c, d, t[0], t[1], t[2])
    M = Rras * sp
    for r in range(3):
        struct.pack_into('<4f', hdr, 280 + 16 * r, M[r, 0], M[r, 1], M[r, 2], t[r])
    hdr[344:348] = b'n+1\0'
    blob = bytes(hdr) + b'\0\0\0\0' + arr.tobytes()
    if fname.endswith('.gz'):
        with gzip.open(fname, 'wb', compresslevel=1) as f:
            f.write(blob)
    else:
        with open(fname, 'wb') as f:
            f.write(blob)


def read_image(fname):
    """-> Image (array [z, y, x], spacing (x, y, z), origin, direction)."""
    if _have_sitk():
        import SimpleITK as sitk
        im = sitk.ReadImage(fname)
        return Image(sitk.GetArrayFromImage(im), im.GetSpacing(), im.GetOrigin(), im.GetDirection())
    return _read_nifti(fname)


def write_image(array, fname, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), direction=(1, 0, 0, 0, 1, 0, 0, 0, 1)):
    if _have_sitk():
        import SimpleITK as sitk
        im = sitk.GetImageFromArray(array)
        im.SetSpacing(tuple(float(i) for i in spacing))
        im.SetOrigin(tuple(float(i) for i in origin))
        im.SetDirection(tuple(float(i) for i in direction))
        sitk.WriteImage(im, fname)
        return
    _write_nifti(Image(array, spacing, origin, direction), fname)
