"""ctypes binding of the CPU oracle (oracle/libnsc_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnsc_oracle.so")


class Params(C.Structure):
    _fields_ = [
        ("n_elevation", C.c_int32), ("n_azimuth", C.c_int32), ("n_bins", C.c_int32),
        ("target_rows", C.c_int32), ("elev_min_rad", C.c_double), ("elev_max_rad", C.c_double),
        ("min_range", C.c_float), ("max_range", C.c_float), ("epsilon", C.c_float),
        ("interpolate", C.c_int32), ("elev_f64", C.c_int32),
    ]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = [os.path.join(_HERE, f) for f in ("nsc_oracle.c", "keyframe_oracle.c", "nsc_oracle.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libnsc_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp, ip, lp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        pp = C.POINTER(Params)
        L.nsc_oracle_default_params.argtypes = [pp]
        L.nsc_oracle_atan2f.argtypes = [C.c_float, C.c_float]
        L.nsc_oracle_atan2f.restype = C.c_float
        L.nsc_oracle_project.argtypes = [fp, C.c_int64, C.c_int32, pp, fp, ip]
        L.nsc_oracle_project.restype = C.c_int64
        L.nsc_oracle_interpolate.argtypes = [fp, C.c_int32, C.c_int32]
        L.nsc_oracle_bin_lut.argtypes = [C.c_float, C.c_int32, C.c_int32, C.c_float, fp, ip]
        L.nsc_oracle_adaptive_rows.argtypes = [fp, C.c_int32, C.c_int32, C.c_int32, fp]
        L.nsc_oracle_encode_range_image.argtypes = [fp, C.c_int32, pp, ip, fp, fp]
        L.nsc_oracle_encode_points.argtypes = [fp, C.c_int64, C.c_int32, pp, ip, fp, fp, fp]
        L.nsc_oracle_encode_clouds.argtypes = [fp, lp, C.c_int32, C.c_int32, pp, ip, fp, fp, fp,
                                               C.c_int32]
        L.nsc_oracle_voxel_overlap.argtypes = [fp, C.c_int64, fp, C.c_int64, C.c_int32,
                                               C.POINTER(C.c_double), C.c_double, ip]
        L.nsc_oracle_voxel_overlap.restype = C.c_double
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def default_params(**kw):
    p = Params()
    lib().nsc_oracle_default_params(C.byref(p))
    if "elevation_range" in kw:
        lo, hi = kw.pop("elevation_range")
        p.elev_min_rad = float(np.deg2rad(lo))
        p.elev_max_rad = float(np.deg2rad(hi))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def bin_lut(alpha=2.0, n_bins=50, n_freqs=181, epsilon=1e-8):
    edges = np.zeros(n_bins + 1, np.float32)
    lut = np.zeros(n_freqs, np.int32)
    lib().nsc_oracle_bin_lut(alpha, n_bins, n_freqs, epsilon, _f(edges), _i(lut))
    return edges, lut


def _pts(points):
    pts = np.ascontiguousarray(points, dtype=np.float32)
    assert pts.ndim == 2 and pts.shape[1] in (3, 4)
    return pts


def project(points, p=None, want_idx=False):
    p = p or default_params()
    pts = _pts(points)
    img = np.empty((p.n_elevation, p.n_azimuth), np.float32)
    idx = np.empty(len(pts), np.int32) if want_idx else None
    kept = lib().nsc_oracle_project(_f(pts), len(pts), pts.shape[1], C.byref(p), _f(img),
                                    _i(idx) if want_idx else None)
    return (img, idx, kept) if want_idx else img


def project_intensity(points, p=None):
    """range_image.py:216-228 on top of the C projection: (range image, intensity image).  The per-point range
    r = sqrt(clip(x^2) + clip(y^2) + clip(z^2)) is recomputed with numpy float32 ops exactly as :159-162 writes it
    (IEEE sqrt, no vendor math), the pixel index comes from nsc_oracle_project."""
    p = p or default_params()
    pts = _pts(points)
    assert pts.shape[1] == 4
    img, idx, _ = project(pts, p, want_idx=True)
    keep = idx >= 0
    x, y, z = pts[keep, 0], pts[keep, 1], pts[keep, 2]
    r = np.sqrt(np.clip(x ** 2, 0, 1e10) + np.clip(y ** 2, 0, 1e10) + np.clip(z ** 2, 0, 1e10))   # :159-162
    flat = img.reshape(-1)
    li = idx[keep].astype(np.int64)
    closest = r == flat[li]                                                                      # :222
    out = np.zeros(flat.shape, dtype=np.float32)                                                 # :219
    with np.errstate(invalid="ignore"):             # NaN intensities propagate, as in the reference; numpy merely warns
        np.maximum.at(out, li[closest], pts[keep, 3][closest])                                   # :225
    return img, out.reshape(img.shape)


def interpolate_nearest(img):
    """interpolate_range_image(img, 'nearest'), range_image.py:33-47 + :66-87, in numpy (TEST INFRASTRUCTURE; pinned
    by tests/golden/interp_nearest.npz, generated from the reference by oracle/gen_golden_nearest.py)."""
    out = np.array(img, dtype=np.float32, copy=True)
    E, A = out.shape
    for r in range(E):
        row = out[r]
        valid = np.where(row > 0)[0]
        if len(valid) == 0 or len(valid) == A:
            continue
        vals = row.copy()
        for c in np.where(~(row > 0))[0]:
            d = np.minimum(np.abs(valid - c), A - np.abs(valid - c))     # :69-72
            out[r, c] = vals[valid[np.argmin(d)]]                        # first minimum = smaller column on a tie
    for r in range(E):                                                   # :77-87
        if not np.any(out[r] > 0):
            for k in range(1, E):
                if r - k >= 0 and np.any(out[r - k] > 0):
                    out[r] = out[r - k]
                    break
                if r + k < E and np.any(out[r + k] > 0):
                    out[r] = out[r + k]
                    break
    return out


def interpolate(img):
    out = np.ascontiguousarray(img, dtype=np.float32).copy()
    lib().nsc_oracle_interpolate(_f(out), out.shape[0], out.shape[1])
    return out


def encode_range_image(img, p=None, lut=None, want_mags=False):
    p = p or default_params()
    if lut is None:
        lut = bin_lut(2.0, p.n_bins, p.n_azimuth // 2 + 1, p.epsilon)[1]
    img = np.ascontiguousarray(img, dtype=np.float32)
    assert img.shape[1] == p.n_azimuth
    desc = np.empty(p.target_rows * p.n_bins, np.float32)
    mags = np.empty((p.target_rows, p.n_azimuth // 2 + 1), np.float32) if want_mags else None
    lib().nsc_oracle_encode_range_image(_f(img), img.shape[0], C.byref(p), _i(lut), _f(desc),
                                        _f(mags) if want_mags else None)
    return (desc, mags) if want_mags else desc


def encode_points(points, p=None, lut=None, want_images=False):
    p = p or default_params()
    if lut is None:
        lut = bin_lut(2.0, p.n_bins, p.n_azimuth // 2 + 1, p.epsilon)[1]
    pts = _pts(points)
    desc = np.empty(p.target_rows * p.n_bins, np.float32)
    raw = np.empty((p.n_elevation, p.n_azimuth), np.float32)
    itp = np.empty((p.n_elevation, p.n_azimuth), np.float32)
    lib().nsc_oracle_encode_points(_f(pts), len(pts), pts.shape[1], C.byref(p), _i(lut), _f(desc),
                                   _f(raw), _f(itp))
    return (desc, raw, itp) if want_images else desc


def encode_clouds(points, offsets, p=None, lut=None, n_threads=1, want_images=False):
    """points: (sum_n, 3|4) float32, offsets: (n_clouds+1,) int64 point offsets."""
    p = p or default_params()
    if lut is None:
        lut = bin_lut(2.0, p.n_bins, p.n_azimuth // 2 + 1, p.epsilon)[1]
    pts = _pts(points)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = len(offsets) - 1
    desc = np.empty((n, p.target_rows * p.n_bins), np.float32)
    raw = np.empty((n, p.n_elevation, p.n_azimuth), np.float32) if want_images else None
    itp = np.empty((n, p.n_elevation, p.n_azimuth), np.float32) if want_images else None
    lib().nsc_oracle_encode_clouds(
        _f(pts), offsets.ctypes.data_as(C.POINTER(C.c_int64)), n, pts.shape[1], C.byref(p),
        _i(lut), _f(desc), _f(raw) if want_images else None, _f(itp) if want_images else None,
        n_threads)
    return (desc, raw, itp) if want_images else desc
